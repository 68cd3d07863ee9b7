class _Type:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "stub.%s" % self.name


dscalar, dvector, dmatrix, dtensor3 = (_Type(n) for n in ("dscalar", "dvector", "dmatrix", "dtensor3"))


def as_tensor_variable(*args, **kwargs):
    raise NotImplementedError("graph construction needs the real pytensor")


concatenate = zeros_like = sum = as_tensor_variable
