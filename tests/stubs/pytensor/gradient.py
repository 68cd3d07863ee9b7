class NotImplementedGrad:
    """What pytensor returns for an input an Op declares non-differentiable."""

    def __init__(self, op, idx, var, comment=""):
        self.op, self.idx, self.var, self.comment = op, idx, var, comment


def grad_not_implemented(op, idx, var, comment=""):
    return NotImplementedGrad(op, idx, var, comment)


class DisconnectedType:
    """As in pytensor: a TYPE; the cotangent of an output the cost does not depend on is a variable of this type
    (``DisconnectedType()()``), which is what ``Op.grad`` implementations test with ``isinstance(g.type, ...)``."""
    ndim = 0

    def __call__(self, name=None):
        from pytensor.graph.basic import Variable
        return Variable(self, name=name)
