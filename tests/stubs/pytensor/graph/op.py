from pytensor.graph.basic import Apply


class Op:
    """itypes / otypes Ops: calling one builds an Apply node; ``perform`` runs it."""
    itypes = otypes = None

    def make_node(self, *inputs):
        from pytensor.tensor import as_tensor_variable
        ins = [as_tensor_variable(i) for i in inputs]
        if self.itypes is not None:
            if len(ins) != len(self.itypes):
                raise TypeError("%s: expected %d inputs, got %d" % (type(self).__name__, len(self.itypes), len(ins)))
            for v, t in zip(ins, self.itypes):
                if v.ndim != t.ndim:
                    raise TypeError("%s: input rank %d, expected %d" % (type(self).__name__, v.ndim, t.ndim))
        return Apply(self, ins, len(self.otypes), self.otypes)

    def __call__(self, *inputs):
        node = self.make_node(*inputs)
        return node.outputs[0] if len(node.outputs) == 1 else node.outputs
