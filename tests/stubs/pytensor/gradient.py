class NotImplementedGrad:
    """What pytensor returns for an input an Op declares non-differentiable."""

    def __init__(self, op, idx, var, comment=""):
        self.op, self.idx, self.var, self.comment = op, idx, var, comment


def grad_not_implemented(op, idx, var, comment=""):
    return NotImplementedGrad(op, idx, var, comment)


class DisconnectedType:
    pass
