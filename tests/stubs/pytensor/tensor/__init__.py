import numpy as np

from pytensor.graph.basic import Apply, Constant, TensorType, Variable
from pytensor.graph.op import Op

dscalar, dvector, dmatrix, dtensor3 = (TensorType(k) for k in range(4))


class _Lambda(Op):
    """Elementwise / shape op defined by a numpy function."""

    def __init__(self, fn, ndim):
        self.fn, self.out_ndim = fn, ndim

    def make_node(self, *inputs):
        return Apply(self, inputs, 1, [TensorType(self.out_ndim)])

    def perform(self, node, inputs, outputs):
        outputs[0][0] = np.asarray(self.fn(*inputs), dtype=np.float64)


def as_tensor_variable(x, dtype=None):
    if isinstance(x, Variable):
        return x
    return Constant(np.asarray(x, dtype=dtype or np.float64))


def concatenate(items, axis=0):
    items = [as_tensor_variable(i) for i in items]
    return _Lambda(lambda *a: np.concatenate(a, axis=axis), items[0].ndim)(*items)


def zeros_like(x):
    x = as_tensor_variable(x)
    return _Lambda(lambda a: np.zeros_like(a), x.ndim)(x)


def sum(x, axis=None):          # noqa: A001 - pytensor's name
    return as_tensor_variable(x).sum(axis)
