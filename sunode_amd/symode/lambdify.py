"""Symbolic helper functions and rewrite rules of the code generator's front end.

Counterpart of the second half of the reference's ``sunode/symode/lambdify.py``
(``/root/reference/sunode/symode/lambdify.py:275-432``); importable under the reference's name
``sunode.symode.lambdify`` through the alias package.  The first half of that file (sympy -> Python AST -> numba,
``:38-270``) is replaced by ``sunode_amd/symode/codegen.py`` (sympy -> HIP C++ device functions).

What a model author gets, with the reference's names and argument meaning:

* ``logaddexp(a, b)``, ``expit(x)``, ``dexpit(x)``: sympy functions that stay *one call* in the generated code
  (``sa_logaddexp / sa_expit / sa_dexpit`` of the generated header, evaluated with the deterministic ``sa_exp`` /
  ``sa_log1p`` -- codegen.MATH_C) instead of being expanded into overflow-prone ``exp`` / ``log`` expressions;
* ``CardinalBSpline(degree, x)``: the cardinal B-spline with knots 0, 1, ..., degree + 1;
* ``interpolate_spline(x, vals, lower, upper, degree)``: a smooth function of (usually) time through coefficients that
  may be parameters -- the reference's way of putting a time-varying input into a right-hand side;
* the log-sum-exp rewrite rules (``logsumexp_2terms_opt``, ``explog_opt``, ``simplify_multiple_exp_sum``) meant to be
  passed as ``SympyProblem(..., simplify=...)``.

Differences from the reference, on purpose (each one is a defect of the reference that makes the call raise there):

* ``expit.fdiff`` / ``dexpit.fdiff`` test ``argindex != 1`` (reference ``:301,318``), so differentiating with respect
  to the only argument ends in ``raise ArgumentIndexError`` -- a name the file never imports -> ``NameError``.  Here
  d expit(x)/dx = dexpit(x) and d dexpit(x)/dx = dexpit(x) (1 - 2 expit(x)), the formulas the reference states.
* ``CardinalBSpline.as_sympy_expr`` calls ``sym.horner`` with ``sym`` undefined (reference ``:339``) -> ``NameError``;
  here it returns the piecewise polynomial in Horner form.  The reference has no derivative rule for the spline (its
  argument may then only depend on time and on parameters that are not differentiated); here
  d B_d(x)/dx = B_{d-1}(x) - B_{d-1}(x - 1), and degrees other than 4 -- for which the reference's numba helper
  returns NaN (``:74-77``) -- are generated from the same piecewise form.
"""
from __future__ import annotations

from functools import partial

import sympy as sy
import sympy.codegen.rewriting
from sympy.assumptions import Q, ask
from sympy.core.function import ArgumentIndexError

__all__ = ["logaddexp", "expit", "dexpit", "CardinalBSpline", "interpolate_spline", "logsumexp_2terms_opt",
           "explog_opt", "simplify_multiple_exp_sum", "is_exp_sum", "is_exp_sum_pow", "is_exp_sum_pow_mult",
           "is_multiple_exp_sum_pow_mult"]


class logaddexp(sy.Function):
    """log(exp(a) + exp(b)) (reference ``:275-291``); derivative = the softmax weight of the argument."""
    nargs = 2

    def fdiff(self, argindex=1):
        if argindex not in (1, 2):
            raise ArgumentIndexError(self, argindex)
        a, b = self.args
        return sy.exp(self.args[argindex - 1]) / (sy.exp(a) + sy.exp(b))

    def _eval_is_real(self):
        return self.args[0].is_real and self.args[1].is_real

    def _eval_is_finite(self):
        return self.args[0].is_finite and self.args[1].is_finite

    def _eval_rewrite_as_log(self, a, b, **kwargs):
        return sy.log(sy.exp(a) + sy.exp(b))


class expit(sy.Function):
    """1 / (1 + exp(-x)) (reference ``:311-325``)."""
    nargs = 1

    def fdiff(self, argindex=1):
        if argindex != 1:
            raise ArgumentIndexError(self, argindex)
        return dexpit(self.args[0])

    def _eval_is_real(self):
        return self.args[0].is_real

    def _eval_is_positive(self):
        return self.args[0].is_real

    def _eval_rewrite_as_exp(self, x, **kwargs):
        return 1 / (1 + sy.exp(-x))


class dexpit(sy.Function):
    """expit'(x) = expit(x) expit(-x) (reference ``:294-308``)."""
    nargs = 1

    def fdiff(self, argindex=1):
        if argindex != 1:
            raise ArgumentIndexError(self, argindex)
        x = self.args[0]
        return dexpit(x) * (1 - 2 * expit(x))

    def _eval_is_real(self):
        return self.args[0].is_real

    def _eval_rewrite_as_exp(self, x, **kwargs):
        return 1 / ((1 + sy.exp(-x)) * (1 + sy.exp(x)))


class CardinalBSpline(sy.Function):
    """B-spline basis function of the given degree on the knots 0, 1, ..., degree + 1 (reference ``:328-340``)."""
    nargs = 2

    def as_sympy_expr(self):
        degree, x = self.args
        knots = tuple(sy.Integer(i) for i in range(int(degree) + 2))
        basis = sy.functions.special.bsplines.bspline_basis(int(degree), knots, 0, x)
        pieces = []
        for expr, cond in basis.args:
            pieces.append((sy.horner(sy.expand(expr), wrt=x) if expr.has(x) else expr, cond))
        return sy.Piecewise(*pieces)

    def fdiff(self, argindex=2):
        if argindex != 2:
            raise ArgumentIndexError(self, argindex)
        degree, x = self.args
        if int(degree) < 1:
            return sy.Integer(0)
        return CardinalBSpline(degree - 1, x) - CardinalBSpline(degree - 1, x - 1)

    def _eval_is_real(self):
        return self.args[1].is_real


def interpolate_spline(x, vals, lower, upper, degree, as_pure=False):
    """sum_i vals[i] B_degree(u - i), with u the image of x under the affine map that sends [lower, upper] onto the
    part of the knot range where the basis functions sum to one (reference ``:343-352``)."""
    vals = list(vals)
    n_knots = degree + len(vals) + 1
    basis = partial(CardinalBSpline, degree)
    u = (x - lower) / (upper - lower)
    u = degree + u * (n_knots - 2 * degree - 1)
    members = [basis(u - i) for i in range(len(vals))]
    if as_pure:
        members = [b.as_sympy_expr() for b in members]
    return sum(v * b for v, b in zip(vals, members))


# ---------------------------------------------------------------------------------------------------------------
# rewrite rules (reference :355-432): keep products / quotients of exponential sums in log space
# ---------------------------------------------------------------------------------------------------------------
def _is_two_exp_log(e):
    return (isinstance(e, sy.log) and e.args[0].is_Add and len(e.args[0].args) == 2
            and all(isinstance(term, sy.exp) for term in e.args[0].args))


#: log(exp(a) + exp(b)) -> logaddexp(a, b)
logsumexp_2terms_opt = sympy.codegen.rewriting.ReplaceOptim(
    _is_two_exp_log,
    lambda e: logaddexp(e.args[0].args[0].args[0], e.args[0].args[1].args[0]),
)


def is_exp_sum(expr):
    """exp(a), or exp(a) + exp(b)."""
    if isinstance(expr, sy.exp):
        return True
    return isinstance(expr, sy.Add) and len(expr.args) == 2 and all(isinstance(e, sy.exp) for e in expr.args)


def is_exp_sum_pow(expr):
    """... or a power of one."""
    return is_exp_sum(expr) or (isinstance(expr, sy.Pow) and is_exp_sum(expr.args[0]))


def is_exp_sum_pow_mult(expr):
    """... or a product with such a factor."""
    return is_exp_sum_pow(expr) or (isinstance(expr, sy.Mul) and any(is_exp_sum_pow(e) for e in expr.args))


def is_multiple_exp_sum_pow_mult(expr):
    """A product with more than one such factor: the shape whose direct evaluation overflows first."""
    return isinstance(expr, sy.Mul) and sum(1 for e in expr.args if is_exp_sum_pow_mult(e)) > 1


def simplify_multiple_exp_sum(expr, do_simplify=False, optims=None):
    """A sub-expression of known sign s is rewritten as s exp(log(s expr)) with the logarithm expanded into a sum
    and its two-term log-sum-exps collected into ``logaddexp`` / ``log1p`` calls; everything else is visited
    recursively (reference ``:404-424``)."""
    if optims is None:
        optims = (sympy.codegen.rewriting.log1p_opt, logsumexp_2terms_opt)
    positive, negative = ask(Q.positive(expr)), ask(Q.negative(expr))
    if not (positive or negative):
        if expr.args:
            return expr.func(*[simplify_multiple_exp_sum(arg, do_simplify, optims) for arg in expr.args])
        return expr
    sign = 1 if positive else -1
    # expand_log does not see assumptions made through a context manager: force, the sign is known
    log_expr = sy.expand_log(sy.log(sign * expr), force=True)
    log_expr = sympy.codegen.rewriting.optimize(log_expr, optims)
    return sign * sy.exp(log_expr, evaluate=False)


explog_opt = sympy.codegen.rewriting.ReplaceOptim(
    lambda e: bool(ask(Q.positive(e)) or ask(Q.negative(e))) and is_multiple_exp_sum_pow_mult(e),
    simplify_multiple_exp_sum,
)
