"""sympy -> HIP C++ / C emitter for the five callbacks of the hot path.

Replaces the reference's sympy -> Python-AST -> numba pipeline
(/root/reference/sunode/symode/lambdify.py:82-146 ``LambdifyAST.add_var_function``,
``:203-270 lambdify_consts``) with an emitter of plain C99 function bodies that
compile unchanged as ``__device__`` functions for gfx950 (the product path) and
as host functions (test oracle / CPU baseline).  Kept from the reference:

* ``sympy.cse`` over the ravelled expression array (``lambdify.py:253-256``);
* the output is zero-filled and only structurally non-zero entries are
  assigned (``lambdify.py:102-127``) -- we emit the explicit ``= 0.0`` stores;
* helper functions ``logaddexp / expit / dexpit / CardinalBSpline(4, t)``
  (``lambdify.py:59-77``);
* non-finite outputs make the callback return 1 = "recoverable error"
  (``symode/problem.py:266-269``).

Different from the reference on purpose: integer powers are expanded to
products and no ``fastmath`` style re-association is allowed, so that host and
device evaluate bit-identical arithmetic (both sides are compiled with
``-ffp-contract=off``); this is what makes the step/order bookkeeping of the
HIP integrator comparable bit-for-bit with the CPU oracle.

Callback ABI (all arrays are flat ``double``):

    int sa_rhs     (double t, const double* y, const double* ps, const double* pr, double* out /*n*/);
    int sa_jac     (double t, const double* y, const double* ps, const double* pr, double* out /*n*n col-major*/);
    int sa_adj_rhs (double t, const double* y, const double* lam, const double* ps, const double* pr, double* out /*n*/);
    int sa_quad_rhs(double t, const double* y, const double* lam, const double* ps, const double* pr, double* out /*p*/);
    int sa_adj_jac (double t, const double* y, const double* ps, const double* pr, double* out /*n*n col-major*/);

``ps`` = differentiated parameters in ``subset_paths`` order, ``pr`` = the
remaining parameters in declaration order (same split as the reference's
pytensor Ops pass around, ``wrappers/as_pytensor.py:86-106``).
"""
from __future__ import annotations

import hashlib
import re
from itertools import count
from typing import Optional,  Dict, Iterable, List, Sequence, Tuple

import numpy as np
import sympy as sym
from sympy.printing.c import C99CodePrinter

MAX_EXPANDED_POW = 8
#: sums with at least this many terms are emitted as SUM_WAYS interleaved partial sums
LONG_SUM_TERMS = 16
SUM_WAYS = 4

HELPERS_C = r"""
#ifndef SA_FN
#error "SA_FN must be defined (e.g. 'static inline' or '__device__ __forceinline__')"
#endif
/* Output plumbing: by default the callbacks write a flat array.  A kernel may instead pass a sink
   object (SA_TEMPLATE / SA_OUT_T) and redefine SA_STORE to keep only the slots a lane owns. */
#ifndef SA_STORE
#define SA_STORE(slot, value) out[slot] = (value)
#endif
#ifndef SA_OUT_T
#define SA_OUT_T double*
#endif
/* input element access: flat arrays by default; the memory-resident kernels stride them */
#ifndef SA_Y
#define SA_Y(i) y[i]
#endif
#ifndef SA_LAM
#define SA_LAM(i) lam[i]
#endif
#ifndef SA_PS
#define SA_PS(j) ps[j]
#endif
#ifndef SA_PR
#define SA_PR(j) pr[j]
#endif
/* chunked callbacks: chunk `c` contributes to the return code; a multi-wavefront kernel redefines
   this to give every wavefront its share of the chunks */
#ifndef SA_CHUNK_CALL
#define SA_CHUNK_CALL(c, call) bad |= call
#endif
/* after every output statement (kernels with huge callbacks fence the instruction scheduler here) */
#ifndef SA_STMT_END
#define SA_STMT_END
#endif
/* chunk functions name the slots lo..hi of the remaining-parameter vector they are about to read */
#ifndef SA_PREFETCH_PR
#define SA_PREFETCH_PR(k, lo, hi)     /* k = 0..7: ordinal of the range within the function */
#endif
/* last statement of every callback body before the return */
#ifndef SA_EPILOGUE
#define SA_EPILOGUE
#endif
/* first statement of every callback body (the wave kernel derives a scalar-load view of pr here) */
#ifndef SA_PROLOGUE
#define SA_PROLOGUE
#endif
#ifndef SA_TEMPLATE
#define SA_TEMPLATE
#endif
/* Dense matrix-vector block of a callback (symode/problem.py extract_matvec):
     SA_MATVEC(tag, NO, NI, OFF, VEC)   computes  mv[i] = sum_j M[j*NO + i] * VEC(j),  i < NO, j < NI,
   with M = the j-major block of the remainder vector starting at slot OFF, in THIS association (part of the
   generated arithmetic, identical in every kernel and in the oracle): four interleaved accumulators
   a_w = M[w][i]*v[w], then a_w = fma(M[j][i], v[j], a_w) for j = w+4, w+8, ...; mv[i] = (a_0 + a_1) + (a_2 + a_3)
   (accumulators without a term are +0.0).  SA_MV(tag, i) reads mv[i]; SA_OWNS(slot) lets a multi-wavefront kernel
   give every wavefront its share of the output statements that follow. */
#ifndef SA_MATVEC
#define SA_MATVEC(tag, NO, NI, OFF, VEC) \
    double sa_mv_##tag[NO]; \
    for (int i_ = 0; i_ < (NO); i_++) { \
        double a_[4] = {0.0, 0.0, 0.0, 0.0}; \
        for (int w_ = 0; w_ < 4 && w_ < (NI); w_++) { \
            double acc_ = SA_PR((OFF) + w_ * (NO) + i_) * VEC(w_); \
            for (int j_ = w_ + 4; j_ < (NI); j_ += 4) acc_ = fma(SA_PR((OFF) + j_ * (NO) + i_), VEC(j_), acc_); \
            a_[w_] = acc_; \
        } \
        sa_mv_##tag[i_] = (a_[0] + a_[1]) + (a_[2] + a_[3]); \
    }
#define SA_MV(tag, i) sa_mv_##tag[i]
#endif
#ifndef SA_OWNS
#define SA_OWNS(slot) 1
#endif
SA_FN double sa_logaddexp(double a, double b) {
    double lo = fmin(a, b), hi = fmax(a, b);
    return hi + log1p(exp(lo - hi));
}
SA_FN double sa_expit(double x) { return 1.0 / (1.0 + exp(-x)); }
SA_FN double sa_dexpit(double x) { return sa_expit(x) * sa_expit(-x); }
SA_FN double sa_cardinal_bspline4(double t) {
    if (t >= 0.0 && t <= 1.0) return (1.0/24.0)*t*t*t*t;
    if (t >= 1.0 && t <= 2.0) return t*(t*(t*(5.0/6.0 - 1.0/6.0*t) - 5.0/4.0) + 5.0/6.0) - 5.0/24.0;
    if (t >= 2.0 && t <= 3.0) return t*(t*(t*((1.0/4.0)*t - 5.0/2.0) + 35.0/4.0) - 25.0/2.0) + 155.0/24.0;
    if (t >= 3.0 && t <= 4.0) return t*(t*(t*(5.0/2.0 - 1.0/6.0*t) - 55.0/4.0) + 65.0/2.0) - 655.0/24.0;
    if (t >= 4.0 && t <= 5.0) return t*(t*(t*((1.0/24.0)*t - 5.0/6.0) + 25.0/4.0) - 125.0/6.0) + 625.0/24.0;
    return 0.0;
}
"""


class HipExprPrinter(C99CodePrinter):
    """C99 printer with symbol -> array-slot mapping and product-expanded powers."""

    def __init__(self, symbol_map: Dict[str, str]):
        super().__init__({"allow_unknown_functions": False})
        self._symbol_map = symbol_map

    def _print_Symbol(self, expr):
        name = expr.name
        if name in self._symbol_map:
            return self._symbol_map[name]
        return super()._print_Symbol(expr)

    def _print_Float(self, expr):
        return repr(float(expr))

    def _print_Add(self, expr, order=None):
        """Long sums are printed as SUM_WAYS interleaved accumulators, each a chain of explicit
        fused multiply-adds, combined pairwise: a left-to-right sum of 100+ products is one
        dependent chain of multiplies and adds that a GPU lane (and a CPU core) can only retire at
        the add latency.  Association and fusing are part of the generated source, so the oracle
        and every kernel round identically; short sums keep the plain C form."""
        from sympy.printing.precedence import PRECEDENCE
        terms = self._as_ordered_terms(expr, order=order)
        if len(terms) < LONG_SUM_TERMS:
            return super()._print_Add(expr, order=order)

        def factors(term):
            """term -> (A, B) printed factors of a product term, or None"""
            coeff, rest = term.as_coeff_Mul()
            parts = list(rest.as_ordered_factors()) if rest.is_Mul else [rest]
            if coeff == 1 or coeff == -1:
                if len(parts) < 2:
                    return None
                a_text = self.parenthesize(parts[0], PRECEDENCE["Mul"])
                if coeff == -1:
                    a_text = "-" + a_text
                b_expr = sym.Mul(*parts[1:])
            else:
                a_text = self._print(coeff)
                b_expr = rest
            return a_text, self.parenthesize(b_expr, PRECEDENCE["Mul"])

        groups = []
        for w in range(SUM_WAYS):
            acc = None
            for term in terms[w::SUM_WAYS]:
                if acc is None:
                    acc = "(%s)" % self._print(term)
                    continue
                ab = factors(term)
                if ab is None:
                    acc = "(%s + (%s))" % (acc, self._print(term))
                else:
                    acc = "fma(%s, %s, %s)" % (ab[0], ab[1], acc)
            groups.append(acc)
        while len(groups) > 1:
            groups = ["(%s + %s)" % (groups[i], groups[i + 1]) for i in range(0, len(groups), 2)]
        return groups[0]

    def _print_Integer(self, expr):
        # keep integers exact but typed double so that 1/2 can never appear
        return "%d.0" % int(expr) if abs(int(expr)) < 2**53 else repr(float(expr))

    def _print_Rational(self, expr):
        return "(%d.0/%d.0)" % (expr.p, expr.q)

    def _print_Pow(self, expr):
        base, exp = expr.base, expr.exp
        if exp.is_Integer:
            e = int(exp)
            if e == 0:
                return "1.0"
            if 1 <= abs(e) <= MAX_EXPANDED_POW:
                b = "(%s)" % self._print(base)
                prod = "*".join([b] * abs(e))
                return "(%s)" % prod if e > 0 else "(1.0/(%s))" % prod
        if exp == sym.Rational(1, 2):
            return "sqrt(%s)" % self._print(base)
        if exp == sym.Rational(-1, 2):
            return "(1.0/sqrt(%s))" % self._print(base)
        return "pow(%s, %s)" % (self._print(base), self._print(exp))

    # helper functions of the reference (lambdify.py:59-77, 275-340)
    def _print_logaddexp(self, expr):
        return "sa_logaddexp(%s, %s)" % tuple(self._print(a) for a in expr.args)

    def _print_expit(self, expr):
        return "sa_expit(%s)" % self._print(expr.args[0])

    def _print_dexpit(self, expr):
        return "sa_dexpit(%s)" % self._print(expr.args[0])

    def _print_CardinalBSpline(self, expr):
        degree, x = expr.args
        if degree != 4:
            return "(0.0/0.0)"      # the reference returns nan for degree != 4
        return "sa_cardinal_bspline4(%s)" % self._print(x)

    def _print_Heaviside(self, expr):
        x = self._print(expr.args[0])
        return "((%s) > 0.0 ? 1.0 : ((%s) < 0.0 ? 0.0 : 0.5))" % (x, x)

    def _print_sign(self, expr):
        x = self._print(expr.args[0])
        return "(((%s) > 0.0) - ((%s) < 0.0))" % (x, x)


#: callbacks with more output statements than this are emitted as a chain of chunk functions
#: (compile time of one huge basic block is superlinear; 10^4 Jacobian entries at n = 100)
CHUNK_STATEMENTS = 400
#: ... and callbacks with more generated text than this are split into up to MAX_COST_CHUNKS chunks
CHUNK_COST = 24000
MAX_COST_CHUNKS = 4
#: chunk functions announce the remaining-parameter ranges they read (SA_PREFETCH_PR): indices closer than
#: PREFETCH_GAP are one range, ranges shorter than PREFETCH_MIN are not worth a touch
PREFETCH_GAP = 16
PREFETCH_MIN = 32
PREFETCH_MAX_RANGES = 8
PREFETCH_PIECE = 1024          # doubles covered by one touch (64 lanes x one 128-byte line)


def _closure(needed, deps, order):
    """CSE temporaries (in definition order) that the expressions using ``needed`` depend on."""
    seen = set()
    stack = list(needed)
    while stack:
        name = stack.pop()
        if name in seen or name not in deps:
            continue
        seen.add(name)
        stack.extend(deps[name])
    return [name for name in order if name in seen]


def emit_function(
    name: str,
    signature: str,
    expr: np.ndarray,
    out_index: Sequence[int],
    n_out: int,
    symbol_map: Dict[str, str],
    prefix: str,
    matvec: Optional[Dict[str, object]] = None,
) -> str:
    """One callback.  ``expr`` is ravelled; ``out_index[k]`` is the flat output
    slot of ``expr.ravel()[k]`` (lets the caller pick column-major storage).
    ``matvec`` (tag, n_out, n_in, offset, vec): the expressions refer to ``SA_MV(tag, i)``, the results of one
    dense matrix-vector product evaluated up front by ``SA_MATVEC``; the function is then emitted as ONE body whose
    output statements carry ``SA_OWNS(slot)`` guards (the kernels split those between wavefronts)."""
    flat = [sym.sympify(e) for e in np.asarray(expr, dtype=object).ravel()]
    names = (sym.Symbol("%s%d" % (prefix, i)) for i in count())
    if flat:
        assigns, reduced = sym.cse(flat, symbols=names, order="canonical")
    else:
        assigns, reduced = [], []
    printer = HipExprPrinter(symbol_map)
    temp_text = {var.name: printer.doprint(value) for var, value in assigns}
    temp_order = [var.name for var, _ in assigns]
    temp_names = set(temp_order)
    temp_deps = {var.name: [s.name for s in value.free_symbols if s.name in temp_names] for var, value in assigns}
    written = {}
    uses = {}
    for k, value in enumerate(reduced):
        slot = int(out_index[k])
        written[slot] = "0.0" if value == 0 else printer.doprint(value)
        uses[slot] = [s.name for s in sym.sympify(value).free_symbols if s.name in temp_names]

    # Outputs go through SA_STORE(slot, value): plain kernels / the oracle define it as
    # `out[slot] = value`; the cooperative kernel (one lane per state component) keeps only the
    # slots a lane owns.  x*0.0 is (+-)0 for finite x and NaN for inf/nan: the finiteness check is
    # straight-line on purpose (constant subscripts only, see bdf_kernels.hip on scalar replacement).
    def body(slots, temps, prefetch=False):
        lines = ["    SA_PROLOGUE"]
        if matvec is not None:
            lines.append("    SA_MATVEC(%s, %d, %d, %d, %s);" % (matvec["tag"], matvec["n_out"], matvec["n_in"],
                                                                matvec["offset"], matvec["vec"]))
        stmts = ["    const double %s = %s; SA_STMT_END" % (tname, temp_text[tname]) for tname in temps]
        if prefetch:
            # remaining-parameter slots this chunk reads, as a few contiguous ranges: a kernel may touch them
            # up front so that the statements' own loads hit the cache (SA_PREFETCH_PR)
            text = "".join(stmts) + "".join(written.get(slot, "") for slot in slots)
            idx = sorted({int(k) for k in re.findall(r"SA_PR\((\d+)\)", text)})
            ranges = []
            for k in idx:
                if ranges and k - ranges[-1][1] <= PREFETCH_GAP:
                    ranges[-1][1] = k
                else:
                    ranges.append([k, k])
            pieces = []
            for lo, hi in ranges:
                if hi - lo < PREFETCH_MIN:
                    continue
                for start in range(lo, hi + 1, PREFETCH_PIECE):
                    pieces.append((start, min(start + PREFETCH_PIECE - 1, hi)))
            if len(pieces) <= PREFETCH_MAX_RANGES:
                lines += ["    SA_PREFETCH_PR(%d, %d, %d);" % (k, lo, hi) for k, (lo, hi) in enumerate(pieces)]
        lines += stmts
        lines.append("    double chk = 0.0;")
        for slot in slots:
            text = written.get(slot, "0.0")
            if text == "0.0":
                lines.append("    SA_STORE(%d, 0.0);" % slot)
            elif matvec is not None:
                lines.append("    if (SA_OWNS(%d)) { const double v_ = %s; SA_STORE(%d, v_); chk += v_ * 0.0; } SA_STMT_END"
                             % (slot, text, slot))
            else:
                lines.append("    { const double v_ = %s; SA_STORE(%d, v_); chk += v_ * 0.0; } SA_STMT_END"
                             % (text, slot))
        lines.append("    (void)t; (void)y; (void)ps; (void)pr; (void)out;")
        lines.append("    SA_EPILOGUE")
        lines.append("    return (chk == 0.0) ? 0 : 1;")
        return lines

    # One function for small callbacks; otherwise a chain of chunk functions balanced by the amount of
    # generated text (a proxy for the arithmetic): at most CHUNK_STATEMENTS statements per chunk and,
    # for expensive callbacks, up to MAX_COST_CHUNKS chunks so that a kernel can spread them over
    # several wavefronts (SA_CHUNK_CALL).
    def _cost(slot):
        temps = _closure(uses.get(slot, []), temp_deps, temp_order)
        return len(written.get(slot, "0.0")) + 16 + sum(len(temp_text[tn]) for tn in temps)

    cost = [_cost(slot) for slot in range(n_out)]
    total = sum(cost)
    n_chunks = -(-n_out // CHUNK_STATEMENTS)
    if total > CHUNK_COST:
        n_chunks = max(n_chunks, min(MAX_COST_CHUNKS, -(-total // CHUNK_COST), n_out))
    if n_chunks <= 1 or matvec is not None:
        lines = ["SA_TEMPLATE SA_FN int %s(%s) {" % (name, signature)]
        lines += body(range(n_out), temp_order)
        lines.append("}")
        return "\n".join(lines)

    # chunked form: same expressions, same evaluation order inside every statement; a temporary
    # needed by several chunks is recomputed in each (identical value)
    bounds, acc, target = [0], 0, total / n_chunks
    for slot in range(n_out):
        acc += cost[slot]
        full = (slot + 1 - bounds[-1]) >= CHUNK_STATEMENTS
        if (acc >= target * len(bounds) or full) and slot + 1 < n_out and len(bounds) < n_chunks:
            bounds.append(slot + 1)
    bounds.append(n_out)
    call_args = ", ".join(part.split()[-1].lstrip("*") for part in signature.split(","))
    parts, calls = [], []
    for c in range(len(bounds) - 1):
        slots = range(bounds[c], bounds[c + 1])
        needed = [u for slot in slots for u in uses.get(slot, [])]
        cname = "%s_c%d" % (name, c)
        parts.append("\n".join(["SA_TEMPLATE SA_FN int %s(%s) {" % (cname, signature)]
                               + body(slots, _closure(needed, temp_deps, temp_order), prefetch=True) + ["}"]))
        calls.append("    SA_CHUNK_CALL(%d, %s(%s));" % (c, cname, call_args))
    parts.append("\n".join(["SA_TEMPLATE SA_FN int %s(%s) {" % (name, signature), "    int bad = 0;"]
                           + calls + ["    return bad;", "}"]))
    return "\n".join(parts)


def generate_problem_source(
    *,
    n_states: int,
    n_sub: int,
    n_rem: int,
    symbol_map: Dict[str, str],
    dydt: np.ndarray,
    jac: np.ndarray,
    dlamdadt: np.ndarray,
    quad: np.ndarray,
    dydp_t: Optional[np.ndarray] = None,
    description: str = "",
    matvec: Optional[Dict[str, Dict[str, object]]] = None,
) -> str:
    """Full generated header: sizes + helpers + the five callbacks of the adjoint path and the
    parameter derivative of the right-hand side (forward sensitivities)."""
    n = n_states
    matvec = matvec or {}

    def mv(tag):
        return dict(matvec[tag], tag=tag) if tag in matvec else None
    # column-major slot of J[i, j] is j*n + i (reference problem.py:345,377 numba.farray)
    col_major = [j * n + i for i in range(n) for j in range(n)]
    jac = np.asarray(jac, dtype=object).reshape(n, n) if n else np.zeros((0, 0), object)
    adj_jac = np.array([[-jac[j, i] for j in range(n)] for i in range(n)], dtype=object).reshape(n, n)
    # no __restrict__ on purpose: with it the compiler hoists every load of a 10^4-statement callback
    # above the stores and spills tens of KB per lane; possible aliasing keeps live ranges per statement
    base = "double t, const double* y, const double* ps, const double* pr, SA_OUT_T out"
    adj = "double t, const double* y, const double* lam, const double* ps, const double* pr, SA_OUT_T out"
    parts = [
        "/* generated by sunode_amd.symode.codegen -- do not edit */",
        "/* %s */" % description.replace("*/", "* /"),
        "#define SA_N_STATES %d" % n_states,
        "#define SA_N_SUB %d" % n_sub,
        "#define SA_N_REM %d" % n_rem,
        HELPERS_C,
        emit_function("sa_rhs", base, np.asarray(dydt, dtype=object).ravel(),
                      list(range(n)), n, symbol_map, "r_", matvec=mv("f")),
        emit_function("sa_jac", base, jac, col_major, n * n, symbol_map, "j_"),
        emit_function("sa_adj_rhs", adj, np.asarray(dlamdadt, dtype=object).ravel(),
                      list(range(n)), n, symbol_map, "a_", matvec=mv("a")).replace(
                          "(void)pr;", "(void)pr; (void)lam;"),
        emit_function("sa_quad_rhs", adj, np.asarray(quad, dtype=object).ravel(),
                      list(range(n_sub)), n_sub, symbol_map, "q_").replace(
                          "(void)pr;", "(void)pr; (void)lam;"),
        emit_function("sa_adj_jac", base, adj_jac, col_major, n * n, symbol_map, "b_"),
        # d f / d p, stored [n_sub][n_states] (row `is` = derivative w.r.t. differentiated parameter `is`):
        # the explicit part of the sensitivity right-hand side yS' = J yS + df/dp (reference
        # symode/problem.py:557-583)
        emit_function("sa_dydp", base,
                      np.asarray(dydp_t if dydp_t is not None else np.zeros((n_sub, n), dtype=object),
                                 dtype=object).ravel(),
                      list(range(n_sub * n)), n_sub * n, symbol_map, "s_"),
        "",
    ]
    return "\n".join(parts)


def source_hash(text: str, extra: Iterable[str] = ()) -> str:
    h = hashlib.sha256(text.encode())
    for item in extra:
        h.update(b"\0" + item.encode())
    return h.hexdigest()[:20]
