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
  (``lambdify.py:59-77``; the sympy classes are in ``symode/lambdify.py``);
* non-finite outputs make the callback return 1 = "recoverable error"
  (``symode/problem.py:266-269``).

Different from the reference on purpose: integer powers are expanded to
products, no ``fastmath`` style re-association is allowed, and ``exp / log /
sin / cos / tan / tanh / sinh / cosh / log1p / expm1 / pow`` are printed as
calls of the deterministic implementations of ``csrc/sa_math.h`` (embedded
into the generated header when used) instead of libm / ocml, so that host and
device evaluate bit-identical arithmetic (both sides are compiled with
``-ffp-contract=off``); this is what makes the step/order bookkeeping of the
HIP integrator comparable bit-for-bit with the CPU oracle -- for rational AND
transcendental right-hand sides.  Functions outside that list (``LIBM_ONLY``)
still compile, through libm / ocml, without the bit-equality guarantee.

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
/* Matrix callbacks with structure (symode/problem.py extract_matfill): entry(slot) = M[slot] + u[line(slot)] (+ an
   exception term for few slots), M = the block of the remainder vector at slot OFF, u = N line values set by
   SA_UVEC_SET, line(slot) = slot % N (AXIS 0: the row of a column-major matrix) or slot / N (AXIS 1: the column).
   SA_MATFILL stores all N*N entries (SA_STORE_DYN: run-time slot) and folds their finiteness into chk;
   SA_MF(tag, slot, line, OFF) is the filled value of one slot (exceptions add their term to it). */
#ifndef SA_MATFILL
#define SA_UVEC_BEGIN(tag, N) double sa_uv_##tag[N];
#define SA_UVEC_SET(tag, k, v) sa_uv_##tag[k] = (v)
#define SA_UVEC(tag, k) sa_uv_##tag[k]
#define SA_MATFILL(tag, N, OFF, AXIS) \
    for (int s_ = 0; s_ < (N) * (N); s_++) { \
        const double f_ = SA_PR((OFF) + s_) + sa_uv_##tag[(AXIS) ? s_ / (N) : s_ % (N)]; \
        SA_STORE_DYN(s_, f_); chk += f_ * 0.0; \
    }
#endif
#ifndef SA_STORE_DYN
#define SA_STORE_DYN(slot, value) out[slot] = (value)
#endif
#ifndef SA_ZERO_RANGE        /* N consecutive structurally zero output slots from S0 on (codegen.ZERO_RUN_MIN) */
#define SA_ZERO_RANGE(S0, N) for (int z_ = 0; z_ < (N); z_++) SA_STORE_DYN((S0) + z_, 0.0)
#endif
#ifndef SA_UVEC_ROLLED       /* the N line values from one expression in i_ */
#define SA_UVEC_ROLLED(tag, N, expr) \
    for (int i_ = 0; i_ < (N); i_++) { const double v_ = (expr); SA_UVEC_SET(tag, i_, v_); chk += v_ * 0.0; }
#endif
#define SA_MF(tag, slot, line, OFF) (SA_PR((OFF) + (slot)) + SA_UVEC(tag, line))
/* Re-rolled loops (codegen.Roller): sympy hands over fully unrolled expressions; N isomorphic terms of a sum, or N
   isomorphic output statements, that differ only by a unit-stride index are emitted ONCE with a loop index.
     SA_SUM(N, term)       sum over j_ = 0..N-1 of `term` (an expression in j_), associated as the balanced binary
                           tree over next_pow2(N) leaves in index order (zeros beyond N): the association of a
                           cross-lane butterfly, so a kernel may evaluate one term per lane;
     SA_ROLLED(N, S0, S1, expr)   for i_ = 0..N-1: slot S0 + S1*i_ <- `expr` (an expression in i_), with the usual
                           finiteness check folded into chk.
   Defaults: plain loops (oracle, thread-per-instance kernels). */
#ifndef SA_SUM
#define SA_SUM(N, term) __extension__({ \
    double s_[(N) <= 1 ? 1 : (N) <= 2 ? 2 : (N) <= 4 ? 4 : (N) <= 8 ? 8 : (N) <= 16 ? 16 : (N) <= 32 ? 32 : (N) <= 64 ? 64 : \
              (N) <= 128 ? 128 : (N) <= 256 ? 256 : (N) <= 512 ? 512 : 1024]; \
    const int p_ = (int)(sizeof(s_) / sizeof(s_[0])); \
    for (int j_ = 0; j_ < p_; j_++) s_[j_] = 0.0; \
    for (int j_ = 0; j_ < (N); j_++) s_[j_] = (term); \
    for (int w_ = 1; w_ < p_; w_ *= 2) for (int q_ = 0; q_ < p_; q_ += 2 * w_) s_[q_] = s_[q_] + s_[q_ + w_]; \
    s_[0]; })
#endif
#ifndef SA_ROLLED
#define SA_ROLLED(N, S0, S1, expr) \
    for (int i_ = 0; i_ < (N); i_++) { const double v_ = (expr); SA_STORE_DYN((S0) + (S1) * i_, v_); chk += v_ * 0.0; }
#endif
/* Lane families (codegen.find_lane_families): a model written as loops over M groups -- compartments x age groups,
   species x sites -- is equivariant under a relabelling of the groups, so the outputs of group f are the outputs of
   group 0 with every group index j replaced by SA_TAU(j), the transposition (0 f).  Such outputs are emitted ONCE,
   for group 0, inside
     SA_FAM_BEGIN(M) ... SA_FAM_STORE(S0, expr); ... SA_FAM_END
   where SA_F is the member index, SA_TAU(j) its relabelling of a group coordinate j and slot S0 + SA_F receives `expr`.
   Default: a plain loop over the M members (oracle, one-lane mappings through a compile-time loop); a kernel with M
   lanes per instance evaluates ONE member per lane (bdf_wave.hip) instead of all outputs in every lane.  The
   expression text -- operand order included -- is the same for all members and all mappings. */
#ifndef SA_FAM_BEGIN
#define SA_FAM_BEGIN(M) for (int sa_f = 0; sa_f < (M); sa_f++) {
#define SA_F sa_f
#define SA_TAU(j) ((j) == 0 ? sa_f : ((j) == sa_f ? 0 : (j)))
#define SA_FAM_STORE(S0, value) { const double v_ = (value); SA_STORE_DYN((S0) + sa_f, v_); chk += v_ * 0.0; }
#define SA_FAM_END }
#endif
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
        return "sa_pow(%s, %s)" % (self._print(base), self._print(exp))

    # transcendental functions: the deterministic implementations of csrc/sa_math.h (embedded into the generated
    # header, MATH_C below), never libm / ocml -- device and oracle must round identically
    def _sa_call(self, name, expr):
        return "sa_%s(%s)" % (name, ", ".join(self._print(a) for a in expr.args))

    def _print_exp(self, expr):
        return self._sa_call("exp", expr)

    def _print_log(self, expr):
        if len(expr.args) == 2:      # log(x, base)
            return "(sa_log(%s)/sa_log(%s))" % (self._print(expr.args[0]), self._print(expr.args[1]))
        return self._sa_call("log", expr)

    def _print_log1p(self, expr):
        return self._sa_call("log1p", expr)

    def _print_expm1(self, expr):
        return self._sa_call("expm1", expr)

    def _print_sin(self, expr):
        return self._sa_call("sin", expr)

    def _print_cos(self, expr):
        return self._sa_call("cos", expr)

    def _print_tan(self, expr):
        return self._sa_call("tan", expr)

    def _print_tanh(self, expr):
        return self._sa_call("tanh", expr)

    def _print_sinh(self, expr):
        return self._sa_call("sinh", expr)

    def _print_cosh(self, expr):
        return self._sa_call("cosh", expr)

    # helper functions of the reference (lambdify.py:59-77, 275-340)
    def _print_logaddexp(self, expr):
        return "sa_logaddexp(%s, %s)" % tuple(self._print(a) for a in expr.args)

    def _print_expit(self, expr):
        return "sa_expit(%s)" % self._print(expr.args[0])

    def _print_dexpit(self, expr):
        return "sa_dexpit(%s)" % self._print(expr.args[0])

    def _print_CardinalBSpline(self, expr):
        degree, x = expr.args
        if degree != 4:             # (the reference's helper returns nan here; symode/lambdify.py CardinalBSpline)
            return "(%s)" % self._print(expr.as_sympy_expr())
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


#: sums / output families with at least this many isomorphic members are re-rolled
ROLL_MIN = 16
ROLL_MAX = 1024
_LEAF_RE = re.compile(r"^(SA_Y|SA_LAM|SA_PS|SA_PR)\((\d+)\)$")
_MV_RE = re.compile(r"^SA_MV\((\w+), (\d+)\)$")


class Roller:
    """Recovers loops from unrolled expressions: members of a family (terms of a long sum, or the outputs of a
    callback) are *isomorphic with unit stride* when replacing every indexed leaf ``ARR(k)`` of member j by the
    placeholder ``ARR(j + (k - j))`` makes them one and the same expression (leaves shared by ALL members stay
    as they are: loop invariants)."""

    def __init__(self, symbol_map: Dict[str, str], prefix: str):
        self.symbol_map = dict(symbol_map)
        self.prefix = prefix
        self.leaf: Dict[str, Tuple[str, int]] = {}
        for name, text in symbol_map.items():
            m = _LEAF_RE.match(text)
            if m:
                self.leaf[name] = (m.group(1), int(m.group(2)))
                continue
            m = _MV_RE.match(text)
            if m:
                self.leaf[name] = ("SA_MV:" + m.group(1), int(m.group(2)))
        self.sums: List[Tuple[sym.Symbol, int, sym.Expr]] = []      # (symbol, N, skeleton) in dependency order
        self._sum_of: Dict[Tuple[int, sym.Expr], sym.Symbol] = {}
        self._ph: Dict[Tuple[str, int, str], sym.Symbol] = {}
        self._memo: Dict[sym.Basic, sym.Basic] = {}

    def placeholder(self, arr: str, off: int, var: str) -> sym.Symbol:
        key = (arr, off, var)
        if key not in self._ph:
            s = sym.Symbol("sa_ph%d_%s" % (len(self._ph), var), real=True)
            self._ph[key] = s
            idx = var if off == 0 else "%s %s %d" % (var, "+" if off > 0 else "-", abs(off))
            self.symbol_map[s.name] = ("SA_MV(%s, %s)" % (arr[6:], idx)) if arr.startswith("SA_MV:") else "%s(%s)" % (arr, idx)
        return self._ph[key]

    def _leaves(self, e):
        # sorted: set iteration order must not leak into placeholder numbering (and from there, through sympy's
        # canonical argument order, into the association of the generated arithmetic)
        return {sy: self.leaf[sy.name] for sy in sorted(e.free_symbols, key=lambda q: q.name) if sy.name in self.leaf}

    def try_roll(self, members: Sequence[sym.Expr], var: str):
        """-> (skeleton, order) with members[order[j]] == skeleton at index j for j = 0..N-1, or None."""
        n = len(members)
        if n < ROLL_MIN or n > ROLL_MAX:
            return None
        leaves = [self._leaves(e) for e in members]
        common = set(leaves[0])
        for lv in leaves[1:]:
            common &= set(lv)
        variable = [{sy: al for sy, al in lv.items() if sy not in common} for lv in leaves]
        if any(not v for v in variable):
            return None

        def skeleton(k, j):
            return members[k].xreplace({sy: self.placeholder(arr, idx - j, var) for sy, (arr, idx) in variable[k].items()})

        cands = sorted({idx for (_, idx) in variable[0].values()})
        for j0 in cands:
            skel = skeleton(0, j0)
            slots = {ph: key for key, ph in self._ph.items() if ph in skel.free_symbols}
            offs = sorted({(arr, off) for (arr, off, _) in slots.values()})
            assign = {0: j0}
            ok = True
            for k in range(1, n):
                found = None
                tried = set()
                for (arr, idx) in variable[k].values():
                    for (arr2, off) in offs:
                        j = idx - off
                        if arr2 != arr or j in tried:
                            continue
                        tried.add(j)
                        if skeleton(k, j) == skel:
                            found = j
                            break
                    if found is not None:
                        break
                if found is None:
                    ok = False
                    break
                assign[k] = found
            if not ok:
                continue
            js = sorted(assign.values())
            if len(set(js)) != n or js[-1] - js[0] != n - 1:
                continue
            base = js[0]
            if base != 0:            # re-base the loop index to 0
                skel = skel.xreplace({ph: self.placeholder(key[0], key[1] + base, var)
                                      for ph, key in slots.items()})
            order = [None] * n
            for k, j in assign.items():
                order[j - base] = k
            return skel, order
        return None

    # -- sums --------------------------------------------------------------------------------------
    def roll_sums(self, e):
        """Replace every rollable long sum inside ``e`` (bottom-up) by a fresh scalar symbol."""
        e = sym.sympify(e)
        if e.is_Atom:
            return e
        if e in self._memo:
            return self._memo[e]
        args = [self.roll_sums(a) for a in e.args]
        new = e.func(*args) if any(a is not b for a, b in zip(args, e.args)) else e
        if new.is_Add and len(new.args) >= ROLL_MIN:
            new = self._roll_add(new)
        self._memo[e] = new
        return new

    def _roll_add(self, add):
        terms = list(sym.Add.make_args(add))
        buckets: Dict[Tuple, List[sym.Expr]] = {}
        for term in terms:
            key = (tuple(sorted(arr for arr, _ in self._leaves(term).values())), term.count_ops())
            buckets.setdefault(key, []).append(term)
        rest, out = [], []
        for key, group in buckets.items():
            got = self.try_roll(group, "j_") if key[0] else None
            if got is None:
                rest += group
                continue
            skel, _ = got
            # loop-invariant factors leave the sum: sum_j c*t_j is emitted as c * SUM(t_j) (our own definition of
            # the arithmetic: the oracle compiles the same source), which also lets different sums share one SUM
            ph = sorted((p_ for p_ in skel.free_symbols if p_ in set(self._ph.values())), key=lambda q: q.name)
            coeff, core = skel.as_independent(*ph, as_Add=False) if ph else (sym.Integer(1), skel)
            if core == 1 or not (core.free_symbols & set(ph)):
                coeff, core = sym.Integer(1), skel
            sign = -1 if core.could_extract_minus_sign() else 1
            core, coeff = sign * core, sign * coeff
            keyc = (len(group), core)
            if keyc not in self._sum_of:
                symbol = sym.Symbol("%ssum%d" % (self.prefix, len(self.sums)), real=True)
                self.sums.append((symbol, len(group), core))
                self._sum_of[keyc] = symbol
            out.append(coeff * self._sum_of[keyc])
        if not out:
            return add
        return sym.Add(*(out + rest))

    def split_invariants(self, skel, names):
        """Loop-invariant sub-expressions of a skeleton (everything that does not depend on the loop index) become
        assignments to emit in front of the loop; returns (assignments, skeleton over those temporaries)."""
        ph = set(self._ph.values())
        keep: List[Tuple[sym.Symbol, sym.Expr]] = []
        made: Dict[sym.Expr, sym.Symbol] = {}

        def temp_for(e):
            if e.is_Atom or (e.is_Mul and len(e.args) == 2 and e.args[0].is_Number and e.args[1].is_Atom):
                return e
            if e not in made:
                made[e] = next(names)
                keep.append((made[e], e))
            return made[e]

        def hoist(e):
            if e.is_Atom:
                return e
            if not (e.free_symbols & ph):
                return temp_for(e)
            if e.is_Mul or e.is_Add:
                inv = [a for a in e.args if not (a.free_symbols & ph)]
                var = [hoist(a) for a in e.args if a.free_symbols & ph]
                if len(inv) >= 2 or (len(inv) == 1 and not inv[0].is_Atom):
                    return e.func(temp_for(e.func(*inv)), *var)
                return e.func(*(inv + var))
            return e.func(*[hoist(a) for a in e.args])

        return keep, hoist(sym.sympify(skel))


#: group sizes tried for lane families (lanes per instance of the lean lane groups are 4 and 8)
FAMILY_SIZES = (4, 8, 2, 3, 5, 6, 7)


def find_lane_families(flat, out_index, n_out, symbol_map, leaf_axes, out_array):
    """Group structure of a vector callback -> (M, [slot0, ...], relabelled symbol map) or None.

    ``flat[k]`` is the expression of output slot ``out_index[k]``; the outputs are laid out like the array
    ``out_array`` ("SA_Y": like the state; "SA_PS": like the differentiated parameters).  A FAMILY is a 1-D leaf of
    length M of that array whose member f equals member 0 with all group coordinates relabelled by the transposition
    (0 f) -- on every axis of length M of every state / adjoint-state / parameter leaf at once.  Checked symbolically
    (``xreplace`` + sympy's canonical form), so a symbol may play the "own group" and the "summed over all groups" role
    in the same expression.  Expressions with generated symbols (hoisted values, matrix-vector results) are left alone."""
    if leaf_axes is None or n_out < 2 or list(out_index) != list(range(n_out)):
        return None
    by_slot = {(arr, idx): (name, axes) for name, (arr, idx, axes) in leaf_axes.items()}
    known = set(leaf_axes)
    free = set().union(*[e.free_symbols for e in flat]) if flat else set()
    if any(sy.name not in known and sy.name != "time" for sy in free):
        return None
    symbols = {sy.name: sy for sy in free}
    for M in FAMILY_SIZES:
        leaves = []                # (slot0 of a 1-D output leaf of length M)
        for (arr, idx), (name, axes) in by_slot.items():
            if arr == out_array and len(axes) == 1 and axes[0][2] == M and axes[0][1] == 0 and idx + M <= n_out:
                leaves.append(idx)
        if not leaves:
            continue

        def relabel(f):
            sub = {}
            for name, sy in symbols.items():
                if name == "time":
                    continue
                arr, idx, axes = leaf_axes[name]
                new = idx
                for stride, coord, length in axes:
                    if length == M:
                        t = f if coord == 0 else (0 if coord == f else coord)
                        new += stride * (t - coord)
                if new != idx:
                    other = by_slot.get((arr, new))
                    if other is None:
                        return None
                    sub[sy] = sym.Symbol(other[0], **sy.assumptions0)
            return sub
        subs = [relabel(f) for f in range(M)]
        if any(sub is None for sub in subs):
            continue
        good = [s0 for s0 in sorted(leaves)
                if all(sym.sympify(flat[s0]).xreplace(subs[f]) == sym.sympify(flat[s0 + f]) for f in range(1, M))
                and any(sym.sympify(flat[s0 + f]) != 0 for f in range(M))]
        if not good:
            continue
        fam_map = dict(symbol_map)
        for name, (arr, idx, axes) in leaf_axes.items():
            grouped = [(stride, coord) for stride, coord, length in axes if length == M]
            if grouped:
                base = idx - sum(stride * coord for stride, coord in grouped)
                terms = ["%d" % base] if base else []
                terms += [("SA_TAU(%d)" % coord) if stride == 1 else "%d*SA_TAU(%d)" % (stride, coord)
                          for stride, coord in grouped]
                fam_map[name] = "%s(%s)" % (arr, " + ".join(terms))
        return M, good, fam_map
    return None


def _emit_families(name, signature, flat, n_out, symbol_map, names, fam):
    """Callback with lane families: the outputs outside the families as ordinary statements, then every family
    once (member 0) inside SA_FAM_BEGIN .. SA_FAM_END."""
    M, starts, fam_map = fam
    in_family = {s0 + f for s0 in starts for f in range(M)}
    rest_slots = [k for k in range(n_out) if k not in in_family]
    printer = HipExprPrinter(symbol_map)
    fprinter = HipExprPrinter(fam_map)
    lines = ["SA_TEMPLATE SA_FN int %s(%s) {" % (name, signature), "    SA_PROLOGUE", "    double chk = 0.0;"]
    if rest_slots:
        assigns, reduced = sym.cse([flat[k] for k in rest_slots], symbols=names, order="canonical")
        lines += ["    const double %s = %s; SA_STMT_END" % (v.name, printer.doprint(val)) for v, val in assigns]
        for k, value in zip(rest_slots, reduced):
            if value == 0:
                lines.append("    SA_STORE(%d, 0.0);" % k)
            else:
                lines.append("    { const double v_ = %s; SA_STORE(%d, v_); chk += v_ * 0.0; } SA_STMT_END"
                             % (printer.doprint(value), k))
    assigns, reduced = sym.cse([flat[s0] for s0 in starts], symbols=names, order="canonical")
    lines.append("    SA_FAM_BEGIN(%d)" % M)
    lines += ["        const double %s = %s; SA_STMT_END" % (v.name, fprinter.doprint(val)) for v, val in assigns]
    for s0, value in zip(starts, reduced):
        lines.append("        SA_FAM_STORE(%d, %s); SA_STMT_END" % (s0, fprinter.doprint(value)))
    lines.append("    SA_FAM_END")
    lines.append("    (void)t; (void)y; (void)ps; (void)pr; (void)out;")
    lines.append("    SA_EPILOGUE")
    lines.append("    return (chk == 0.0) ? 0 : 1;")
    lines.append("}")
    return "\n".join(lines)


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


def _emit_rolled(name, signature, rolled, out_index, n_out, roller, names, matvec, fam):
    """Body of a callback whose sums (SA_SUM) and / or outputs (SA_ROLLED) were re-rolled."""
    printer = HipExprPrinter(roller.symbol_map)
    lines = ["SA_TEMPLATE SA_FN int %s(%s) {" % (name, signature), "    SA_PROLOGUE"]
    if matvec is not None:
        lines.append("    SA_MATVEC(%s, %d, %d, %d, %s);" % (matvec["tag"], matvec["n_out"], matvec["n_in"],
                                                            matvec["offset"], matvec["vec"]))
    lines.append("    double chk = 0.0;")
    for symbol, n_terms, skel in roller.sums:
        keep, body = roller.split_invariants(skel, names)
        lines += ["    const double %s = %s; SA_STMT_END" % (v.name, printer.doprint(val)) for v, val in keep]
        lines.append("    const double %s = SA_SUM(%d, %s); SA_STMT_END" % (symbol.name, n_terms, printer.doprint(body)))
    if fam is not None:
        keep, body = roller.split_invariants(fam[0], names)
        lines += ["    const double %s = %s; SA_STMT_END" % (v.name, printer.doprint(val)) for v, val in keep]
        lines.append("    SA_ROLLED(%d, 0, 1, %s); SA_STMT_END" % (n_out, printer.doprint(body)))
    else:
        assigns, reduced = sym.cse(rolled, symbols=names, order="canonical") if rolled else ([], [])
        lines += ["    const double %s = %s; SA_STMT_END" % (v.name, printer.doprint(val)) for v, val in assigns]
        for k, value in enumerate(reduced):
            slot = int(out_index[k])
            if value == 0:
                lines.append("    SA_STORE(%d, 0.0);" % slot)
            else:
                lines.append("    if (SA_OWNS(%d)) { const double v_ = %s; SA_STORE(%d, v_); chk += v_ * 0.0; } SA_STMT_END"
                             % (slot, printer.doprint(value), slot))
    lines.append("    (void)t; (void)y; (void)ps; (void)pr; (void)out;")
    lines.append("    SA_EPILOGUE")
    lines.append("    return (chk == 0.0) ? 0 : 1;")
    lines.append("}")
    return "\n".join(lines)


#: runs of at least this many consecutive structurally zero output slots are emitted as ONE SA_ZERO_RANGE loop (a
#: banded 512 x 512 Jacobian is 260 000 zero stores otherwise: 15 MB of generated text); shorter runs -- every model of
#: up to 8 x 8 -- keep the literal stores, which a register mapping folds away
ZERO_RUN_MIN = 64


def _zero_runs(slots: Sequence[int], written: Dict[int, str]) -> Dict[int, int]:
    """{position in `slots`: run length at the first slot of a long zero run, 0 at its other slots}."""
    out: Dict[int, int] = {}
    k = 0
    while k < len(slots):
        e = k
        while e < len(slots) and written.get(slots[e], "0.0") == "0.0" and (e == k or slots[e] == slots[e - 1] + 1):
            e += 1
        if e - k >= ZERO_RUN_MIN:
            out[k] = e - k
            for j in range(k + 1, e):
                out[j] = 0
        k = max(e, k + 1)
    return out


def _temp_names(prefix: str, symbol_map: Dict[str, str]):
    """Names of the CSE temporaries of one callback: ``<prefix><i>``, skipping every name a model symbol already has
    (a parameter ``a`` of shape (5,) owns ``a_0 .. a_4``: the reference refuses such collisions,
    lambdify.py:114-116; here the temporaries step aside)."""
    for i in count():
        name = "%s%d" % (prefix, i)
        if name not in symbol_map:
            yield sym.Symbol(name)


def emit_function(
    name: str,
    signature: str,
    expr: np.ndarray,
    out_index: Sequence[int],
    n_out: int,
    symbol_map: Dict[str, str],
    prefix: str,
    matvec: Optional[Dict[str, object]] = None,
    leaf_axes=None,
    out_array: Optional[str] = None,
) -> str:
    """One callback.  ``expr`` is ravelled; ``out_index[k]`` is the flat output
    slot of ``expr.ravel()[k]`` (lets the caller pick column-major storage).
    ``matvec`` (tag, n_out, n_in, offset, vec): the expressions refer to ``SA_MV(tag, i)``, the results of one
    dense matrix-vector product evaluated up front by ``SA_MATVEC``; the function is then emitted as ONE body whose
    output statements carry ``SA_OWNS(slot)`` guards (the kernels split those between wavefronts)."""
    flat = [sym.sympify(e) for e in np.asarray(expr, dtype=object).ravel()]
    names = _temp_names(prefix, symbol_map)
    # group structure: one member per lane (SA_FAM_BEGIN); small callbacks only -- the large ones have their own forms
    if out_array is not None and matvec is None and n_out <= 64:
        fam = find_lane_families(flat, out_index, n_out, symbol_map, leaf_axes, out_array)
        if fam is not None:
            return _emit_families(name, signature, flat, n_out, symbol_map, names, fam)
    # re-rolling (Roller): long sums of isomorphic terms -> SA_SUM, isomorphic outputs -> SA_ROLLED
    if len(flat) * max((len(sym.Add.make_args(e)) for e in flat), default=0) >= ROLL_MIN and any(
            len(sym.Add.make_args(a)) >= ROLL_MIN for e in flat for a in sym.preorder_traversal(e) if a.is_Add) \
            or len(flat) >= ROLL_MIN:
        roller = Roller(symbol_map, prefix)
        rolled = [roller.roll_sums(e) for e in flat]
        identity = list(out_index) == list(range(n_out))
        fam = roller.try_roll(rolled, "i_") if identity else None
        if fam is not None and fam[1] != list(range(n_out)):
            fam = None
        if roller.sums or fam is not None:
            return _emit_rolled(name, signature, rolled, out_index, n_out, roller, names, matvec, fam)
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
    # `out[slot] = value`; the lane-group kernels (a few state components per lane) keep only the
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
        slots = list(slots)
        zero_run = _zero_runs(slots, written)
        for k, slot in enumerate(slots):
            text = written.get(slot, "0.0")
            if text == "0.0":
                if k in zero_run:                      # a long run of structural zeros: one loop, not N statements
                    if zero_run[k]:
                        lines.append("    SA_ZERO_RANGE(%d, %d);" % (slot, zero_run[k]))
                    continue
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
    # statements per slot: the slots of a long zero run are ONE statement together (SA_ZERO_RANGE)
    runs = _zero_runs(list(range(n_out)), written)
    stmt = [0 if runs.get(slot, 1) == 0 else 1 for slot in range(n_out)]
    for slot in range(n_out):
        if not stmt[slot]:
            cost[slot] = 0
    total = sum(cost)
    n_chunks = -(-sum(stmt) // CHUNK_STATEMENTS)
    if total > CHUNK_COST:
        n_chunks = max(n_chunks, min(MAX_COST_CHUNKS, -(-total // CHUNK_COST), n_out))
    if n_chunks <= 1 or matvec is not None:
        lines = ["SA_TEMPLATE SA_FN int %s(%s) {" % (name, signature)]
        lines += body(range(n_out), temp_order)
        lines.append("}")
        return "\n".join(lines)

    # chunked form: same expressions, same evaluation order inside every statement; a temporary
    # needed by several chunks is recomputed in each (identical value)
    bounds, acc, target, in_chunk = [0], 0, total / n_chunks, 0
    for slot in range(n_out):
        acc += cost[slot]
        in_chunk += stmt[slot]
        full = in_chunk >= CHUNK_STATEMENTS
        if (acc >= target * len(bounds) or full) and slot + 1 < n_out and len(bounds) < n_chunks:
            bounds.append(slot + 1)
            in_chunk = 0
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


def emit_matfill_function(name: str, signature: str, n: int, info: Dict[str, object], symbol_map: Dict[str, str],
                          prefix: str, tag: str) -> str:
    """Matrix callback in the structured form (see HELPERS_C, SA_MATFILL): N line-vector statements, one fill,
    then one statement per exception slot."""
    axis, u, exceptions, offset = info["axis"], info["u"], info["exceptions"], info["offset"]
    keys = sorted(exceptions)
    names = _temp_names(prefix, symbol_map)
    roller = Roller(symbol_map, prefix)
    u_r = [roller.roll_sums(e) for e in u]
    x_r = [roller.roll_sums(exceptions[k]) for k in keys]
    printer = HipExprPrinter(roller.symbol_map)
    lines = ["SA_TEMPLATE SA_FN int %s(%s) {" % (name, signature), "    SA_PROLOGUE", "    double chk = 0.0;"]
    for symbol, n_terms, skel in roller.sums:
        keep, body = roller.split_invariants(skel, names)
        lines += ["    const double %s = %s; SA_STMT_END" % (v.name, printer.doprint(val)) for v, val in keep]
        lines.append("    const double %s = SA_SUM(%d, %s); SA_STMT_END" % (symbol.name, n_terms, printer.doprint(body)))
    lines.append("    SA_UVEC_BEGIN(%s, %d)" % (tag, n))
    fam_u = roller.try_roll(u_r, "i_")
    if fam_u is not None and fam_u[1] != list(range(n)):
        fam_u = None
    # the exceptions as a family: entries (i, i) of the diagonal, slot (n + 1) * i, line i
    diag = keys == [(i, i) for i in range(n)]
    fam_x = roller.try_roll(x_r, "i_") if diag else None
    if fam_x is not None and fam_x[1] != list(range(n)):
        fam_x = None
    rest = ([] if fam_u is not None else u_r) + ([] if fam_x is not None else x_r)
    assigns, reduced = sym.cse(rest, symbols=names, order="canonical") if rest else ([], [])
    lines += ["    const double %s = %s; SA_STMT_END" % (v.name, printer.doprint(val)) for v, val in assigns]
    if fam_u is not None:
        keep, body = roller.split_invariants(fam_u[0], names)
        lines += ["    const double %s = %s; SA_STMT_END" % (v.name, printer.doprint(val)) for v, val in keep]
        lines.append("    SA_UVEC_ROLLED(%s, %d, %s); SA_STMT_END" % (tag, n, printer.doprint(body)))
        red_x = reduced
    else:
        for k in range(n):
            lines.append("    if (SA_OWNS(%d)) { const double v_ = %s; SA_UVEC_SET(%s, %d, v_); chk += v_ * 0.0; } SA_STMT_END"
                         % (k, printer.doprint(reduced[k]), tag, k))
        red_x = reduced[n:]
    lines.append("    SA_MATFILL(%s, %d, %d, %d);" % (tag, n, offset, axis))
    if fam_x is not None:
        keep, body = roller.split_invariants(fam_x[0], names)
        lines += ["    const double %s = %s; SA_STMT_END" % (v.name, printer.doprint(val)) for v, val in keep]
        lines.append("    SA_ROLLED(%d, 0, %d, SA_MF(%s, %d * i_, i_, %d) + (%s)); SA_STMT_END"
                     % (n, n + 1, tag, n + 1, offset, printer.doprint(body)))
    else:
        for q, (i, j) in enumerate(keys):
            slot = j * n + i
            line = i if axis == 0 else j
            lines.append("    if (SA_OWNS(%d)) { const double v_ = SA_MF(%s, %d, %d, %d) + (%s); SA_STORE(%d, v_); "
                         "chk += v_ * 0.0; } SA_STMT_END" % (q, tag, slot, line, offset, printer.doprint(red_x[q]), slot))
    lines.append("    (void)t; (void)y; (void)ps; (void)pr; (void)out;")
    lines.append("    SA_EPILOGUE")
    lines.append("    return (chk == 0.0) ? 0 : 1;")
    lines.append("}")
    return "\n".join(lines)


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
    matfill: Optional[Dict[str, Dict[str, object]]] = None,
    leaf_axes=None,
) -> str:
    """Full generated header: sizes + helpers + the five callbacks of the adjoint path and the
    parameter derivative of the right-hand side (forward sensitivities)."""
    n = n_states
    matvec = matvec or {}
    matfill = matfill or {}

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
        None,           # MATH_C, when a callback calls one of its functions
        emit_function("sa_rhs", base, np.asarray(dydt, dtype=object).ravel(),
                      list(range(n)), n, symbol_map, "r_", matvec=mv("f"), leaf_axes=leaf_axes, out_array="SA_Y"),
        (emit_matfill_function("sa_jac", base, n, matfill["j"], symbol_map, "j_", "j") if "j" in matfill else
         emit_function("sa_jac", base, jac, col_major, n * n, symbol_map, "j_")),
        emit_function("sa_adj_rhs", adj, np.asarray(dlamdadt, dtype=object).ravel(),
                      list(range(n)), n, symbol_map, "a_", matvec=mv("a"), leaf_axes=leaf_axes, out_array="SA_Y").replace(
                          "(void)pr;", "(void)pr; (void)lam;"),
        emit_function("sa_quad_rhs", adj, np.asarray(quad, dtype=object).ravel(),
                      list(range(n_sub)), n_sub, symbol_map, "q_", leaf_axes=leaf_axes, out_array="SA_PS").replace(
                          "(void)pr;", "(void)pr; (void)lam;"),
        (emit_matfill_function("sa_adj_jac", base, n, matfill["b"], symbol_map, "b_", "b") if "b" in matfill else
         emit_function("sa_adj_jac", base, adj_jac, col_major, n * n, symbol_map, "b_")),
        # d f / d p, stored [n_sub][n_states] (row `is` = derivative w.r.t. differentiated parameter `is`):
        # the explicit part of the sensitivity right-hand side yS' = J yS + df/dp (reference
        # symode/problem.py:557-583)
        emit_function("sa_dydp", base,
                      np.asarray(dydp_t if dydp_t is not None else np.zeros((n_sub, n), dtype=object),
                                 dtype=object).ravel(),
                      list(range(n_sub * n)), n_sub * n, symbol_map, "s_"),
        "",
    ]
    uses_math = any(_MATH_CALL.search(part) for part in parts if part)
    parts[6] = math_c() if uses_math else "/* (no transcendental function: csrc/sa_math.h not embedded) */"
    return "\n".join(parts)


#: a call of one of the deterministic functions of csrc/sa_math.h in generated text
_MATH_CALL = re.compile(r"\bsa_(exp|expm1|log|log1p|sin|cos|tan|tanh|sinh|cosh|pow|logaddexp|expit|dexpit|"
                        r"cardinal_bspline4)\(")
#: functions the C99 printer would hand to libm / ocml (not bit-reproducible between host and device)
LIBM_ONLY = ("asin", "acos", "atan", "atan2", "asinh", "acosh", "atanh", "erf", "erfc", "tgamma", "lgamma", "cbrt",
             "exp2", "log2", "log10", "hypot")


def math_c() -> str:
    """Text of csrc/sa_math.h (the deterministic exp / log / sin / pow ... shared by device and oracle)."""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(os.path.dirname(here), "csrc", "sa_math.h")) as fh:
        return fh.read()


def libm_calls(source: str) -> List[str]:
    """Names of libm-only functions a generated header calls: such a model integrates, but the device (ocml) and
    the host (libm) may round them differently -- bit-equality with the oracle is then not guaranteed."""
    return sorted({m for m in LIBM_ONLY if re.search(r"(?<![\w.])%s\(" % m, source)})


def source_hash(text: str, extra: Iterable[str] = ()) -> str:
    h = hashlib.sha256(text.encode())
    for item in extra:
        h.update(b"\0" + item.encode())
    return h.hexdigest()[:20]
