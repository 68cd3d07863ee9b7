/*
 * bdf_mem.hip -- memory-resident thread-per-instance MAPPING for large systems (n > 128) and for forward
 * sensitivities beyond the lane groups.
 *
 * One lane = one integrator as in bdf_kernels.hip, but the vectors, the Nordsieck array, the Newton
 * matrix / saved Jacobian and the LU live in an HBM workspace laid out [field element][instance]
 * (instance index fastest), so that the 64 lanes of a wave -- which execute the same loop over
 * the components in lockstep -- always touch 64 consecutive doubles: every access is one fully
 * coalesced 512-byte transaction served by L1/L2.  Control scalars (h, q, tau, l, tq, counters) stay
 * in registers.  Loops over components are real loops (code size independent of n); the generated
 * callbacks read the state through SA_Y / SA_LAM and write through SA_STORE with the workspace
 * stride, and run with full lane utilisation (each lane evaluates its own instance).  The dense LU
 * is the serial denseGETRF per lane; its n^3/3 multiply-adds stream through L2.
 *
 * Round 5: this file is a MAPPING of csrc/bdf_core.h like bdf_kernels.hip and bdf_wave.hip -- the
 * controller (cvHin, Nordsieck updates, Newton loop, error tests, order selection, sensitivity
 * correctors, cv_attempt) is that one header; here a vector field of the state is a strided VIEW of
 * the workspace (m.zn[j][r], m.acor[r], ... index HBM), the loops over vector slots are run-time loops
 * (VFOR / QFOR), and the hooks below are the norms, the callbacks, the dense LU and the interpolation.
 * Until round 4 the file carried its own loop-based restatement of the controller (1 805 lines).
 *
 * Same algorithm, operation order and rounding as the CPU oracle (restated CVODES 5.x; reference call
 * sites /root/reference/sunode/solver.py:467-527, 682-784): explicit FMAs, reciprocal pivots,
 * balanced-tree WRMS sums, deterministic pow.  Entry points / argument blocks as in bdf_kernels.hip;
 * the host library allocates the workspace (sa_meta[5] doubles per instance).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

/* the generated callbacks are real functions here (one copy each, called with the sink by value):
   for n = 100 the Jacobian alone is 10^4 statements */
#define SA_FN static __device__ __attribute__((noinline))
#define SA_TEMPLATE template <class SinkT>
#define SA_OUT_T SinkT
#define SA_STORE(slot, value) out.template put<(slot)>(value)
#define SA_STORE_DYN(slot, value) out.put_dyn(slot, value)
#define SA_Y(i) y[(int64_t)(i) * sa_ystride]
#define SA_LAM(i) lam[(int64_t)(i) * sa_ystride]
/* the state views handed to the generated functions have the stride of their output sink */
#define sa_ystride (out.stride)
#include SA_PROBLEM_HEADER
#undef sa_ystride
#include "sa_device_abi.h"
/* bdf_core.h's loops over the slots of a vector: real loops over the n (p) components */
#define VFOR(r) _Pragma("nounroll") for (int r = 0; r < NS; r++) {
#define VEND }
#define QFOR(r) _Pragma("nounroll") for (int r = 0; r < NQ; r++) {
#define QEND }
#include "sa_common.h"

#define TREC (8 + 6 * NS)
#define SA_NAN __builtin_bit_cast(double, (uint64_t)0x7ff8000000000000ULL)
constexpr int next_pow2_c(int n) { int p = 1; while (p < n) p <<= 1; return p; }
#define PS_TREE next_pow2_c(NS > NQ ? (NS > 0 ? NS : 1) : (NQ > 0 ? NQ : 1))

/* workspace field offsets (in elements; element e of the lane's instance is w[e * S]) */
#define O_ZN 0
#define O_ZNQ (O_ZN + 6 * NS)
#define O_ZSAVE (O_ZNQ + 6 * NQ)
#define O_ZSAVEQ (O_ZSAVE + NS)
#define O_EWT (O_ZSAVEQ + NQ)
#define O_ACOR (O_EWT + NS)
#define O_TEMPV (O_ACOR + NS)
#define O_FTEMP (O_TEMPV + NS)
#define O_Y (O_FTEMP + NS)
#define O_YTMP (O_Y + NS)
#define O_EWTQ (O_YTMP + NS)
#define O_ACORQ (O_EWTQ + NQ)
#define O_TEMPVQ (O_ACORQ + NQ)
#define O_LAM (O_TEMPVQ + NQ)
#define O_QUAD (O_LAM + NS)
#define O_QOUT (O_QUAD + NQ)
#define O_A (O_QOUT + NQ)
#define O_SJ (O_A + NS * NS)
#define O_PIV (O_SJ + NS * NS)
#define O_INVP (O_PIV + NS)
#define O_HY (O_INVP + NS)
#define O_F0 (O_HY + 6 * NS)
#define O_TREE (O_F0 + NS)
/* the controller's vector temporaries (bdf_core.h TMPV / TMPQ: 20 state-sized + 5 quadrature-sized slots) */
#define O_TMP (O_TREE + PS_TREE)
#define O_TMPQ (O_TMP + 20 * NS)
#define O_TMP_END (O_TMPQ + 5 * NQ)
#ifdef SA_SENS      /* forward sensitivities: the SV_COUNT vectors of the NQ sensitivity systems (sa_common.h), J and df/dp */
#define O_SV O_TMP_END
#define O_DP (O_SV + SV_COUNT * NQ * NS)
#define O_JT (O_DP + NQ * NS)
#define WS_DOUBLES (O_JT + NS * NS)
#else
#define WS_DOUBLES O_TMP_END
#endif

/* strided sink for the generated callbacks: slot k of the output lives at p[k * stride] */
struct StrideSink {
    double *p;
    int64_t stride;
    template <int S> __device__ __forceinline__ void put(double x) const { p[(int64_t)S * stride] = x; }
    __device__ __forceinline__ void put_dyn(int slot, double x) const { p[(int64_t)slot * stride] = x; }
};

/* a vector of the instance: element r at p[r * S] (workspace: S = instance stride; caller arrays: S = 1) */
struct Vec {
    double *p;
    int64_t S;
    __device__ __forceinline__ double &operator[](int r) const { return p[(int64_t)r * S]; }
};
/* the Nordsieck arrays: column j = the vector starting n elements further on */
struct Cols {
    double *p;
    int64_t S;
    int n;
    __device__ __forceinline__ Vec operator[](int j) const { return Vec{p + (int64_t)j * n * S, S}; }
};
/* absolute tolerances: a per-component vector (forward problem) or one scalar (backward problem) */
struct Atol {
    const double *p;
    double s;
    __device__ __forceinline__ double operator[](int r) const { return p ? p[r] : s; }
};

template <bool BWD>
struct Cm {
    double *w;                 /* workspace, pre-offset to the lane's instance */
    int64_t S;                 /* instance stride of the workspace */
    Cols zn, znQ;
    Vec zsave, zsaveQ, ewt, acor, tempv, ftemp, y, ytmp, ewtQ, acorQ, tempvQ, f0;
    Atol atol;
    double rtol, rtolQ, atolQ;
    double tn, h, hprime, hscale, eta, etamax, hu;
    int q, qprime, L, qwait, qu;
    double tau[7], tq[6], l[7];
    double rl1, gamma, gammap, gamrat, crate, delp, acnrm, saved_tq5;
    double etaq, etaqm1, etaqp1, tstop;
    int nst, nfe, nje, nsetups, nni, ncfn, netf, nfQe, netfQ, nstlp, nstlj;
    int jcur, nls_jcur;
    double ps[NQD];
    const double *pr;          /* remaining parameters (global, unit stride) */
    const double *traj;        /* trajectory arena, pre-offset to the lane's instance */
    int64_t tS;                /* instance stride of the arena: field f of point s at traj[(s*TREC+f)*tS] */
    int np;
    double tfinal;
    int ilast, newdata, have_last, cur_idx;
    double last_t, tlo, thi, tlo2;
    int n_interp, n_rebuild;
    const double *cons;        /* CVodeSetConstraints vector (global) or nullptr */
    int constr;
    int lane;                  /* (unused: one lane = one instance) */
    /* forward sensitivities (only the SA_SENS build sets sensi) */
    int sensi, ism;
    double pbar[NQD], crateS, delpS, acnrmS;
    int nfSe, nniS, ncfnS, netfS, nsetupsS;
};

#define W(m, off, i) (m).w[(int64_t)((off) + (i)) * (m).S]
template <bool BWD>
DEV Vec wvec(const Cm<BWD> &m, int off) { return Vec{m.w + (int64_t)off * m.S, m.S}; }

/* point the vector fields of the state at the lane's workspace */
template <bool BWD>
DEV void bind_workspace(Cm<BWD> &m, double *ws, int64_t stride, int inst)
{
    m.S = stride;
    m.w = ws + inst;
    m.lane = 0;
    m.zn = Cols{m.w + (int64_t)O_ZN * m.S, m.S, NS};
    m.znQ = Cols{m.w + (int64_t)O_ZNQ * m.S, m.S, NQ};
    m.zsave = wvec(m, O_ZSAVE); m.zsaveQ = wvec(m, O_ZSAVEQ); m.ewt = wvec(m, O_EWT); m.acor = wvec(m, O_ACOR);
    m.tempv = wvec(m, O_TEMPV); m.ftemp = wvec(m, O_FTEMP); m.y = wvec(m, O_Y); m.ytmp = wvec(m, O_YTMP);
    m.ewtQ = wvec(m, O_EWTQ); m.acorQ = wvec(m, O_ACORQ); m.tempvQ = wvec(m, O_TEMPVQ); m.f0 = wvec(m, O_F0);
}

/* ---- the mapping bdf_core.h needs (see its header) ---- */
#define SA_STATE Cm
#define RS (NS > 0 ? NS : 1)
#define RQ NQD
#define IDX(m, r) (r)
#define wave_max(lane, x) (x)
#define COLD_STORE(m)
#define COLD_LOAD(m)
#define PH_T0
#define PH_ADD(m, k)
#define SA_POLY_CM(BWD) false
/* vector temporaries of the controller: slots of the workspace, not n-sized arrays in every lane's scratch */
#define TMPV(m, name, slot) const Vec name = wvec(m, O_TMP + (slot) * NS)
#define TMPQ(m, name, slot) const Vec name = wvec(m, O_TMPQ + (slot) * NQ)
#define TMPV2(m, name, rows, slot) const Cols name{(m).w + (int64_t)(O_TMP + (slot) * NS) * (m).S, (m).S, NS}
#ifdef SA_SENS
#define SENS_ON(m) (!BWD && (m).sensi)
#define SV(m, v, is, r) W(m, O_SV, (((v) * NQ + (is)) * NS) + (r))
#define SLOOP_BEGIN(is) _Pragma("nounroll") for (int is = 0; is < NQ; is++) {     /* (a loop: see bdf_wave.hip) */
#define SLOOP_END }
#endif

/* ---- norms: balanced tree over the zero-padded power-of-two array, as in the oracle ---- */
template <bool BWD, class X, class Wt>
DEV double wrms_gen(const Cm<BWD> &m, const X &x, const Wt &w, int n)
{
    if (n == 0) return 0.0;
    int P = 1;
    while (P < n) P <<= 1;
    for (int i = 0; i < P; i++) {
        double prod = (i < n) ? x[i] * w[i] : 0.0;
        W(m, O_TREE, i) = prod * prod;
    }
    for (int s = 1; s < P; s <<= 1)
        for (int i = 0; i < P; i += 2 * s) W(m, O_TREE, i) = W(m, O_TREE, i) + W(m, O_TREE, i + s);
    return sqrt(W(m, O_TREE, 0) / n);
}
template <bool BWD, class X, class Wt> DEV double wrms_n(const Cm<BWD> &m, const X &x, const Wt &w) { return wrms_gen(m, x, w, NS); }
template <bool BWD, class X, class Wt> DEV double wrms_q(const Cm<BWD> &m, const X &x, const Wt &w) { return wrms_gen(m, x, w, NQ); }
template <bool BWD, class X>
DEV double quad_update_norm(const Cm<BWD> &m, double old_nrm, const X &xQ)
{
    double qnrm = wrms_q(m, xQ, m.ewtQ);
    return old_nrm > qnrm ? old_nrm : qnrm;
}

template <bool BWD, class Y, class Wt>
DEV int ewt_set(const Cm<BWD> &m, const Y &ycur, Wt &&w)
{
    int bad = 0;
    for (int i = 0; i < NS; i++) {
        double v = FMA(m.rtol, fabs(ycur[i]), m.atol[i]);
        bad |= (v <= 0.0);
        w[i] = 1.0 / v;
    }
    return bad ? -1 : 0;
}

template <bool BWD, class Q, class Wt>
DEV int ewtQ_set(const Cm<BWD> &m, const Q &qcur, Wt &&w)
{
    int bad = 0;
    for (int i = 0; i < NQ; i++) {
        double v = FMA(m.rtolQ, fabs(qcur[i]), m.atolQ);
        bad |= (v <= 0.0);
        w[i] = 1.0 / v;
    }
    return bad ? -1 : 0;
}

/* ---- stored trajectory (records as in bdf_kernels.hip, read straight from global memory) ---- */
template <bool BWD>
DEV double rec(const Cm<BWD> &m, int s, int f) { return m.traj[((int64_t)s * TREC + f) * m.tS]; }
template <bool BWD>
DEV double point_time(const Cm<BWD> &m, int s) { return rec(m, s, 2); }

template <bool BWD>
DEV int interp_y(Cm<BWD> &m, double t)
{
    if (m.have_last && t == m.last_t) return CV_SUCCESS;
    m.n_interp++;
    int newpoint = 0, indx;
    if (m.newdata) {
        m.ilast = m.np - 1; newpoint = 1; m.newdata = 0;
        m.tlo = point_time(m, m.ilast - 1); m.thi = point_time(m, m.ilast);
        m.tlo2 = (m.ilast >= 2) ? point_time(m, m.ilast - 2) : m.tlo;
    }
    const int ilast = m.ilast;
    const bool to_left = (t - m.tlo) < 0.0;
    const bool to_right = (t - m.thi) > 0.0;
    indx = ilast;
    if (to_left) {
        newpoint = 1;
        double tprev = m.tlo, tcur = m.thi;
        for (;;) {
            if (indx == 0) break;
            if ((t - tprev) <= 0.0) {
                indx--;
                tcur = tprev;
                if (indx > 0) tprev = (indx == ilast - 1) ? m.tlo2 : point_time(m, indx - 1);
            } else break;
        }
        m.ilast = (indx == 0) ? 1 : indx;
        if (indx == 0) {
            m.tlo = tcur; m.thi = point_time(m, 1);
            if (fabs(t - m.tlo) > FUZZ_FACTOR_ADJ * UROUND) return CV_GETY_BADT;
        } else {
            m.tlo = tprev; m.thi = tcur;
        }
    } else if (to_right) {
        newpoint = 1;
        double tcur = m.thi, tprev = m.tlo;
        for (;;) {
            if (indx >= m.np - 1) break;
            if ((t - tcur) > 0.0) {
                indx++;
                tprev = tcur;
                tcur = point_time(m, indx);
            } else break;
        }
        m.ilast = indx;
        m.tlo = tprev; m.thi = tcur;
        if ((t - m.thi) > FUZZ_FACTOR_ADJ * UROUND * (fabs(m.tfinal) + 1.0)) return CV_GETY_BADT;
    }
    m.have_last = 1;
    m.last_t = t;
    if (indx == 0) {
        for (int i = 0; i < NS; i++) m.ytmp[i] = rec(m, 0, 8 + i);
        return CV_SUCCESS;
    }
#ifdef SA_HERMITE
    {   /* CVAhermiteGetY (see the oracle); Y0 / Y1 of the interval live in O_HY[0..n) / O_HY[n..2n) */
        const double t0 = rec(m, indx - 1, 2), t1 = rec(m, indx, 2);
        const double delta = t1 - t0;
        if (newpoint) {
            m.n_rebuild++;
            m.cur_idx = indx;
            for (int i = 0; i < NS; i++) {
                const double y0 = rec(m, indx - 1, 8 + i), yd0 = rec(m, indx - 1, 8 + NS + i);
                const double y1 = rec(m, indx, 8 + i), yd1 = rec(m, indx, 8 + NS + i);
                const double dy = y1 - y0;
                W(m, O_HY, i) = FMA(-delta, yd0, dy);
                W(m, O_HY, NS + i) = FMA(delta, yd1 + yd0, -2.0 * dy);
            }
            if (indx == m.ilast) m.tlo2 = (indx >= 2) ? point_time(m, indx - 2) : m.tlo;
        }
        const double factor1 = t - t0;
        double factor2 = factor1 / delta;
        factor2 = factor2 * factor2;
        const double factor3 = factor2 * (t - t1) / delta;
        for (int i = 0; i < NS; i++) {
            double acc = FMA(factor1, rec(m, indx - 1, 8 + NS + i), rec(m, indx - 1, 8 + i));
            acc = FMA(factor2, W(m, O_HY, i), acc);
            acc = FMA(factor3, W(m, O_HY, NS + i), acc);
            m.ytmp[i] = acc;
        }
        return CV_SUCCESS;
    }
#endif
    if (newpoint) {
        m.n_rebuild++;
        m.cur_idx = indx;
        if (rec(m, indx, 0) > (double)indx) return CV_GETY_BADT;
        if (indx == m.ilast) m.tlo2 = rec(m, indx, 4);
    }
    {
        const int ci = m.cur_idx;
        const int order = (int)rec(m, ci, 0);
        const double inv_dt = 1.0 / rec(m, ci, 1);
        double cvals[QMAX + 1];
        cvals[0] = 1.0;
        SFOR(i, 0, QMAX) cvals[i + 1] = (i < order) ? cvals[i] * (t - rec(m, ci, 2 + i)) * inv_dt : 0.0; SEND
        for (int k = 0; k < NS; k++) {
            double acc = cvals[0] * rec(m, ci, 8 + k);
            SFOR(i, 1, (QMAX) + 1) if (i <= order) acc = FMA(cvals[i], rec(m, ci, 8 + i * NS + k), acc); SEND
            m.ytmp[k] = acc;
        }
    }
    return CV_SUCCESS;
}

/* ---- callbacks through strided views of the workspace (inputs and outputs are workspace vectors: one stride) ---- */
template <bool BWD>
DEV int cv_f(Cm<BWD> &m, double t, const Vec &y, const Vec &out)
{
    m.nfe++;
    StrideSink sink{out.p, out.S};
    if (BWD) return sa_adj_rhs(t, m.ytmp.p, y.p, m.ps, m.pr, sink);
    return sa_rhs(t, y.p, m.ps, m.pr, sink);
}

template <bool BWD>
DEV int cv_fQ(Cm<BWD> &m, double t, const Vec &y, const Vec &out)
{
    m.nfQe++;
    StrideSink sink{out.p, out.S};
    return sa_quad_rhs(t, m.ytmp.p, y.p, m.ps, m.pr, sink);
}

template <bool BWD>
DEV int cv_jac(Cm<BWD> &m, double t, const Vec &y)
{
    StrideSink sink{&W(m, O_A, 0), m.S};
    if (BWD) return sa_adj_jac(t, m.ytmp.p, m.ps, m.pr, sink);
    return sa_jac(t, y.p, m.ps, m.pr, sink);
}

#ifdef SA_SENS
/* sensitivity right-hand side for all parameters: SV(v_out)[is] = J(t,y) SV(v_in)[is] + df/dp_is (oracle cv_fS) */
template <int v_in, int v_out, bool BWD>
DEV int cv_fS(Cm<BWD> &m, double t, const Vec &y)
{
    m.nfSe++;
    StrideSink sj{&W(m, O_JT, 0), m.S}, sp{&W(m, O_DP, 0), m.S};
    int rc = sa_jac(t, y.p, m.ps, m.pr, sj);
    if (rc != 0) return rc;
    rc = sa_dydp(t, y.p, m.ps, m.pr, sp);
    int bad = 0;
    for (int is = 0; is < NQ; is++)
        for (int i = 0; i < NS; i++) {
            double acc = W(m, O_JT, 0 * NS + i) * SV(m, v_in, is, 0);
            for (int j = 1; j < NS; j++) acc = FMA(W(m, O_JT, j * NS + i), SV(m, v_in, is, j), acc);
            acc = acc + W(m, O_DP, is * NS + i);
            SV(m, v_out, is, i) = acc;
            bad |= !(acc * 0.0 == 0.0);
        }
    return (rc != 0 || bad) ? 1 : 0;
}
#endif

/* ---- dense LU (denseGETRF / denseGETRS, column-major) on the workspace matrix ---- */
#define AE(m, i, j) W(m, O_A, (j) * NS + (i))

template <bool BWD>
DEV int dense_getrf(Cm<BWD> &m)
{
    for (int k = 0; k < NS; k++) {
        int l = k;
        double best = fabs(AE(m, k, k));
        for (int i = k + 1; i < NS; i++) {
            double v = fabs(AE(m, i, k));
            if (v > best) { best = v; l = i; }
        }
        W(m, O_PIV, k) = (double)l;
        if (AE(m, l, k) == 0.0) return k + 1;
        if (l != k) {
            for (int c = 0; c < NS; c++) {
                double tmp = AE(m, l, c);
                AE(m, l, c) = AE(m, k, c);
                AE(m, k, c) = tmp;
            }
        }
        double mult = 1.0 / AE(m, k, k);
        W(m, O_INVP, k) = mult;
        for (int i = k + 1; i < NS; i++) AE(m, i, k) *= mult;
        for (int j = k + 1; j < NS; j++) {
            double a_kj = AE(m, k, j);
            if (a_kj != 0.0) {
                for (int i = k + 1; i < NS; i++) AE(m, i, j) = FMA(-a_kj, AE(m, i, k), AE(m, i, j));
            }
        }
    }
    return 0;
}

template <bool BWD, class B>
DEV void dense_getrs(const Cm<BWD> &m, B &b)
{
    for (int k = 0; k < NS; k++) {
        int pk = (int)W(m, O_PIV, k);
        if (pk != k) { double tmp = b[k]; b[k] = b[pk]; b[pk] = tmp; }
    }
    for (int k = 0; k < NS - 1; k++) {
        double bk = b[k];
        for (int i = k + 1; i < NS; i++) b[i] = FMA(-AE(m, i, k), bk, b[i]);
    }
    for (int k = NS - 1; k > 0; k--) {
        double bk = b[k] * W(m, O_INVP, k);
        b[k] = bk;
        for (int i = 0; i < k; i++) b[i] = FMA(-AE(m, i, k), bk, b[i]);
    }
    if (NS > 0) b[0] = b[0] * W(m, O_INVP, 0);
}

/* cvLsSetup on SUNLinSol_Dense: Jacobian (fresh or saved), I - gamma J, LU */
template <bool BWD>
DEV int cv_lsetup(Cm<BWD> &m, int convfail)
{
    double dgamma = fabs((m.gamma / m.gammap) - 1.0);
    int jbad = (m.nst == 0) || (m.nst > m.nstlj + MSBJ) ||
               ((convfail == CV_FAIL_BAD_J) && (dgamma < CVLS_DGMAX)) ||
               (convfail == CV_FAIL_OTHER);
    int jret = 0;
    if (!jbad) {
        m.jcur = 0;
        for (int i = 0; i < NS * NS; i++) W(m, O_A, i) = W(m, O_SJ, i);
    } else {
        m.nje++;
        m.nstlj = m.nst;
        m.jcur = 1;
        jret = cv_jac(m, m.tn, m.y);
        if (jret == 0) { for (int i = 0; i < NS * NS; i++) W(m, O_SJ, i) = W(m, O_A, i); }
    }
    if (jret < 0) return -1;
    if (jret > 0) return 1;
    const double c = -m.gamma;
    for (int j = 0; j < NS; j++)
        for (int i = 0; i < NS; i++) {
            if (i == j) AE(m, i, j) = FMA(c, AE(m, i, j), 1.0);
            else AE(m, i, j) *= c;
        }
    int ier = dense_getrf(m);
    return ier > 0 ? 1 : 0;
}

#include "bdf_core.h"
static_assert(SA_TMPV_SLOTS == 20 && SA_TMPQ_SLOTS == 5, "O_TMP / O_TMPQ reserve 20 + 5 slots");

/* forward: build the divided-difference record of the newest point from the history in O_HY
   (hY[j] = point s-j) directly in the trajectory record (see bdf_kernels.hip::store_table) */
template <bool BWD>
DEV void store_table(Cm<BWD> &m, double *r, int64_t tS, int order, double dt, const double (&hT)[QMAX + 1])
{
#define RF(f) r[(int64_t)(f) * tS]
    RF(0) = (double)order;
    RF(1) = dt;
    SFOR(j, 0, (QMAX) + 1) RF(2 + j) = hT[j]; SEND
    for (int j = 0; j <= QMAX; j++)
        for (int k = 0; k < NS; k++) RF(8 + j * NS + k) = W(m, O_HY, j * NS + k);
    for (int i = 1; i <= order; i++)
        for (int j = order; j >= i; j--) {
            double factor = dt / (pick(hT, j) - pick(hT, j - i));
            for (int k = 0; k < NS; k++) RF(8 + j * NS + k) = factor * (RF(8 + j * NS + k) - RF(8 + (j - 1) * NS + k));
        }
#undef RF
}

#ifdef SA_HERMITE
/* CV_HERMITE data point {t, y, y'}; y' = f(t0, y0) (kept in O_HY[5n..6n) by cv_first_call) for the first
   point, zn[1] / h afterwards */
template <bool BWD>
DEV void store_hermite(Cm<BWD> &m, double *r, int64_t tS, double t, bool first)
{
    r[0] = 0.0;
    r[(int64_t)1 * tS] = 1.0;
    r[(int64_t)2 * tS] = t;
    for (int i = 0; i < NS; i++) {
        r[(int64_t)(8 + i) * tS] = m.zn[0][i];
        r[(int64_t)(8 + NS + i) * tS] = first ? m.f0[i] : (1.0 / m.h) * m.zn[1][i];
    }
}
#endif

/* common part of the kernels' set-up */
template <bool BWD>
DEV void init_state(Cm<BWD> &m, double *ws, int64_t ws_stride, int inst, const double *ps, const double *pr, int rem_stride)
{
    bind_workspace(m, ws, ws_stride, inst);
    SFOR(i, 0, NQ) m.ps[i] = ps[(int64_t)inst * NQ + i]; SEND
    m.pr = pr + (int64_t)inst * rem_stride;
    m.np = 0; m.tfinal = 0.0; m.ilast = 0; m.newdata = 0; m.have_last = 0; m.cur_idx = 0;
    m.last_t = 0.0; m.tlo = m.thi = m.tlo2 = 0.0; m.n_interp = 0; m.n_rebuild = 0;
    m.traj = nullptr; m.tS = 0;
    m.sensi = 0; m.ism = 0; m.cons = nullptr; m.constr = 0;
    m.rtolQ = 0.0; m.atolQ = 0.0; m.tstop = 0.0;
    for (int i = 0; i < NS; i++) m.ytmp[i] = 0.0;
}

/* ------------------------------------------------------------------------------------ */
extern "C" __global__ void __launch_bounds__(64) sa_k_forward(sa_fwd_args a)
{
    const int inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= a.B) return;
    Cm<false> m;
    init_state(m, a.ws, a.ws_stride, inst, a.ps, a.pr, a.rem_stride);
    m.cons = a.constraints; m.constr = (a.constraints != nullptr);
    m.rtol = a.rtol; m.atol = Atol{a.atol, 0.0};

    const Vec y0{const_cast<double *>(a.y0) + (int64_t)inst * NS, 1};
    const Vec q0 = wvec(m, O_QUAD);                       /* (no quadratures in the forward problem: never read) */
    cv_reinit(m, a.t0, y0, q0);

    /* store: CVodeF semantics (every step is a data point, no mxstep budget); wr: the points are written to the
       arena (SA_MODE_ADJ_COUNT runs the identical pass and only counts them, see sunode_amd.cpp) */
    const bool store = (a.mode != SA_MODE_PLAIN), wr = (a.mode == SA_MODE_ADJ_FWD);
    double *yo = a.y_out + (int64_t)inst * a.n_t * NS;
    double *trec = a.traj + inst;                       /* point s: trec + s*TREC*tS */
    const int64_t tS = a.traj_stride;
    double hT[QMAX + 1];
    SFOR(j, 0, (QMAX) + 1) hT[j] = 0.0; SEND
    for (int j = 0; j <= QMAX; j++) for (int i = 0; i < NS; i++) W(m, O_HY, j * NS + i) = 0.0;

    int status = CV_SUCCESS, k = 0, np = 0, nstloc = 0, retries = 0, total_retries = 0, attempts = 0;
    while (k < a.n_t && a.tvals[k] == a.t0) {
        for (int i = 0; i < NS; i++) yo[(int64_t)k * NS + i] = y0[i];
        k++;
    }
    bool done = (k >= a.n_t);
    StepCtl c;
    c.in_step = 0; c.redo = 0; c.nflag = FIRST_CALL; c.ncf = c.nef = c.nefQ = 0; c.ncfS = c.nefS = 0; c.convfail = 0; c.saved_t = a.t0;
    if (!done) {
        int flag = cv_first_call(m, a.tvals[k]);
        if (flag != CV_SUCCESS) { status = flag; done = true; }
        else if (store) {
#ifdef SA_HERMITE
            if (wr) store_hermite(m, trec, tS, m.tn, true);
#else
            hT[0] = m.tn;
            for (int i = 0; i < NS; i++) W(m, O_HY, i) = m.zn[0][i];
            if (wr) store_table(m, trec, tS, 0, 1.0, hT);
#endif
            np = 1;
        }
    }
    while (!done) {
        if (!c.in_step) {
            int ier = cv_pre_step(m);
            if (ier == CV_ILL_INPUT) { status = ier; done = true; }
            else if (!store && a.mxstep > 0 && nstloc >= a.mxstep) {
                retries++; total_retries++;
                if (retries >= a.max_retries) { status = CV_TOO_MUCH_WORK; done = true; }
                else nstloc = 0;
            }
            if (!done && ier != CV_SUCCESS) { status = ier; done = true; }
        }
        if (!done) {
            attempts++;
            int r = cv_attempt(m, c);
            if (r < 0) { status = r; done = true; }
            else if (r == 1) {
                nstloc++;
                if (store) {
                    if (np >= a.traj_max) { status = SA_TRAJ_FULL; done = true; }    /* bounded in every store mode */
                    else {
#ifdef SA_HERMITE
                        if (wr && np < a.traj_cap) store_hermite(m, trec + (int64_t)np * TREC * tS, tS, m.tn, false);
#else
                        SFOR_DOWN(j, QMAX, 1) hT[j] = hT[j - 1]; SEND
                        hT[0] = m.tn;
                        for (int j = QMAX; j >= 1; j--)
                            for (int i = 0; i < NS; i++) W(m, O_HY, j * NS + i) = W(m, O_HY, (j - 1) * NS + i);
                        for (int i = 0; i < NS; i++) W(m, O_HY, i) = m.zn[0][i];
                        if (wr && np < a.traj_cap) store_table(m, trec + (int64_t)np * TREC * tS, tS, m.qu, fabs(hT[0] - hT[1]), hT);
#endif
                        np++;
                    }
                }
                while (!done && k < a.n_t) {
                    double tout = a.tvals[k];
                    if (tout == a.t0) {
                        for (int i = 0; i < NS; i++) yo[(int64_t)k * NS + i] = y0[i];
                        k++;
                    } else if ((m.tn - tout) * m.h >= 0.0) {
                        cv_get_dky0(m, tout, Vec{yo + (int64_t)k * NS, 1}, wvec(m, O_QOUT));
                        k++;
                        nstloc = 0; retries = 0;
                    } else break;
                }
                if (k >= a.n_t) done = true;
            }
        }
    }
    if (status != CV_SUCCESS) {
        for (int j = 0; j < a.n_t * NS; j++) yo[j] = SA_NAN;
    }
    a.status[inst] = status;
    if (store) {
        a.traj_np[inst] = (status == CV_SUCCESS) ? np : 0;
        /* outgrew the rows of this launch (nothing written beyond them): the host re-integrates exactly sized */
        if (wr && status == CV_SUCCESS && np > a.traj_cap)
            (void)__hip_atomic_fetch_max(a.overflow, np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int64_t st[SA_N_STATS];
    SFOR(i, 0, SA_N_STATS) st[i] = 0; SEND
    accumulate_stats(m, st);
    st[ST_NPTS] = np; st[ST_RETRIES] = total_retries; st[ST_ATTEMPTS] = attempts;
    SFOR(i, 0, SA_N_STATS) a.stats[(int64_t)inst * SA_N_STATS + i] = st[i]; SEND
}

extern "C" __global__ void __launch_bounds__(64) sa_k_backward(sa_bwd_args a)
{
    const int inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= a.B) return;
    int64_t st[SA_N_STATS];
    SFOR(i, 0, SA_N_STATS) st[i] = 0; SEND
    int status = CV_SUCCESS;
    const int np = a.traj_np[inst];
    if (a.fwd_status[inst] != CV_SUCCESS || np < 2) status = CV_NO_FWD;

    Cm<true> m;
    init_state(m, a.ws, a.ws_stride, inst, a.ps, a.pr, a.rem_stride);
    m.rtol = a.rtolB; m.atol = Atol{nullptr, a.atolB};
    m.rtolQ = a.rtolQB; m.atolQ = a.atolQB;
    m.tstop = a.tinitial;
    m.traj = a.traj + inst;
    m.tS = a.traj_stride;
    m.np = np;
    m.tfinal = (status == CV_SUCCESS) ? rec(m, np - 1, 2) : a.tinitial;
    m.newdata = 1;

    const Vec lam = wvec(m, O_LAM), quad = wvec(m, O_QUAD), quad_out = wvec(m, O_QOUT);
    for (int i = 0; i < NS; i++) lam[i] = 0.0;
    for (int i = 0; i < NQ; i++) { quad[i] = 0.0; quad_out[i] = 0.0; }
    const double *g = a.grads + (int64_t)inst * a.grads_stride;
    bool first_call = true;
    int total_retries = 0, attempts = 0;
    cv_reinit(m, a.t0, lam, quad);

    for (int iv = 0; iv <= a.n_t; iv++) {
        const double t_upper = (iv == 0) ? a.t0 : a.tvals[a.n_t - iv];
        const double t_lower = (iv == a.n_t) ? a.tend : a.tvals[a.n_t - 1 - iv];
        if (t_lower < t_upper) {
            if (status == CV_SUCCESS) {
                cv_reinit(m, t_upper, lam, quad);          /* CVodeReInitB + CVodeQuadReInitB */
                if (first_call) {
                    if ((t_upper - a.tinitial) < 0.0 || (m.tfinal - t_upper) < 0.0) status = CV_BAD_TB0;
                    first_call = false;
                }
                if (status == CV_SUCCESS && ((t_lower - a.tinitial) < 0.0 || (m.tfinal - t_lower) < 0.0)) {
                    double tfuzz = 100.0 * UROUND * (fabs(a.tinitial) + fabs(m.tfinal));
                    if ((t_lower - a.tinitial) < -tfuzz || (m.tfinal - t_lower) < -tfuzz) status = CV_ILL_INPUT;
                }
                if (status == CV_SUCCESS) {
                    int flag = cv_first_call(m, t_lower);
                    if (flag != CV_SUCCESS) status = flag;
                }
            }
            int nstloc = 0, retries = 0;
            StepCtl c;
            c.in_step = 0; c.redo = 0; c.nflag = FIRST_CALL; c.ncf = c.nef = c.nefQ = 0; c.ncfS = c.nefS = 0; c.convfail = 0;
            c.saved_t = t_upper;
            bool idone = (status != CV_SUCCESS);
            while (!idone) {
                if (!c.in_step) {
                    int ier = cv_pre_step(m);
                    if (ier == CV_ILL_INPUT) { status = ier; idone = true; }
                    else if (a.mxstep > 0 && nstloc >= a.mxstep) {
                        retries++; total_retries++;
                        if (retries >= a.max_retries) { status = CV_TOO_MUCH_WORK; idone = true; }
                        else nstloc = 0;
                    }
                    if (!idone && ier != CV_SUCCESS) { status = ier; idone = true; }
                }
                if (!idone) {
                    attempts++;
                    int r = cv_attempt(m, c);
                    if (r < 0) { status = r; idone = true; }
                    else if (r == 1) {
                        nstloc++;
                        double troundoff = FUZZ_FACTOR * UROUND * (fabs(m.tn) + fabs(m.h));
                        if (fabs(m.tn - m.tstop) <= troundoff) m.tn = m.tstop;
                        if ((m.tn - t_lower) * m.h >= 0.0) {
                            cv_get_dky0(m, t_lower, lam, quad_out);
                            idone = true;
                        } else {
                            troundoff = FUZZ_FACTOR * UROUND * (fabs(m.tn) + fabs(m.h));
                            if (fabs(m.tn - m.tstop) <= troundoff) { status = CV_TSTOP_RETURN; idone = true; }
                            else if ((m.tn + m.hprime - m.tstop) * m.h > 0.0) {
                                m.hprime = (m.tstop - m.tn) * (1.0 - 4.0 * UROUND);
                                m.eta = m.hprime / m.h;
                            }
                        }
                    }
                }
            }
            if (status == CV_SUCCESS || m.nst > 0) accumulate_stats(m, st);
            if (status == CV_SUCCESS) { for (int i = 0; i < NQ; i++) quad[i] = quad_out[i]; }
        }
        if (iv < a.n_t && status == CV_SUCCESS) {
            const double *gi = g + (int64_t)(a.n_t - 1 - iv) * NS;
            for (int i = 0; i < NS; i++) lam[i] -= gi[i];
            const int64_t row = (int64_t)inst * a.n_t + (iv == 0 ? 0 : a.n_t - iv);
            if (a.lamda_all) for (int i = 0; i < NS; i++) a.lamda_all[row * NS + i] = lam[i];
            if (a.quad_all) for (int i = 0; i < NQ; i++) a.quad_all[row * NQ + i] = quad[i];
        }
    }
    if (status != CV_SUCCESS) {
        if (a.lamda_all) for (int j = 0; j < a.n_t * NS; j++) a.lamda_all[(int64_t)inst * a.n_t * NS + j] = SA_NAN;
        if (a.quad_all) for (int j = 0; j < a.n_t * NQ; j++) a.quad_all[(int64_t)inst * a.n_t * NQ + j] = SA_NAN;
    }
    for (int i = 0; i < NQ; i++) a.grad_out[(int64_t)inst * NQ + i] = (status == CV_SUCCESS) ? quad_out[i] : SA_NAN;
    for (int i = 0; i < NS; i++) a.lamda_out[(int64_t)inst * NS + i] = (status == CV_SUCCESS) ? lam[i] : SA_NAN;
    a.status[inst] = status;
    st[ST_NPTS] = np; st[ST_NINTERP] = m.n_interp; st[ST_NREBUILD] = m.n_rebuild;
    st[ST_RETRIES] = total_retries; st[ST_ATTEMPTS] = attempts;
    SFOR(i, 0, SA_N_STATS) a.stats[(int64_t)inst * SA_N_STATS + i] = st[i]; SEND
}

#ifdef SA_SENS
/* Solver(sens_mode).solve (reference solver.py:467-527): CVodeReInit + CVodeSensReInit, then per output
   time CVode(NORMAL) with the mxstep x max_retries budget, CVodeGetSens */
extern "C" __global__ void __launch_bounds__(64) sa_k_sens(sa_sens_args a)
{
    const int inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= a.B) return;
    Cm<false> m;
    init_state(m, a.ws, a.ws_stride, inst, a.ps, a.pr, a.rem_stride);
    SFOR(i, 0, NQ) m.pbar[i] = a.pbar[i]; SEND
    m.rtol = a.rtol; m.atol = Atol{a.atol, 0.0};
    m.sensi = 1; m.ism = a.ism;

    const Vec y0{const_cast<double *>(a.y0) + (int64_t)inst * NS, 1};
    const double *s0 = a.sens0 + (int64_t)inst * NQ * NS;
    cv_reinit(m, a.t0, y0, wvec(m, O_QUAD));
    for (int v = 0; v < SV_COUNT; v++)
        for (int is = 0; is < NQ; is++)
            for (int i = 0; i < NS; i++) SV(m, v, is, i) = (v == SV_ZN0) ? s0[is * NS + i] : 0.0;

    double *yo = a.y_out + (int64_t)inst * a.n_t * NS;
    double *so = a.sens_out + (int64_t)inst * a.n_t * NQ * NS;
    int status = CV_SUCCESS, k = 0, nstloc = 0, retries = 0, total_retries = 0, attempts = 0;
    while (k < a.n_t && a.tvals[k] == a.t0) {
        for (int i = 0; i < NS; i++) yo[(int64_t)k * NS + i] = y0[i];
        for (int j = 0; j < NQ * NS; j++) so[(int64_t)k * NQ * NS + j] = s0[j];
        k++;
    }
    bool done = (k >= a.n_t);
    StepCtl c;
    c.in_step = 0; c.redo = 0; c.nflag = FIRST_CALL; c.ncf = c.nef = c.nefQ = 0; c.ncfS = c.nefS = 0; c.convfail = 0;
    c.saved_t = a.t0;
    if (!done) {
        int flag = cv_first_call(m, a.tvals[k]);
        if (flag != CV_SUCCESS) { status = flag; done = true; }
    }
    while (!done) {
        if (!c.in_step) {
            int ier = cv_pre_step(m);
            if (ier == CV_ILL_INPUT) { status = ier; done = true; }
            else if (a.mxstep > 0 && nstloc >= a.mxstep) {
                retries++; total_retries++;
                if (retries >= a.max_retries) { status = CV_TOO_MUCH_WORK; done = true; }
                else nstloc = 0;
            }
            if (!done && ier != CV_SUCCESS) { status = ier; done = true; }
        }
        if (!done) {
            attempts++;
            int r = cv_attempt(m, c);
            if (r < 0) { status = r; done = true; }
            else if (r == 1) {
                nstloc++;
                while (!done && k < a.n_t) {
                    double tout = a.tvals[k];
                    if (tout == a.t0) {
                        for (int i = 0; i < NS; i++) yo[(int64_t)k * NS + i] = y0[i];
                        for (int j = 0; j < NQ * NS; j++) so[(int64_t)k * NQ * NS + j] = s0[j];
                        k++;
                    } else if ((m.tn - tout) * m.h >= 0.0) {
                        cv_get_dky0(m, tout, Vec{yo + (int64_t)k * NS, 1}, wvec(m, O_QOUT));
                        {   /* CVodeGetSensDky, k = 0, all parameters (t validated by cv_get_dky0; bdf_kernels.hip's form) */
                            const double sx = (tout - m.tn) / m.h;
                            double pw[QMAX + 1];
                            pw[0] = 1.0;
                            SFOR(j, 1, (QMAX) + 1) pw[j] = pw[j - 1] * sx; SEND
                            for (int is = 0; is < NQ; is++)
                                for (int i = 0; i < NS; i++) {
                                    double acc = pw[QMAX] * SV(m, SV_ZN0 + QMAX, is, i);
                                    SFOR_DOWN(j, QMAX - 1, 0) acc = FMA(pw[j], SV(m, SV_ZN0 + j, is, i), acc); SEND
                                    so[((int64_t)k * NQ + is) * NS + i] = acc;
                                }
                        }
                        k++;
                        nstloc = 0; retries = 0;
                    } else break;
                }
                if (k >= a.n_t) done = true;
            }
        }
    }
    if (status != CV_SUCCESS) {
        for (int j = 0; j < a.n_t * NS; j++) yo[j] = SA_NAN;
        for (int j = 0; j < a.n_t * NQ * NS; j++) so[j] = SA_NAN;
    }
    a.status[inst] = status;
    int64_t st[SA_N_STATS];
    SFOR(i, 0, SA_N_STATS) st[i] = 0; SEND
    accumulate_stats(m, st);
    /* sensitivity counters ride in the quadrature / interpolation slots of the adjoint path */
    st[ST_NFQE] = m.nfSe; st[ST_NETFQ] = m.netfS; st[ST_NINTERP] = m.nniS; st[ST_NREBUILD] = m.ncfnS;
    st[ST_RETRIES] = total_retries; st[ST_ATTEMPTS] = attempts;
    SFOR(i, 0, SA_N_STATS) a.stats[(int64_t)inst * SA_N_STATS + i] = st[i]; SEND
}
#endif

/* callback evaluation + arithmetic probe (plain arrays, unit stride) */
extern "C" __global__ void __launch_bounds__(64) sa_k_eval(sa_eval_args a)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.npts) return;
    double ps[NQD];
    SFOR(k, 0, NQ) ps[k] = a.ps[(int64_t)i * NQ + k]; SEND
    const double *prp = a.pr + (int64_t)i * NR;
    const double *y = a.y + (int64_t)i * NS, *lam = a.lam + (int64_t)i * NS;
    const double t = a.t[i];
    StrideSink s_rhs{a.rhs + (int64_t)i * NS, 1}, s_jac{a.jac + (int64_t)i * NS * NS, 1}, s_adj{a.adj + (int64_t)i * NS, 1},
        s_quad{a.quad + (int64_t)i * NQ, 1}, s_ajac{a.adjjac + (int64_t)i * NS * NS, 1};
    a.codes[i * 5 + 0] = sa_rhs(t, y, ps, prp, s_rhs);
    a.codes[i * 5 + 1] = sa_jac(t, y, ps, prp, s_jac);
    a.codes[i * 5 + 2] = sa_adj_rhs(t, y, lam, ps, prp, s_adj);
    a.codes[i * 5 + 3] = sa_quad_rhs(t, y, lam, ps, prp, s_quad);
    a.codes[i * 5 + 4] = sa_adj_jac(t, y, ps, prp, s_ajac);
}

extern "C" __global__ void __launch_bounds__(64) sa_k_math(sa_math_args a)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n) return;
    {   /* both coefficient sources of the deterministic pow (literals / constant memory, sa_common.h) must agree bit for bit */
        const double plit = rpower_r<false>(a.x[i], a.y[i]), pcm = rpower_r<true>(a.x[i], a.y[i]);
        a.pow_out[i] = (__builtin_bit_cast(uint64_t, plit) == __builtin_bit_cast(uint64_t, pcm)) ? plit : SA_NAN;
    }
    a.sqrt_out[i] = sqrt(a.x[i]);
    {   /* odd entries with operands far from the exponent limits go through fdiv (cvSet's division): the host test
           compares every entry with the IEEE quotient */
        const double xa = fabs(a.x[i]), ya = fabs(a.y[i]);
        const bool safe = (i & 1) && xa > 1e-100 && xa < 1e100 && ya > 1e-100 && ya < 1e100;
        a.div_out[i] = safe ? fdiv(a.x[i], a.y[i]) : a.x[i] / a.y[i];
    }
}

/* {n_states, n_sub, n_rem, ABI version, lanes per instance, workspace doubles per instance} */
/* arena records [point][field][instance] (a wavefront's lanes touch consecutive doubles); every other family: [instance][point] */
extern "C" __device__ __attribute__((used)) const int32_t sa_traj_point_major = 1;
extern "C" __device__ __attribute__((used)) const int32_t sa_meta[6] = {NS, NQ, NR, SA_DEVICE_ABI_VERSION, 1, WS_DOUBLES};
