/*
 * bdf_mem.hip -- memory-resident thread-per-instance integrator for large systems (n > 64).
 *
 * One lane = one integrator as in bdf_kernels.hip, but the vectors, the Nordsieck array, the Newton
 * matrix / saved Jacobian and the LU live in an HBM workspace laid out [field element][instance]
 * (instance index fastest), so that the 64 lanes of a wave -- which execute the same loop over
 * the components in lockstep -- always touch 64 consecutive doubles: every access is one fully
 * coalesced 512-byte transaction served by L1/L2.  Control scalars (h, q, tau, l, tq, counters) stay
 * in registers.  Loops over components are real loops (code size independent of n); the generated
 * callbacks read the state through SA_Y / SA_LAM and write through SA_STORE with the workspace
 * stride, and run with full lane utilisation (each lane evaluates its own instance), which is what
 * makes this mapping preferable to a cooperative one when the callbacks dominate (n = 100: the
 * dense Jacobian has 10^4 entries).  The dense LU is the serial denseGETRF per lane; its n^3/3
 * multiply-adds stream through L2.
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
#include "sa_common.h"

#define TREC (8 + 6 * NS)
#define SA_NAN __builtin_bit_cast(double, (uint64_t)0x7ff8000000000000ULL)
constexpr int next_pow2_c(int n) { int p = 1; while (p < n) p <<= 1; return p; }
#define PS_TREE next_pow2_c(NS > NQ ? (NS > 0 ? NS : 1) : (NQ > 0 ? NQ : 1))

/* workspace field offsets (in elements; element e of the lane's instance is w[e * S]) */
#define O_ZN 0
#define O_ZNQ (O_ZN + 6 * NS)
#define O_EWT (O_ZNQ + 6 * NQ)
#define O_ACOR (O_EWT + NS)
#define O_TEMPV (O_ACOR + NS)
#define O_FTEMP (O_TEMPV + NS)
#define O_Y (O_FTEMP + NS)
#define O_YTMP (O_Y + NS)
#define O_DELTA (O_YTMP + NS)
#define O_EWTQ (O_DELTA + NS)
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
#define O_TREE (O_HY + 6 * NS)
#ifdef SA_SENS      /* forward sensitivities: Nordsieck arrays and work vectors of the NQ sensitivity systems */
#define O_ZNS (O_TREE + PS_TREE)
#define O_EWTS (O_ZNS + 6 * NQ * NS)
#define O_ACORS (O_EWTS + NQ * NS)
#define O_TEMPVS (O_ACORS + NQ * NS)
#define O_FTEMPS (O_TEMPVS + NQ * NS)
#define O_YS (O_FTEMPS + NQ * NS)
#define O_DELTAS (O_YS + NQ * NS)
#define O_DP (O_DELTAS + NQ * NS)
#define O_JT (O_DP + NQ * NS)
#define WS_DOUBLES (O_JT + NS * NS)
#define ZNS(m, j, is, i) W(m, O_ZNS, ((j) * NQ + (is)) * NS + (i))
#define VS(m, off, is, i) W(m, off, (is) * NS + (i))
#else
#define WS_DOUBLES (O_TREE + PS_TREE)
#endif

/* strided sink for the generated callbacks: slot k of the output lives at p[k * stride] */
struct StrideSink {
    double *p;
    int64_t stride;
    template <int S> __device__ __forceinline__ void put(double x) const { p[(int64_t)S * stride] = x; }
    __device__ __forceinline__ void put_dyn(int slot, double x) const { p[(int64_t)slot * stride] = x; }
};

template <bool BWD>
struct Cm {
    double *w;                 /* workspace, pre-offset to the lane's instance */
    int64_t S;                 /* instance stride of the workspace */
    double atol_s;             /* backward: scalar atol; forward: per-component vector via atol_p */
    const double *atol_p;
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
    /* forward sensitivities (only the SA_SENS build sets sensi) */
    const double *cons;        /* CVodeSetConstraints vector (global) or nullptr */
    int sensi, ism;
    double pbar[NQD], crateS, delpS, acnrmS;
    int nfSe, nniS, ncfnS, netfS, nsetupsS;
};

#define W(m, off, i) (m).w[(int64_t)((off) + (i)) * (m).S]
#define ZN(m, j, i) W(m, O_ZN, (j) * NS + (i))
#define ZNQ(m, j, i) W(m, O_ZNQ, (j) * NQ + (i))

/* ---- norms: balanced tree over the zero-padded power-of-two array, as in the oracle ---- */
template <bool BWD>
DEV double wrms_off(Cm<BWD> &m, int xoff, int woff, int n)
{
    if (n == 0) return 0.0;
    int P = 1;
    while (P < n) P <<= 1;
    for (int i = 0; i < P; i++) {
        double prod = (i < n) ? W(m, xoff, i) * W(m, woff, i) : 0.0;
        W(m, O_TREE, i) = prod * prod;
    }
    for (int s = 1; s < P; s <<= 1)
        for (int i = 0; i < P; i += 2 * s) W(m, O_TREE, i) = W(m, O_TREE, i) + W(m, O_TREE, i + s);
    return sqrt(W(m, O_TREE, 0) / n);
}

template <bool BWD> DEV double wrms_n(Cm<BWD> &m, int xoff) { return wrms_off(m, xoff, O_EWT, NS); }
template <bool BWD> DEV double wrms_q(Cm<BWD> &m, int xoff) { return wrms_off(m, xoff, O_EWTQ, NQ); }

template <bool BWD>
DEV double quad_update_norm(Cm<BWD> &m, double old_nrm, int xoff)
{
    double qnrm = wrms_q(m, xoff);
    return old_nrm > qnrm ? old_nrm : qnrm;
}

template <bool BWD> DEV double atol_of(const Cm<BWD> &m, int i) { return BWD ? m.atol_s : m.atol_p[i]; }

template <bool BWD>
DEV int ewt_set(Cm<BWD> &m, int yoff, int woff)
{
    int bad = 0;
    for (int i = 0; i < NS; i++) {
        double v = FMA(m.rtol, fabs(W(m, yoff, i)), atol_of(m, i));
        bad |= (v <= 0.0);
        W(m, woff, i) = 1.0 / v;
    }
    return bad ? -1 : 0;
}

template <bool BWD>
DEV int ewtQ_set(Cm<BWD> &m, int qoff, int woff)
{
    int bad = 0;
    for (int i = 0; i < NQ; i++) {
        double v = FMA(m.rtolQ, fabs(W(m, qoff, i)), m.atolQ);
        bad |= (v <= 0.0);
        W(m, woff, i) = 1.0 / v;
    }
    return bad ? -1 : 0;
}

#ifdef SA_SENS
/* cvSensEwtSetEE / cvSensUpdateNorm (see the oracle) */
template <bool BWD>
DEV int sens_ewt_set(Cm<BWD> &m, int ysoff, int woff)
{
    int bad = 0;
    for (int is = 0; is < NQ; is++) {
        const double pb = pick(m.pbar, is);
        for (int i = 0; i < NS; i++) {
            double v = FMA(m.rtol, fabs(pb * VS(m, ysoff, is, i)), atol_of(m, i));
            bad |= (v <= 0.0);
            VS(m, woff, is, i) = pb * (1.0 / v);
        }
    }
    return bad ? -1 : 0;
}

template <bool BWD>
DEV double sens_update_norm(Cm<BWD> &m, double old_nrm, int xoff, int woff)
{
    double nrm = old_nrm;
    for (int is = 0; is < NQ; is++) {
        double snrm = wrms_off(m, xoff + is * NS, woff + is * NS, NS);
        if (snrm > nrm) nrm = snrm;
    }
    return nrm;
}
#endif

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
        for (int i = 0; i < NS; i++) W(m, O_YTMP, i) = rec(m, 0, 8 + i);
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
            W(m, O_YTMP, i) = acc;
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
            W(m, O_YTMP, k) = acc;
        }
    }
    return CV_SUCCESS;
}

/* ---- callbacks through strided views of the workspace ---- */
template <bool BWD>
DEV int cv_f(Cm<BWD> &m, double t, int yoff, int outoff)
{
    m.nfe++;
    StrideSink sink{&W(m, outoff, 0), m.S};
    if (BWD) return sa_adj_rhs(t, &W(m, O_YTMP, 0), &W(m, yoff, 0), m.ps, m.pr, sink);
    return sa_rhs(t, &W(m, yoff, 0), m.ps, m.pr, sink);
}

template <bool BWD>
DEV int cv_fQ(Cm<BWD> &m, double t, int yoff, int outoff)
{
    m.nfQe++;
    StrideSink sink{&W(m, outoff, 0), m.S};
    return sa_quad_rhs(t, &W(m, O_YTMP, 0), &W(m, yoff, 0), m.ps, m.pr, sink);
}

template <bool BWD>
DEV int cv_jac(Cm<BWD> &m, double t, int yoff)
{
    StrideSink sink{&W(m, O_A, 0), m.S};
    if (BWD) return sa_adj_jac(t, &W(m, O_YTMP, 0), m.ps, m.pr, sink);
    return sa_jac(t, &W(m, yoff, 0), m.ps, m.pr, sink);
}

#ifdef SA_SENS
/* sensitivity right-hand side for all parameters: out[is] = J(t,y) yS[is] + df/dp_is (oracle cv_fS) */
template <bool BWD>
DEV int cv_fS(Cm<BWD> &m, double t, int yoff, int ysoff, int outoff)
{
    m.nfSe++;
    StrideSink sj{&W(m, O_JT, 0), m.S}, sp{&W(m, O_DP, 0), m.S};
    int rc = sa_jac(t, &W(m, yoff, 0), m.ps, m.pr, sj);
    if (rc != 0) return rc;
    rc = sa_dydp(t, &W(m, yoff, 0), m.ps, m.pr, sp);
    int bad = 0;
    for (int is = 0; is < NQ; is++)
        for (int i = 0; i < NS; i++) {
            double acc = W(m, O_JT, 0 * NS + i) * VS(m, ysoff, is, 0);
            for (int j = 1; j < NS; j++) acc = FMA(W(m, O_JT, j * NS + i), VS(m, ysoff, is, j), acc);
            acc = acc + VS(m, O_DP, is, i);
            VS(m, outoff, is, i) = acc;
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

template <bool BWD>
DEV void dense_getrs(Cm<BWD> &m, int boff)
{
    for (int k = 0; k < NS; k++) {
        int pk = (int)W(m, O_PIV, k);
        if (pk != k) { double tmp = W(m, boff, k); W(m, boff, k) = W(m, boff, pk); W(m, boff, pk) = tmp; }
    }
    for (int k = 0; k < NS - 1; k++) {
        double bk = W(m, boff, k);
        for (int i = k + 1; i < NS; i++) W(m, boff, i) = FMA(-AE(m, i, k), bk, W(m, boff, i));
    }
    for (int k = NS - 1; k > 0; k--) {
        double bk = W(m, boff, k) * W(m, O_INVP, k);
        W(m, boff, k) = bk;
        for (int i = 0; i < k; i++) W(m, boff, i) = FMA(-AE(m, i, k), bk, W(m, boff, i));
    }
    if (NS > 0) W(m, boff, 0) = W(m, boff, 0) * W(m, O_INVP, 0);
}

/* ---- CVodeInit / CVodeReInit (y0 / q0 already stored in zn[0] / znQ[0] by the caller) ---- */
template <bool BWD>
DEV void cv_reinit(Cm<BWD> &m, double t0)
{
    m.tn = t0;
    m.q = 1; m.L = 2; m.qwait = 2; m.etamax = ETAMX1;
    m.qu = 0; m.hu = 0.0;
    m.nst = m.nfe = m.ncfn = m.netf = m.nni = m.nsetups = 0;
    m.nje = 0; m.nstlp = 0; m.nstlj = 0; m.nfQe = m.netfQ = 0;
    m.h = 0.0; m.hprime = 0.0; m.hscale = 0.0; m.eta = 1.0;
    m.qprime = 1;
    m.gamma = m.gammap = 0.0; m.gamrat = 1.0; m.crate = 1.0; m.delp = 0.0;
    m.acnrm = 0.0; m.saved_tq5 = 0.0;
    m.jcur = 0; m.nls_jcur = 0;
    m.crateS = 1.0; m.delpS = 0.0; m.acnrmS = 0.0;
    m.nfSe = m.nniS = m.ncfnS = m.netfS = m.nsetupsS = 0;
    SFOR(i, 0, 7) { m.tau[i] = 0.0; m.l[i] = 0.0; } SEND
    SFOR(i, 0, 6) m.tq[i] = 0.0; SEND
}

/* ---- cvHin ---- */
template <bool BWD>
DEV double cv_upper_bound_h0(Cm<BWD> &m, double tdist)
{
    double hub_inv = 0.0;
    ewt_set(m, O_ZN, O_TEMPV);                         /* temp1 = ewt(zn[0]) */
    for (int i = 0; i < NS; i++) {
        double t2 = fabs(ZN(m, 0, i));
        double t1 = 1.0 / W(m, O_TEMPV, i);
        t1 = FMA(HUB_FACTOR, t2, t1);
        double v = fabs(ZN(m, 1, i)) / t1;
        if (v > hub_inv) hub_inv = v;
    }
    if (BWD) {
        ewtQ_set(m, O_ZNQ, O_TEMPVQ);
        double hubQ_inv = 0.0;
        for (int i = 0; i < NQ; i++) {
            double t2 = fabs(ZNQ(m, 0, i));
            double t1 = 1.0 / W(m, O_TEMPVQ, i);
            t1 = FMA(HUB_FACTOR, t2, t1);
            double v = fabs(ZNQ(m, 1, i)) / t1;
            if (v > hubQ_inv) hubQ_inv = v;
        }
        if (hubQ_inv > hub_inv) hub_inv = hubQ_inv;
    }
#ifdef SA_SENS
    if (m.sensi) {
        sens_ewt_set(m, O_ZNS, O_TEMPVS);
        for (int is = 0; is < NQ; is++)
            for (int i = 0; i < NS; i++) {
                double t2 = fabs(ZNS(m, 0, is, i));
                double t1 = 1.0 / VS(m, O_TEMPVS, is, i);
                t1 = FMA(HUB_FACTOR, t2, t1);
                double v = fabs(ZNS(m, 1, is, i)) / t1;
                if (v > hub_inv) hub_inv = v;
            }
    }
#endif
    double hub = HUB_FACTOR * tdist;
    if (hub * hub_inv > 1.0) hub = 1.0 / hub_inv;
    return hub;
}

template <bool BWD>
DEV int cv_ydd_norm(Cm<BWD> &m, double hg, double *yddnrm)
{
    for (int i = 0; i < NS; i++) W(m, O_Y, i) = FMA(hg, ZN(m, 1, i), ZN(m, 0, i));
    if (BWD) { if (interp_y(m, m.tn + hg) != CV_SUCCESS) { m.nfe++; return CV_RHSFUNC_FAIL; } }
#ifdef SA_SENS
    if (m.sensi)
        for (int is = 0; is < NQ; is++)
            for (int i = 0; i < NS; i++) VS(m, O_YS, is, i) = FMA(hg, ZNS(m, 1, is, i), ZNS(m, 0, is, i));
#endif
    int retval = cv_f(m, m.tn + hg, O_Y, O_TEMPV);
    if (retval < 0) return CV_RHSFUNC_FAIL;
    if (retval > 0) return RHSFUNC_RECVR;
#ifdef SA_SENS
    if (m.sensi) {
        retval = cv_fS(m, m.tn + hg, O_Y, O_YS, O_TEMPVS);
        if (retval < 0) return CV_SRHSFUNC_FAIL;
        if (retval > 0) return SRHSFUNC_RECVR;
    }
#endif
    if (BWD) {
        retval = cv_fQ(m, m.tn + hg, O_Y, O_TEMPVQ);
        if (retval < 0) return CV_QRHSFUNC_FAIL;
        if (retval > 0) return QRHSFUNC_RECVR;
    }
    for (int i = 0; i < NS; i++) {
        double v = W(m, O_TEMPV, i) - ZN(m, 1, i);
        W(m, O_TEMPV, i) = (1.0 / hg) * v;
    }
    *yddnrm = wrms_n(m, O_TEMPV);
    if (BWD) {
        for (int i = 0; i < NQ; i++) {
            double v = W(m, O_TEMPVQ, i) - ZNQ(m, 1, i);
            W(m, O_TEMPVQ, i) = (1.0 / hg) * v;
        }
        *yddnrm = quad_update_norm(m, *yddnrm, O_TEMPVQ);
    }
#ifdef SA_SENS
    if (m.sensi) {
        for (int is = 0; is < NQ; is++)
            for (int i = 0; i < NS; i++) {
                double v = VS(m, O_TEMPVS, is, i) - ZNS(m, 1, is, i);
                VS(m, O_TEMPVS, is, i) = (1.0 / hg) * v;
            }
        *yddnrm = sens_update_norm(m, *yddnrm, O_TEMPVS, O_EWTS);
    }
#endif
    return CV_SUCCESS;
}

template <bool BWD>
DEV int cv_hin(Cm<BWD> &m, double tout)
{
    double tdiff = tout - m.tn;
    if (tdiff == 0.0) return CV_TOO_CLOSE;
    double sign = (tdiff > 0.0) ? 1.0 : -1.0;
    double tdist = fabs(tdiff);
    double tround = UROUND * fmax(fabs(m.tn), fabs(tout));
    if (tdist < 2.0 * tround) return CV_TOO_CLOSE;
    double hlb = HLB_FACTOR * tround;
    double hub = cv_upper_bound_h0(m, tdist);
    double hg = sqrt(hlb * hub);
    if (hub < hlb) {
        m.h = (sign < 0.0) ? -hg : hg;
        return CV_SUCCESS;
    }
    double hs = hg, hnew = hg, yddnrm = 0.0;
    int result = 1;
    for (int count1 = 1; count1 <= HIN_MAX_ITERS && result == 1; count1++) {
        int hgOK = 0;
        for (int count2 = 1; count2 <= HIN_MAX_ITERS; count2++) {
            double hgs = hg * sign;
            int retval = cv_ydd_norm(m, hgs, &yddnrm);
            if (retval < 0) { result = CV_RHSFUNC_FAIL; break; }
            if (retval == CV_SUCCESS) { hgOK = 1; break; }
            hg *= 0.2;
        }
        if (result != 1) break;
        if (!hgOK) {
            if (count1 <= 2) { result = CV_REPTD_RHSFUNC_ERR; break; }
            hnew = hs;
            result = 0;
            break;
        }
        hs = hg;
        hnew = (yddnrm * hub * hub > 2.0) ? sqrt(2.0 / yddnrm) : sqrt(hg * hub);
        if (count1 == HIN_MAX_ITERS) { result = 0; break; }
        double hrat = hnew / hg;
        if ((hrat > 0.5) && (hrat < 2.0)) { result = 0; break; }
        if ((count1 > 1) && (hrat > 2.0)) { hnew = hg; result = 0; break; }
        hg = hnew;
    }
    if (result < 0) return result;
    double h0 = H_BIAS * hnew;
    if (h0 < hlb) h0 = hlb;
    if (h0 > hub) h0 = hub;
    if (sign < 0.0) h0 = -h0;
    m.h = h0;
    return CV_SUCCESS;
}

/* ---- Nordsieck array manipulation (oracle form: columns 1..q, saved correction in zn[qmax]) ---- */
template <bool BWD>
DEV void cv_rescale(Cm<BWD> &m)
{
    double factor = m.eta;
    for (int j = 1; j <= m.q; j++) {
        for (int i = 0; i < NS; i++) ZN(m, j, i) *= factor;
        if (BWD) for (int i = 0; i < NQ; i++) ZNQ(m, j, i) *= factor;
#ifdef SA_SENS
        if (m.sensi) for (int is = 0; is < NQ; is++) for (int i = 0; i < NS; i++) ZNS(m, j, is, i) *= factor;
#endif
        factor *= m.eta;
    }
    m.h = m.hscale * m.eta;
    m.hscale = m.h;
}

template <bool BWD>
DEV void cv_increase_bdf(Cm<BWD> &m)
{
    SFOR(i, 0, (QMAX) + 1) m.l[i] = 0.0; SEND
    double alpha1 = 1.0, prod = 1.0, xiold = 1.0, alpha0 = -1.0, hsum = m.hscale;
    m.l[2] = 1.0;
    SFOR(j, 1, QMAX - 1) {
        if (j < m.q) {
            hsum += m.tau[j + 1];
            double xi = hsum / m.hscale;
            prod *= xi;
            alpha0 -= 1.0 / (j + 1);
            alpha1 += 1.0 / xi;
            SFOR_DOWN(i, j + 2, 2) m.l[i] = FMA(m.l[i], xiold, m.l[i - 1]); SEND
            xiold = xi;
        }
    } SEND
    const double A1 = (-alpha0 - alpha1) / prod;
    const int L = m.L;
    for (int i = 0; i < NS; i++) ZN(m, L, i) = A1 * ZN(m, QMAX, i);
    for (int j = 2; j <= m.q; j++) {
        const double lj = pick(m.l, j);
        for (int i = 0; i < NS; i++) ZN(m, j, i) = FMA(lj, ZN(m, L, i), ZN(m, j, i));
    }
    if (BWD) {
        for (int i = 0; i < NQ; i++) ZNQ(m, L, i) = A1 * ZNQ(m, QMAX, i);
        for (int j = 2; j <= m.q; j++) {
            const double lj = pick(m.l, j);
            for (int i = 0; i < NQ; i++) ZNQ(m, j, i) = FMA(lj, ZNQ(m, L, i), ZNQ(m, j, i));
        }
    }
#ifdef SA_SENS
    if (m.sensi)
        for (int is = 0; is < NQ; is++) {
            for (int i = 0; i < NS; i++) ZNS(m, L, is, i) = A1 * ZNS(m, QMAX, is, i);
            for (int j = 2; j <= m.q; j++) {
                const double lj = pick(m.l, j);
                for (int i = 0; i < NS; i++) ZNS(m, j, is, i) = FMA(lj, ZNS(m, L, is, i), ZNS(m, j, is, i));
            }
        }
#endif
}

template <bool BWD>
DEV void cv_decrease_bdf(Cm<BWD> &m)
{
    SFOR(i, 0, (QMAX) + 1) m.l[i] = 0.0; SEND
    m.l[2] = 1.0;
    double hsum = 0.0;
    SFOR(j, 1, (QMAX - 2) + 1) {
        if (j <= m.q - 2) {
            hsum += m.tau[j];
            double xi = hsum / m.hscale;
            SFOR_DOWN(i, j + 2, 2) m.l[i] = FMA(m.l[i], xi, m.l[i - 1]); SEND
        }
    } SEND
    for (int j = 2; j < m.q; j++) {
        const double lj = pick(m.l, j);
        for (int i = 0; i < NS; i++) ZN(m, j, i) = FMA(-lj, ZN(m, m.q, i), ZN(m, j, i));
        if (BWD) for (int i = 0; i < NQ; i++) ZNQ(m, j, i) = FMA(-lj, ZNQ(m, m.q, i), ZNQ(m, j, i));
#ifdef SA_SENS
        if (m.sensi)
            for (int is = 0; is < NQ; is++)
                for (int i = 0; i < NS; i++) ZNS(m, j, is, i) = FMA(-lj, ZNS(m, m.q, is, i), ZNS(m, j, is, i));
#endif
    }
}

template <bool BWD>
DEV void cv_adjust_order(Cm<BWD> &m, int deltaq)
{
    if ((m.q == 2) && (deltaq != 1)) return;
    if (deltaq == 1) cv_increase_bdf(m);
    else if (deltaq == -1) cv_decrease_bdf(m);
}

template <bool BWD>
DEV void cv_predict(Cm<BWD> &m)
{
    m.tn += m.h;
    if (BWD) {
        if ((m.tn - m.tstop) * m.h > 0.0) m.tn = m.tstop;
    }
    for (int k = 1; k <= m.q; k++)
        for (int j = m.q; j >= k; j--) {
            for (int i = 0; i < NS; i++) ZN(m, j - 1, i) = ZN(m, j - 1, i) + ZN(m, j, i);
            if (BWD) for (int i = 0; i < NQ; i++) ZNQ(m, j - 1, i) = ZNQ(m, j - 1, i) + ZNQ(m, j, i);
#ifdef SA_SENS
            if (m.sensi)
                for (int is = 0; is < NQ; is++)
                    for (int i = 0; i < NS; i++) ZNS(m, j - 1, is, i) = ZNS(m, j - 1, is, i) + ZNS(m, j, is, i);
#endif
        }
}

template <bool BWD>
DEV void cv_restore(Cm<BWD> &m, double saved_t)
{
    m.tn = saved_t;
    for (int k = 1; k <= m.q; k++)
        for (int j = m.q; j >= k; j--) {
            for (int i = 0; i < NS; i++) ZN(m, j - 1, i) = ZN(m, j - 1, i) - ZN(m, j, i);
            if (BWD) for (int i = 0; i < NQ; i++) ZNQ(m, j - 1, i) = ZNQ(m, j - 1, i) - ZNQ(m, j, i);
#ifdef SA_SENS
            if (m.sensi)
                for (int is = 0; is < NQ; is++)
                    for (int i = 0; i < NS; i++) ZNS(m, j - 1, is, i) = ZNS(m, j - 1, is, i) - ZNS(m, j, is, i);
#endif
        }
}

/* ---- linear solver interface ---- */
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
        jret = cv_jac(m, m.tn, O_Y);
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

template <bool BWD>
DEV int cv_nls_lsetup(Cm<BWD> &m, int jbad, int &convfail)
{
    if (jbad) convfail = CV_FAIL_BAD_J;
    int retval = cv_lsetup(m, convfail);
    m.nsetups++;
    m.nls_jcur = m.jcur;
    m.gamrat = 1.0;
    m.gammap = m.gamma;
    m.crate = 1.0;
    m.crateS = 1.0;
    m.nstlp = m.nst;
    if (retval < 0) return CV_LSETUP_FAIL;
    if (retval > 0) return NLS_CONV_RECVR;
    return CV_SUCCESS;
}

template <bool BWD>
DEV int cv_nls_residual(Cm<BWD> &m)          /* res -> O_DELTA */
{
    for (int i = 0; i < NS; i++) W(m, O_Y, i) = ZN(m, 0, i) + W(m, O_ACOR, i);
    int retval = cv_f(m, m.tn, O_Y, O_FTEMP);
    if (retval < 0) return CV_RHSFUNC_FAIL;
    if (retval > 0) return RHSFUNC_RECVR;
    for (int i = 0; i < NS; i++) {
        double r = FMA(m.rl1, ZN(m, 1, i), W(m, O_ACOR, i));
        W(m, O_DELTA, i) = FMA(-m.gamma, W(m, O_FTEMP, i), r);
    }
    return CV_SUCCESS;
}

#ifdef SA_SENS
/* cvNlsResidualSensSim: residuals of the sensitivity systems -> O_DELTAS */
template <bool BWD>
DEV int cv_nls_residual_sens(Cm<BWD> &m)
{
    for (int is = 0; is < NQ; is++)
        for (int i = 0; i < NS; i++) VS(m, O_YS, is, i) = ZNS(m, 0, is, i) + VS(m, O_ACORS, is, i);
    int retval = cv_fS(m, m.tn, O_Y, O_YS, O_FTEMPS);
    if (retval < 0) return CV_SRHSFUNC_FAIL;
    if (retval > 0) return SRHSFUNC_RECVR;
    for (int is = 0; is < NQ; is++)
        for (int i = 0; i < NS; i++) {
            double r = FMA(m.rl1, ZNS(m, 1, is, i), VS(m, O_ACORS, is, i));
            VS(m, O_DELTAS, is, i) = FMA(-m.gamma, VS(m, O_FTEMPS, is, i), r);
        }
    return CV_SUCCESS;
}

/* one Newton update of every sensitivity system with the current factorisation */
template <bool BWD>
DEV void cv_sens_newton_update(Cm<BWD> &m)
{
    for (int is = 0; is < NQ; is++) {
        for (int i = 0; i < NS; i++) VS(m, O_DELTAS, is, i) = -1.0 * VS(m, O_DELTAS, is, i);
        dense_getrs(m, O_DELTAS + is * NS);
        if (m.gamrat != 1.0) {
            double s = 2.0 / (1.0 + m.gamrat);
            for (int i = 0; i < NS; i++) VS(m, O_DELTAS, is, i) *= s;
        }
        for (int i = 0; i < NS; i++) VS(m, O_ACORS, is, i) = VS(m, O_ACORS, is, i) + VS(m, O_DELTAS, is, i);
    }
}
#endif

template <bool BWD>
DEV int cv_newton_pass(Cm<BWD> &m, int callSetup, int jbad, int &convfail, int &in_loop)
{
#ifdef SA_SENS
    const bool sim = m.sensi && m.ism == 0;
#else
    const bool sim = false;
#endif
    in_loop = 0;
    for (int i = 0; i < NS; i++) W(m, O_ACOR, i) = 0.0;
#ifdef SA_SENS
    if (sim) for (int j = 0; j < NQ * NS; j++) W(m, O_ACORS, j) = 0.0;
#endif
    int retval = cv_nls_residual(m);
    if (retval != CV_SUCCESS) return retval;
#ifdef SA_SENS
    if (sim) {
        retval = cv_nls_residual_sens(m);
        if (retval != CV_SUCCESS) return retval;
    }
#endif
    if (callSetup) {
        retval = cv_nls_lsetup(m, jbad, convfail);
        if (retval != CV_SUCCESS) return retval;
    }
    int curiter = 0;
    in_loop = 1;
    for (;;) {
        m.nni++;
        for (int i = 0; i < NS; i++) W(m, O_DELTA, i) = -1.0 * W(m, O_DELTA, i);
        dense_getrs(m, O_DELTA);
        if (m.gamrat != 1.0) {
            double s = 2.0 / (1.0 + m.gamrat);
            for (int i = 0; i < NS; i++) W(m, O_DELTA, i) *= s;
        }
        for (int i = 0; i < NS; i++) W(m, O_ACOR, i) = W(m, O_ACOR, i) + W(m, O_DELTA, i);
        double del = wrms_n(m, O_DELTA);
#ifdef SA_SENS
        if (sim) {
            cv_sens_newton_update(m);
            del = sens_update_norm(m, del, O_DELTAS, O_EWTS);
        }
#endif
        if (curiter > 0) m.crate = fmax(CRDOWN * m.crate, del / m.delp);
        double dcon = del * fmin(1.0, m.crate) * m.tq[4];
        if (dcon <= 1.0) {
            if (curiter == 0) m.acnrm = del;
            else {
                m.acnrm = wrms_n(m, O_ACOR);
#ifdef SA_SENS
                if (sim) m.acnrm = sens_update_norm(m, m.acnrm, O_ACORS, O_EWTS);
#endif
            }
            m.nls_jcur = 0;
            return CV_SUCCESS;
        }
        if ((curiter >= 1) && (del > RDIV * m.delp)) return NLS_CONV_RECVR;
        m.delp = del;
        curiter++;
        if (curiter >= NLS_MAXCOR) return NLS_CONV_RECVR;
        retval = cv_nls_residual(m);
        if (retval != CV_SUCCESS) return retval;
#ifdef SA_SENS
        if (sim) {
            retval = cv_nls_residual_sens(m);
            if (retval != CV_SUCCESS) return retval;
        }
#endif
    }
}

#ifdef SA_SENS
/* cvStgrNls (ism = CV_STAGGERED): Newton on the sensitivity systems with the state fixed */
template <bool BWD>
DEV int cv_stgr_nls(Cm<BWD> &m)
{
    int callSetup = 0, jbad = 0, convfail = CV_FAIL_OTHER, retval;
    for (int j = 0; j < NQ * NS; j++) W(m, O_ACORS, j) = 0.0;
    for (;;) {
        retval = cv_nls_residual_sens(m);
        if (retval != CV_SUCCESS) break;
        if (callSetup) {
            retval = cv_nls_lsetup(m, jbad, convfail);
            m.nsetupsS++;
            if (retval != CV_SUCCESS) break;
        }
        int curiter = 0;
        for (;;) {
            m.nniS++;
            cv_sens_newton_update(m);
            double del = sens_update_norm(m, 0.0, O_DELTAS, O_EWTS);
            if (curiter > 0) m.crateS = fmax(CRDOWN * m.crateS, del / m.delpS);
            double dcon = del * fmin(1.0, m.crateS) * m.tq[4];
            if (dcon <= 1.0) {
                m.acnrmS = (curiter == 0) ? del : sens_update_norm(m, 0.0, O_ACORS, O_EWTS);
                retval = CV_SUCCESS;
                m.nls_jcur = 0;
                break;
            }
            if ((curiter >= 1) && (del > RDIV * m.delpS)) { retval = NLS_CONV_RECVR; break; }
            m.delpS = del;
            curiter++;
            if (curiter >= NLS_MAXCOR) { retval = NLS_CONV_RECVR; break; }
            retval = cv_nls_residual_sens(m);
            if (retval != CV_SUCCESS) break;
        }
        if (retval == CV_SUCCESS) break;
        if ((retval > 0) && !m.nls_jcur) {
            callSetup = 1;
            jbad = 1;
            for (int j = 0; j < NQ * NS; j++) W(m, O_ACORS, j) = 0.0;
            continue;
        }
        break;
    }
    if (retval != CV_SUCCESS) return retval;
    for (int is = 0; is < NQ; is++)
        for (int i = 0; i < NS; i++) VS(m, O_YS, is, i) = ZNS(m, 0, is, i) + VS(m, O_ACORS, is, i);
    return CV_SUCCESS;
}
#endif

template <bool BWD>
DEV int cv_error_test_failed(Cm<BWD> &m, double saved_t, double dsm, int &nef, int &netf_counter)
{
    nef++;
    netf_counter++;
    cv_restore(m, saved_t);
    if (nef == MXNEF) return CV_ERR_FAILURE;
    m.etamax = 1.0;
    if (nef <= MXNEF1) {
        m.eta = 1.0 / (rpower_r(BIAS2 * dsm, inv_int(m.L)) + ADDON);
        m.eta = fmax(ETAMIN, m.eta);
        if (nef >= SMALL_NEF) m.eta = fmin(m.eta, ETAMXF);
        cv_rescale(m);
        return 0;
    }
    if (m.q > 1) {
        m.eta = ETAMIN;
        cv_adjust_order(m, -1);
        m.L = m.q;
        m.q--;
        m.qwait = m.L;
        cv_rescale(m);
        return 0;
    }
    m.eta = ETAMIN;
    m.h *= m.eta;
    m.hscale = m.h;
    m.qwait = LONG_WAIT;
    if (BWD) { if (interp_y(m, m.tn) != CV_SUCCESS) { m.nfe++; return CV_RHSFUNC_FAIL; } }
    int retval = cv_f(m, m.tn, O_ZN, O_TEMPV);
    if (retval < 0) return CV_RHSFUNC_FAIL;
    if (retval > 0) return CV_UNREC_RHSFUNC_ERR;
    for (int i = 0; i < NS; i++) ZN(m, 1, i) = m.h * W(m, O_TEMPV, i);
#ifdef SA_SENS
    if (m.sensi) {
        retval = cv_fS(m, m.tn, O_ZN, O_ZNS, O_TEMPVS);
        if (retval < 0) return CV_SRHSFUNC_FAIL;
        if (retval > 0) return CV_UNREC_SRHSFUNC_ERR;
        for (int is = 0; is < NQ; is++)
            for (int i = 0; i < NS; i++) ZNS(m, 1, is, i) = m.h * VS(m, O_TEMPVS, is, i);
    }
#endif
    if (BWD) {
        retval = cv_fQ(m, m.tn, O_ZN, O_TEMPVQ);
        if (retval < 0) return CV_QRHSFUNC_FAIL;
        if (retval > 0) return CV_UNREC_QRHSFUNC_ERR;
        for (int i = 0; i < NQ; i++) ZNQ(m, 1, i) = m.h * W(m, O_TEMPVQ, i);
    }
    return 0;
}

template <bool BWD>
DEV void cv_complete_step(Cm<BWD> &m)
{
    m.nst++;
    m.hu = m.h;
    m.qu = m.q;
    SFOR_DOWN(i, QMAX, 2) m.tau[i] = (i <= m.q) ? m.tau[i - 1] : m.tau[i]; SEND
    m.tau[2] = ((m.q == 1) && (m.nst > 1)) ? m.tau[1] : m.tau[2];
    m.tau[1] = m.h;
    for (int j = 0; j <= m.q; j++) {
        const double lj = pick(m.l, j);
        for (int i = 0; i < NS; i++) ZN(m, j, i) = FMA(lj, W(m, O_ACOR, i), ZN(m, j, i));
        if (BWD) for (int i = 0; i < NQ; i++) ZNQ(m, j, i) = FMA(lj, W(m, O_ACORQ, i), ZNQ(m, j, i));
    }
#ifdef SA_SENS
    if (m.sensi)
        for (int is = 0; is < NQ; is++)
            for (int j = 0; j <= m.q; j++) {
                const double lj = pick(m.l, j);
                for (int i = 0; i < NS; i++) ZNS(m, j, is, i) = FMA(lj, VS(m, O_ACORS, is, i), ZNS(m, j, is, i));
            }
#endif
    m.qwait--;
    if ((m.qwait == 1) && (m.q != QMAX)) {
#ifdef SA_SENS
        if (m.sensi) for (int is = 0; is < NQ; is++) for (int i = 0; i < NS; i++) ZNS(m, QMAX, is, i) = VS(m, O_ACORS, is, i);
#endif
        for (int i = 0; i < NS; i++) ZN(m, QMAX, i) = W(m, O_ACOR, i);
        if (BWD) for (int i = 0; i < NQ; i++) ZNQ(m, QMAX, i) = W(m, O_ACORQ, i);
        m.saved_tq5 = m.tq[5];
    }
}

template <bool BWD>
DEV void cv_set_eta(Cm<BWD> &m)
{
    if (m.eta < THRESH) {
        m.eta = 1.0;
        m.hprime = m.h;
    } else {
        m.eta = fmin(m.eta, m.etamax);
        m.hprime = m.h * m.eta;
    }
}

template <bool BWD>
DEV void cv_prepare_next_step(Cm<BWD> &m, double dsm)
{
    if (m.etamax == 1.0) {
        m.qwait = m.qwait > 2 ? m.qwait : 2;
        m.qprime = m.q;
        m.hprime = m.h;
        m.eta = 1.0;
        return;
    }
    m.etaq = 1.0 / (rpower_r(BIAS2 * dsm, inv_int(m.L)) + ADDON);
    if (m.qwait != 0) {
        m.eta = m.etaq;
        m.qprime = m.q;
        cv_set_eta(m);
        return;
    }
    m.qwait = 2;
    m.etaqm1 = 0.0;
    if (m.q > 1) {
        double ddn = wrms_off(m, O_ZN + m.q * NS, O_EWT, NS);
        if (BWD) { double dq = wrms_off(m, O_ZNQ + m.q * NQ, O_EWTQ, NQ); ddn = ddn > dq ? ddn : dq; }
#ifdef SA_SENS
        if (m.sensi) ddn = sens_update_norm(m, ddn, O_ZNS + m.q * NQ * NS, O_EWTS);
#endif
        ddn = ddn * m.tq[1];
        m.etaqm1 = 1.0 / (rpower_r(BIAS1 * ddn, inv_int(m.q)) + ADDON);
    }
    m.etaqp1 = 0.0;
    if (m.q != QMAX) {
        if (m.saved_tq5 != 0.0) {
            double base = m.h / m.tau[2];
            double pw = 1.0;
            SFOR(i, 1, (QMAX + 1) + 1) { if (i <= m.L) pw *= base; } SEND
            double cquot = (m.tq[5] / m.saved_tq5) * pw;
            for (int i = 0; i < NS; i++) W(m, O_TEMPV, i) = FMA(-cquot, ZN(m, QMAX, i), W(m, O_ACOR, i));
            double dup = wrms_n(m, O_TEMPV);
            if (BWD) {
                for (int i = 0; i < NQ; i++) W(m, O_TEMPVQ, i) = FMA(-cquot, ZNQ(m, QMAX, i), W(m, O_ACORQ, i));
                dup = quad_update_norm(m, dup, O_TEMPVQ);
            }
#ifdef SA_SENS
            if (m.sensi) {
                for (int is = 0; is < NQ; is++)
                    for (int i = 0; i < NS; i++)
                        VS(m, O_TEMPVS, is, i) = FMA(-cquot, ZNS(m, QMAX, is, i), VS(m, O_ACORS, is, i));
                dup = sens_update_norm(m, dup, O_TEMPVS, O_EWTS);
            }
#endif
            dup = dup * m.tq[3];
            m.etaqp1 = 1.0 / (rpower_r(BIAS3 * dup, inv_int(m.L + 1)) + ADDON);
        }
    }
    double etam = fmax(m.etaqm1, fmax(m.etaq, m.etaqp1));
    if (etam < THRESH) {
        m.eta = 1.0;
        m.qprime = m.q;
    } else if (etam == m.etaq) {
        m.eta = m.etaq;
        m.qprime = m.q;
    } else if (etam == m.etaqm1) {
        m.eta = m.etaqm1;
        m.qprime = m.q - 1;
    } else {
        m.eta = m.etaqp1;
        m.qprime = m.q + 1;
        for (int i = 0; i < NS; i++) ZN(m, QMAX, i) = W(m, O_ACOR, i);
        if (BWD) for (int i = 0; i < NQ; i++) ZNQ(m, QMAX, i) = W(m, O_ACORQ, i);
#ifdef SA_SENS
        if (m.sensi) for (int is = 0; is < NQ; is++) for (int i = 0; i < NS; i++) ZNS(m, QMAX, is, i) = VS(m, O_ACORS, is, i);
#endif
    }
    cv_set_eta(m);
}

/* CVodeGetDky (k = 0): oracle form, sum from column q down to 0 */
template <bool BWD>
DEV int cv_get_dky0(Cm<BWD> &m, double t, double *dky, int64_t dstride, int qoff_out)
{
    double tfuzz = FUZZ_FACTOR * UROUND * (fabs(m.tn) + fabs(m.hu));
    if (m.hu < 0.0) tfuzz = -tfuzz;
    double tp = m.tn - m.hu - tfuzz;
    double tn1 = m.tn + tfuzz;
    if ((t - tp) * (t - tn1) > 0.0) return CV_BAD_T;
    double s = (t - m.tn) / m.h;
    double pw[QMAX + 1];
    pw[0] = 1.0;
    SFOR(j, 1, (QMAX) + 1) pw[j] = pw[j - 1] * s; SEND
    const double pq = pick(pw, m.q);
    for (int i = 0; i < NS; i++) {
        double acc = pq * ZN(m, m.q, i);
        for (int j = m.q - 1; j >= 0; j--) acc = FMA(pick(pw, j), ZN(m, j, i), acc);
        dky[(int64_t)i * dstride] = acc;
    }
    if (BWD) {
        for (int i = 0; i < NQ; i++) {
            double acc = pq * ZNQ(m, m.q, i);
            for (int j = m.q - 1; j >= 0; j--) acc = FMA(pick(pw, j), ZNQ(m, j, i), acc);
            W(m, qoff_out, i) = acc;
        }
    }
    return CV_SUCCESS;
}

template <bool BWD>
DEV int cv_first_call(Cm<BWD> &m, double tout)
{
#ifdef SA_CONSTRAINTS
    if (!BWD && m.cons) {
        if (m.sensi && m.ism == 0) return CV_ILL_INPUT;       /* CVODES: no constraints with the simultaneous corrector */
        for (int i = 0; i < NS; i++) if (constr_violated(m.cons[i], ZN(m, 0, i))) return CV_ILL_INPUT;
    }
#endif
    if (ewt_set(m, O_ZN, O_EWT) != 0) return CV_ILL_INPUT;
    if (BWD) { if (ewtQ_set(m, O_ZNQ, O_EWTQ) != 0) return CV_ILL_INPUT; }
#ifdef SA_SENS
    if (m.sensi) { if (sens_ewt_set(m, O_ZNS, O_EWTS) != 0) return CV_ILL_INPUT; }
#endif
    if (BWD) { if (interp_y(m, m.tn) != CV_SUCCESS) { m.nfe++; return CV_RHSFUNC_FAIL; } }
    int retval = cv_f(m, m.tn, O_ZN, O_ZN + NS);
    if (retval < 0) return CV_RHSFUNC_FAIL;
    if (retval > 0) return CV_FIRST_RHSFUNC_ERR;
#ifdef SA_HERMITE
    if (!BWD) for (int i = 0; i < NS; i++) W(m, O_HY, 5 * NS + i) = ZN(m, 1, i);
#endif
#ifdef SA_SENS
    if (m.sensi) {
        retval = cv_fS(m, m.tn, O_ZN, O_ZNS, O_ZNS + NQ * NS);
        if (retval < 0) return CV_SRHSFUNC_FAIL;
        if (retval > 0) return CV_FIRST_SRHSFUNC_ERR;
    }
#endif
    if (BWD) {
        retval = cv_fQ(m, m.tn, O_ZN, O_ZNQ + NQ);
        if (retval < 0) return CV_QRHSFUNC_FAIL;
        if (retval > 0) return CV_FIRST_QRHSFUNC_ERR;
    }
    double tout_hin = tout;
    if (BWD) {
        if ((m.tstop - m.tn) * (tout - m.tn) <= 0.0) return CV_ILL_INPUT;
        if ((tout - m.tn) * (tout - m.tstop) > 0.0) tout_hin = m.tstop;
    }
    int hflag = cv_hin(m, tout_hin);
    if (hflag != CV_SUCCESS) return hflag;
    if (BWD) {
        if ((m.tn + m.h - m.tstop) * m.h > 0.0) m.h = (m.tstop - m.tn) * (1.0 - 4.0 * UROUND);
    }
    m.hscale = m.h;
    m.hprime = m.h;
    for (int i = 0; i < NS; i++) ZN(m, 1, i) = m.h * ZN(m, 1, i);
    if (BWD) for (int i = 0; i < NQ; i++) ZNQ(m, 1, i) = m.h * ZNQ(m, 1, i);
#ifdef SA_SENS
    if (m.sensi) for (int is = 0; is < NQ; is++) for (int i = 0; i < NS; i++) ZNS(m, 1, is, i) = m.h * ZNS(m, 1, is, i);
#endif
    return CV_SUCCESS;
}

template <bool BWD>
DEV int cv_pre_step(Cm<BWD> &m)
{
    if (ewt_set(m, O_ZN, O_EWT) != 0) return CV_ILL_INPUT;
    if (BWD) { if (ewtQ_set(m, O_ZNQ, O_EWTQ) != 0) return CV_ILL_INPUT; }
#ifdef SA_SENS
    if (m.sensi) { if (sens_ewt_set(m, O_ZNS, O_EWTS) != 0) return CV_ILL_INPUT; }
#endif
    double nrm = wrms_n(m, O_ZN);
    if (BWD) nrm = quad_update_norm(m, nrm, O_ZNQ);
#ifdef SA_SENS
    if (m.sensi) nrm = sens_update_norm(m, nrm, O_ZNS, O_EWTS);
#endif
    if (UROUND * nrm > 1.0) return CV_TOO_MUCH_ACC;
    return CV_SUCCESS;
}

struct StepCtl {
    int in_step, redo, nflag, ncf, nef, nefQ, convfail, ncfS, nefS;
    double saved_t;
};

template <bool BWD>
DEV int cv_handle_nflag_failed(Cm<BWD> &m, StepCtl &c, int nflag, int &ncf, int &ncfn)
{
    ncfn++;
    cv_restore(m, c.saved_t);
    if (nflag < 0) return nflag;
    ncf++;
    m.etamax = 1.0;
    if (ncf == MXNCF) {
        if (nflag == NLS_CONV_RECVR) return CV_CONV_FAILURE;
        if (nflag == RHSFUNC_RECVR) return CV_REPTD_RHSFUNC_ERR;
        if (nflag == SRHSFUNC_RECVR) return CV_REPTD_SRHSFUNC_ERR;
        if (nflag == CONSTR_RECVR) return CV_CONSTR_FAIL;
        return CV_REPTD_QRHSFUNC_ERR;
    }
    if (nflag != CONSTR_RECVR) m.eta = ETACF;         /* CONSTR_RECVR: eta was set by the constraint check */
    c.nflag = PREV_CONV_FAIL;
    cv_rescale(m);
    return 0;
}

template <bool BWD>
DEV int cv_attempt(Cm<BWD> &m, StepCtl &c)
{
    if (!c.in_step) {
        c.saved_t = m.tn;
        c.ncf = c.nef = c.nefQ = 0;
        c.ncfS = c.nefS = 0;
        c.nflag = FIRST_CALL;
        c.redo = 0;
        if ((m.nst > 0) && (m.hprime != m.h)) {
            if (m.qprime != m.q) {
                cv_adjust_order(m, m.qprime - m.q);
                m.q = m.qprime;
                m.L = m.q + 1;
                m.qwait = m.L;
            }
            cv_rescale(m);
        }
        c.in_step = 1;
    }
    int callSetup, jbad;
    if (!c.redo) {
        cv_predict(m);
        cv_set(m);
        if (BWD) { if (interp_y(m, m.tn) != CV_SUCCESS) { m.nfe++; return CV_RHSFUNC_FAIL; } }
        c.convfail = ((c.nflag == FIRST_CALL) || (c.nflag == PREV_ERR_FAIL)) ? CV_NO_FAILURES : CV_FAIL_OTHER;
        callSetup = (c.nflag == PREV_CONV_FAIL) || (c.nflag == PREV_ERR_FAIL) || (m.nst == 0) ||
                    (m.nst >= m.nstlp + MSBP) || (fabs(m.gamrat - 1.0) > DGMAX);
        jbad = 0;
    } else {
        callSetup = 1;
        jbad = 1;
    }
    int in_loop;
    int nls = cv_newton_pass(m, callSetup, jbad, c.convfail, in_loop);
    if ((nls > 0) && in_loop && !m.nls_jcur) {
        c.redo = 1;
        return 0;
    }
    c.redo = 0;
    if (nls != CV_SUCCESS) return cv_handle_nflag_failed(m, c, nls, c.ncf, m.ncfn);

    for (int i = 0; i < NS; i++) W(m, O_Y, i) = ZN(m, 0, i) + W(m, O_ACOR, i);
#ifdef SA_CONSTRAINTS
    if (!BWD && m.cons) {               /* cvCheckConstraints (see the oracle); mask in O_FTEMP, v in O_TEMPV */
        bool any = false;
        for (int i = 0; i < NS; i++) {
            const bool bad = constr_violated(m.cons[i], W(m, O_Y, i));
            W(m, O_FTEMP, i) = bad ? 1.0 : 0.0;
            any = any || bad;
        }
        if (any) {
            for (int i = 0; i < NS; i++) {
                const double aa = (fabs(m.cons[i]) >= 1.5) ? 1.0 : 0.0;
                double tmp = (aa * m.cons[i]) / W(m, O_EWT, i);
                tmp = FMA(-0.1, tmp, W(m, O_Y, i));
                W(m, O_TEMPV, i) = tmp * W(m, O_FTEMP, i);
            }
            const double vnorm = wrms_n(m, O_TEMPV);
            if (vnorm * m.tq[4] <= 1.0) {
                for (int i = 0; i < NS; i++) W(m, O_ACOR, i) = W(m, O_ACOR, i) - W(m, O_TEMPV, i);
            } else {
                double minq = 1e308;
                for (int i = 0; i < NS; i++) {
                    const double d = W(m, O_FTEMP, i) * (ZN(m, 0, i) - W(m, O_Y, i));
                    if (d != 0.0) { const double qv = ZN(m, 0, i) / d; if (qv < minq) minq = qv; }
                }
                m.eta = fmax(0.9 * minq, 0.1);
                return cv_handle_nflag_failed(m, c, CONSTR_RECVR, c.ncf, m.ncfn);
            }
        }
    }
#endif
    double dsm = m.acnrm * m.tq[2];
    if (dsm > 1.0) {
        c.nflag = PREV_ERR_FAIL;
        return cv_error_test_failed(m, c.saved_t, dsm, c.nef, m.netf);
    }
#ifdef SA_SENS
    if (m.sensi && m.ism == 0) {
        for (int is = 0; is < NQ; is++)
            for (int i = 0; i < NS; i++) VS(m, O_YS, is, i) = ZNS(m, 0, is, i) + VS(m, O_ACORS, is, i);
    }
    if (m.sensi && m.ism == 1) {         /* CV_STAGGERED: sensitivities after the state passed (oracle cv_step) */
        c.ncf = c.nef = 0;
        int retval = cv_f(m, m.tn, O_Y, O_FTEMP);
        if (retval < 0) return CV_RHSFUNC_FAIL;
        if (retval > 0) { c.nflag = PREV_CONV_FAIL; return 0; }
        const int nflagS = cv_stgr_nls(m);
        if (nflagS != CV_SUCCESS) return cv_handle_nflag_failed(m, c, nflagS, c.ncfS, m.ncfnS);
        m.acnrmS = sens_update_norm(m, 0.0, O_ACORS, O_EWTS);
        const double dsmS = m.acnrmS * m.tq[2];
        if (dsmS > 1.0) {
            c.nflag = PREV_ERR_FAIL;
            return cv_error_test_failed(m, c.saved_t, dsmS, c.nefS, m.netfS);
        }
        if (dsmS > dsm) dsm = dsmS;
    }
#endif
    if (BWD) {
        c.ncf = c.nef = 0;
        int retval = cv_fQ(m, m.tn, O_Y, O_ACORQ);
        if (retval != 0)
            return cv_handle_nflag_failed(m, c, retval < 0 ? CV_QRHSFUNC_FAIL : QRHSFUNC_RECVR, c.ncf, m.ncfn);
        for (int i = 0; i < NQ; i++) {
            double v = FMA(m.h, W(m, O_ACORQ, i), -ZNQ(m, 1, i));
            W(m, O_ACORQ, i) = m.rl1 * v;
        }
        double acnrmQ = wrms_q(m, O_ACORQ);
        double dsmQ = acnrmQ * m.tq[2];
        if (dsmQ > 1.0) {
            c.nflag = PREV_ERR_FAIL;
            return cv_error_test_failed(m, c.saved_t, dsmQ, c.nefQ, m.netfQ);
        }
        if (dsmQ > dsm) dsm = dsmQ;
    }
    cv_complete_step(m);
    cv_prepare_next_step(m, dsm);
    m.etamax = (m.nst <= SMALL_NST) ? ETAMX2 : ETAMX3;
    for (int i = 0; i < NS; i++) W(m, O_ACOR, i) = m.tq[2] * W(m, O_ACOR, i);
    if (BWD) for (int i = 0; i < NQ; i++) W(m, O_ACORQ, i) = m.tq[2] * W(m, O_ACORQ, i);
#ifdef SA_SENS
    if (m.sensi) for (int j = 0; j < NQ * NS; j++) W(m, O_ACORS, j) = m.tq[2] * W(m, O_ACORS, j);
#endif
    c.in_step = 0;
    return 1;
}

template <bool BWD>
DEV void accumulate_stats(const Cm<BWD> &m, int64_t *acc)
{
    acc[ST_NST] += m.nst; acc[ST_NFE] += m.nfe; acc[ST_NSETUPS] += m.nsetups; acc[ST_NJE] += m.nje;
    acc[ST_NNI] += m.nni; acc[ST_NCFN] += m.ncfn; acc[ST_NETF] += m.netf; acc[ST_QLAST] = m.qu;
    acc[ST_NFQE] += m.nfQe; acc[ST_NETFQ] += m.netfQ;
}

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
        r[(int64_t)(8 + i) * tS] = ZN(m, 0, i);
        r[(int64_t)(8 + NS + i) * tS] = first ? W(m, O_HY, 5 * NS + i) : (1.0 / m.h) * ZN(m, 1, i);
    }
}
#endif

/* ------------------------------------------------------------------------------------ */
extern "C" __global__ void __launch_bounds__(64) sa_k_forward(sa_fwd_args a)
{
    const int inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= a.B) return;
    Cm<false> m;
    m.sensi = 0; m.ism = 0; m.cons = a.constraints;
    m.S = a.ws_stride;
    m.w = a.ws + inst;
    SFOR(i, 0, NQ) m.ps[i] = a.ps[(int64_t)inst * NQ + i]; SEND
    m.pr = a.pr + (int64_t)inst * a.rem_stride;
    m.rtol = a.rtol; m.atol_p = a.atol; m.atol_s = 0.0;
    m.rtolQ = 0.0; m.atolQ = 0.0; m.tstop = 0.0;
    m.np = 0; m.tfinal = 0.0; m.ilast = 0; m.newdata = 0; m.have_last = 0; m.cur_idx = 0;
    m.last_t = 0.0; m.tlo = m.thi = m.tlo2 = 0.0; m.n_interp = 0; m.n_rebuild = 0;
    m.traj = nullptr; m.tS = 0;

    const double *y0 = a.y0 + (int64_t)inst * NS;
    for (int j = 0; j <= QMAX; j++) for (int i = 0; i < NS; i++) ZN(m, j, i) = 0.0;
    for (int i = 0; i < NS; i++) {
        ZN(m, 0, i) = y0[i];
        W(m, O_ACOR, i) = 0.0; W(m, O_TEMPV, i) = 0.0; W(m, O_FTEMP, i) = 0.0; W(m, O_Y, i) = 0.0;
    }
    cv_reinit(m, a.t0);

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
            for (int i = 0; i < NS; i++) W(m, O_HY, i) = ZN(m, 0, i);
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
                        for (int i = 0; i < NS; i++) W(m, O_HY, i) = ZN(m, 0, i);
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
                        cv_get_dky0(m, tout, yo + (int64_t)k * NS, 1, O_QOUT);
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
    m.sensi = 0; m.ism = 0; m.cons = nullptr;
    m.S = a.ws_stride;
    m.w = a.ws + inst;
    SFOR(i, 0, NQ) m.ps[i] = a.ps[(int64_t)inst * NQ + i]; SEND
    m.pr = a.pr + (int64_t)inst * a.rem_stride;
    m.rtol = a.rtolB; m.atol_s = a.atolB; m.atol_p = nullptr;
    m.rtolQ = a.rtolQB; m.atolQ = a.atolQB;
    m.tstop = a.tinitial;
    m.traj = a.traj + inst;
    m.tS = a.traj_stride;
    m.np = np;
    m.tfinal = (status == CV_SUCCESS) ? rec(m, np - 1, 2) : a.tinitial;
    m.cur_idx = 0; m.tlo2 = 0.0; m.tlo = m.thi = 0.0;
    m.ilast = 0; m.newdata = 1; m.have_last = 0; m.last_t = 0.0;
    m.n_interp = 0; m.n_rebuild = 0;

    for (int i = 0; i < NS; i++) { W(m, O_LAM, i) = 0.0; W(m, O_YTMP, i) = 0.0; }
    for (int i = 0; i < NQ; i++) { W(m, O_QUAD, i) = 0.0; W(m, O_QOUT, i) = 0.0; }
    for (int j = 0; j <= QMAX; j++) {
        for (int i = 0; i < NS; i++) ZN(m, j, i) = 0.0;
        for (int i = 0; i < NQ; i++) ZNQ(m, j, i) = 0.0;
    }
    for (int i = 0; i < NS; i++) { W(m, O_ACOR, i) = 0.0; W(m, O_TEMPV, i) = 0.0; W(m, O_FTEMP, i) = 0.0; W(m, O_Y, i) = 0.0; }
    for (int i = 0; i < NQ; i++) { W(m, O_ACORQ, i) = 0.0; W(m, O_TEMPVQ, i) = 0.0; }
    const double *g = a.grads + (int64_t)inst * a.grads_stride;
    bool first_call = true;
    int total_retries = 0, attempts = 0;
    cv_reinit(m, a.t0);

    for (int iv = 0; iv <= a.n_t; iv++) {
        const double t_upper = (iv == 0) ? a.t0 : a.tvals[a.n_t - iv];
        const double t_lower = (iv == a.n_t) ? a.tend : a.tvals[a.n_t - 1 - iv];
        if (t_lower < t_upper) {
            if (status == CV_SUCCESS) {
                for (int i = 0; i < NS; i++) ZN(m, 0, i) = W(m, O_LAM, i);        /* CVodeReInitB */
                for (int i = 0; i < NQ; i++) ZNQ(m, 0, i) = W(m, O_QUAD, i);      /* CVodeQuadReInitB */
                cv_reinit(m, t_upper);
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
                            cv_get_dky0(m, t_lower, &W(m, O_LAM, 0), m.S, O_QOUT);
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
            if (status == CV_SUCCESS) { for (int i = 0; i < NQ; i++) W(m, O_QUAD, i) = W(m, O_QOUT, i); }
        }
        if (iv < a.n_t && status == CV_SUCCESS) {
            const double *gi = g + (int64_t)(a.n_t - 1 - iv) * NS;
            for (int i = 0; i < NS; i++) W(m, O_LAM, i) -= gi[i];
            const int64_t row = (int64_t)inst * a.n_t + (iv == 0 ? 0 : a.n_t - iv);
            if (a.lamda_all) for (int i = 0; i < NS; i++) a.lamda_all[row * NS + i] = W(m, O_LAM, i);
            if (a.quad_all) for (int i = 0; i < NQ; i++) a.quad_all[row * NQ + i] = W(m, O_QUAD, i);
        }
    }
    if (status != CV_SUCCESS) {
        if (a.lamda_all) for (int j = 0; j < a.n_t * NS; j++) a.lamda_all[(int64_t)inst * a.n_t * NS + j] = SA_NAN;
        if (a.quad_all) for (int j = 0; j < a.n_t * NQ; j++) a.quad_all[(int64_t)inst * a.n_t * NQ + j] = SA_NAN;
    }
    for (int i = 0; i < NQ; i++) a.grad_out[(int64_t)inst * NQ + i] = (status == CV_SUCCESS) ? W(m, O_QOUT, i) : SA_NAN;
    for (int i = 0; i < NS; i++) a.lamda_out[(int64_t)inst * NS + i] = (status == CV_SUCCESS) ? W(m, O_LAM, i) : SA_NAN;
    a.status[inst] = status;
    st[ST_NPTS] = np; st[ST_NINTERP] = m.n_interp; st[ST_NREBUILD] = m.n_rebuild;
    st[ST_RETRIES] = total_retries; st[ST_ATTEMPTS] = attempts;
    SFOR(i, 0, SA_N_STATS) a.stats[(int64_t)inst * SA_N_STATS + i] = st[i]; SEND
}

#ifdef SA_SENS
/* CVodeGetSensDky(k = 0) for all parameters -> dst[is * NS + i] */
template <bool BWD>
DEV int cv_get_sens_dky0(Cm<BWD> &m, double t, double *dst)
{
    double tfuzz = FUZZ_FACTOR * UROUND * (fabs(m.tn) + fabs(m.hu));
    if (m.hu < 0.0) tfuzz = -tfuzz;
    double tp = m.tn - m.hu - tfuzz;
    double tn1 = m.tn + tfuzz;
    if ((t - tp) * (t - tn1) > 0.0) return CV_BAD_T;
    double s = (t - m.tn) / m.h;
    double pw[QMAX + 1];
    pw[0] = 1.0;
    SFOR(j, 1, (QMAX) + 1) pw[j] = pw[j - 1] * s; SEND
    const double pq = pick(pw, m.q);
    for (int is = 0; is < NQ; is++)
        for (int i = 0; i < NS; i++) {
            double acc = pq * ZNS(m, m.q, is, i);
            for (int j = m.q - 1; j >= 0; j--) acc = FMA(pick(pw, j), ZNS(m, j, is, i), acc);
            dst[is * NS + i] = acc;
        }
    return CV_SUCCESS;
}

/* Solver(sens_mode).solve (reference solver.py:467-527): CVodeReInit + CVodeSensReInit, then per output
   time CVode(NORMAL) with the mxstep x max_retries budget, CVodeGetSens */
extern "C" __global__ void __launch_bounds__(64) sa_k_sens(sa_sens_args a)
{
    const int inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= a.B) return;
    Cm<false> m;
    m.S = a.ws_stride;
    m.w = a.ws + inst;
    SFOR(i, 0, NQ) { m.ps[i] = a.ps[(int64_t)inst * NQ + i]; m.pbar[i] = a.pbar[i]; } SEND
    m.pr = a.pr + (int64_t)inst * a.rem_stride;
    m.rtol = a.rtol; m.atol_p = a.atol; m.atol_s = 0.0;
    m.rtolQ = 0.0; m.atolQ = 0.0; m.tstop = 0.0;
    m.np = 0; m.tfinal = 0.0; m.ilast = 0; m.newdata = 0; m.have_last = 0; m.cur_idx = 0;
    m.last_t = 0.0; m.tlo = m.thi = m.tlo2 = 0.0; m.n_interp = 0; m.n_rebuild = 0;
    m.traj = nullptr; m.tS = 0;
    m.sensi = 1; m.ism = a.ism; m.cons = nullptr;

    const double *y0 = a.y0 + (int64_t)inst * NS;
    const double *s0 = a.sens0 + (int64_t)inst * NQ * NS;
    for (int j = 0; j <= QMAX; j++) {
        for (int i = 0; i < NS; i++) ZN(m, j, i) = 0.0;
        for (int k = 0; k < NQ * NS; k++) W(m, O_ZNS, j * NQ * NS + k) = 0.0;
    }
    for (int i = 0; i < NS; i++) {
        ZN(m, 0, i) = y0[i];
        W(m, O_ACOR, i) = 0.0; W(m, O_TEMPV, i) = 0.0; W(m, O_FTEMP, i) = 0.0; W(m, O_Y, i) = 0.0;
    }
    for (int k = 0; k < NQ * NS; k++) {
        W(m, O_ZNS, k) = s0[k];
        W(m, O_ACORS, k) = 0.0; W(m, O_TEMPVS, k) = 0.0; W(m, O_FTEMPS, k) = 0.0; W(m, O_YS, k) = 0.0;
        W(m, O_EWTS, k) = 0.0;
    }
    cv_reinit(m, a.t0);

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
                        cv_get_dky0(m, tout, yo + (int64_t)k * NS, 1, O_QOUT);
                        cv_get_sens_dky0(m, tout, so + (int64_t)k * NQ * NS);
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
