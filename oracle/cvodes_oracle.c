/*
 * oracle/cvodes_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar, one instance at a time) of the algorithm the
 * reference's hot path runs:
 *
 *   sunode AdjointSolver.solve_forward   /root/reference/sunode/solver.py:682-721
 *   sunode AdjointSolver.solve_backward  /root/reference/sunode/solver.py:723-784
 *   sunode Solver.solve                  /root/reference/sunode/solver.py:467-527
 *   ... with forward sensitivities       /root/reference/sunode/solver.py:360-392 (CVodeSensInit, EE tolerances,
 *                                        errconS; simultaneous and staggered correctors)
 *   ... with inequality constraints      /root/reference/sunode/solver.py:230-233, 569-572 (CVodeSetConstraints)
 *   ... Hermite interpolation            /root/reference/sunode/solver.py:581-582 (CVodeAdjInit, CV_HERMITE)
 *   ... lamda_all_out / quad_all_out     /root/reference/sunode/solver.py:778-781
 *
 * all of whose arithmetic happens inside SUNDIALS CVODES 5.x (conda-forge
 * `sundials<6.0`, /root/reference/.github/workflows/main.yml:36), a third-party
 * dependency whose source is NOT under /root/reference (only declaration headers,
 * /root/reference/include/cvodes/16_cvodes.h).  The integrator below is therefore a
 * from-memory restatement of the published CVODES 5.x algorithm: variable-order
 * variable-step BDF(1-5) in Nordsieck form, SUNNonlinSol_Newton, SUNLinSol_Dense
 * (getrf/getrs with partial pivoting), WRMS norms, cvHin, CVodeGetDky, the adjoint
 * module's data-point store + Newton-form polynomial interpolation (CV_POLYNOMIAL),
 * backward quadratures with error control, tstop handling of CVodeB.
 *
 * PARITY STATUS: the reference's own tests pin no numeric result on this path and
 * CVODES cannot be built here, so parity with CVODES itself is UNPINNED at the
 * bit/step level.  This oracle is pinned instead by (tests/test_oracle_*.py):
 *   - scipy's DVODE (the Fortran ancestor of CVODE) step statistics + states,
 *   - tight-tolerance DOP853/Radau truth solutions + finite-difference gradients,
 *   - the reference notebook's printed known-answer (notebooks/from_sympy.ipynb:240-242),
 *   - sensitivity-equation truth for the forward sensitivities, analytic cases, the textbook Robertson
 *     blow-up for the constraints (tests/test_forward_sens.py, tests/test_oracle_pinning.py).
 *
 *
 * RULE FOR EDITS (binding): the fixtures under tests/golden/ that come from OTHER codes -- dvode_stats.json
 * (DVODE's own counters and step traces), truth_*.npz (DOP853/Radau + sensitivity equations), callbacks.json /
 * layout.json (the reference's own lambdify output) -- are IMMUTABLE.  This file has been edited together with the
 * kernels for bit-equality (explicit FMAs, balanced-tree WRMS sums, deterministic pow, reciprocal pivots); such
 * edits may only re-associate or re-round arithmetic.  An edit after which any DVODE counter or trace in
 * tests/test_oracle_pinning.py changes is WRONG and is rejected -- the fixture is never regenerated to follow
 * the oracle.  (Division sharing / reciprocal forms in cvSet and friends fall under this rule too.)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC \
 *            -DSA_PROBLEM_HEADER='"<generated header>"' cvodes_oracle.c -lm
 * The generated header (sunode_amd.symode.codegen) supplies SA_N_STATES, SA_N_SUB,
 * SA_N_REM and the callbacks sa_rhs / sa_jac / sa_adj_rhs / sa_quad_rhs / sa_adj_jac
 * (the role the numba cfuncs of /root/reference/sunode/problem.py:156-383 play).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SA_FN static inline
#include SA_PROBLEM_HEADER

/* Fused multiply-add is used explicitly (never by compiler contraction: -ffp-contract=off) at the
   same places as in the HIP kernel, so both sides round identically.  CVODES itself evaluates these
   a*b+c forms with two roundings; the difference is one ulp per operation. */
#define FMA(a, b, c) __builtin_fma((a), (b), (c))

#define NS SA_N_STATES
#define NQ SA_N_SUB
#define NR SA_N_REM
#define NSD (NS > 0 ? NS : 1)
#define NQD (NQ > 0 ? NQ : 1)

/* ---- CVODES return codes (/root/reference/include/cvodes/16_cvodes.h:45-106) ---- */
#define CV_SUCCESS 0
#define CV_TSTOP_RETURN 1
#define CV_TOO_MUCH_WORK (-1)
#define CV_TOO_MUCH_ACC (-2)
#define CV_ERR_FAILURE (-3)
#define CV_CONV_FAILURE (-4)
#define CV_LSETUP_FAIL (-6)
#define CV_LSOLVE_FAIL (-7)
#define CV_RHSFUNC_FAIL (-8)
#define CV_FIRST_RHSFUNC_ERR (-9)
#define CV_REPTD_RHSFUNC_ERR (-10)
#define CV_UNREC_RHSFUNC_ERR (-11)
#define CV_ILL_INPUT (-22)
#define CV_BAD_T (-25)
#define CV_TOO_CLOSE (-27)
#define CV_QRHSFUNC_FAIL (-31)
#define CV_FIRST_QRHSFUNC_ERR (-32)
#define CV_REPTD_QRHSFUNC_ERR (-33)
#define CV_SRHSFUNC_FAIL (-41)
#define CV_FIRST_SRHSFUNC_ERR (-42)
#define CV_REPTD_SRHSFUNC_ERR (-43)
#define CV_UNREC_SRHSFUNC_ERR (-44)
#define CV_BAD_TB0 (-104)
#define CV_GETY_BADT (-107)

/* ---- CVODES constants (cvodes.c / cvodes_impl.h / cvodes_ls.c, 5.x) ---- */
#define QMAX 5
#define LMAXP 7
#define UROUND 2.220446049250313e-16
#define ETAMX1 10000.0
#define ETAMX2 10.0
#define ETAMX3 10.0
#define ETAMXF 0.2
#define ETAMIN 0.1
#define ETACF 0.25
#define ADDON 0.000001
#define BIAS1 6.0
#define BIAS2 6.0
#define BIAS3 10.0
#define ONEPSM 1.000001
#define THRESH 1.5
#define MXNCF 10
#define MXNEF 7
#define MXNEF1 3
#define SMALL_NEF 2
#define LONG_WAIT 10
#define SMALL_NST 10
#define NLS_MAXCOR 3
#define CRDOWN 0.3
#define DGMAX 0.3
#define RDIV 2.0
#define MSBP 20
#define NLSCOEF 0.1
#define MSBJ 50
#define CVLS_DGMAX 0.2
#define HLB_FACTOR 100.0
#define HUB_FACTOR 0.1
#define H_BIAS 0.5
#define HIN_MAX_ITERS 4
#define FUZZ_FACTOR 100.0
#define FUZZ_FACTOR_ADJ 1000000.0

/* internal flags */
#define FIRST_CALL 101
#define PREV_CONV_FAIL 102
#define PREV_ERR_FAIL 103
#define DO_ERROR_TEST 2
#define PREDICT_AGAIN 3
#define TRY_AGAIN 5
#define RHSFUNC_RECVR 9
#define QRHSFUNC_RECVR 11
#define SRHSFUNC_RECVR 12
#define CONSTR_RECVR 10
#define CV_CONSTR_FAIL (-15)
#define NLS_CONTINUE 901
#define NLS_CONV_RECVR 902
#define CV_NO_FAILURES 0
#define CV_FAIL_BAD_J 1
#define CV_FAIL_OTHER 2

/* statistics slots (int64 per instance) -- same order as include/sunode_amd.h */
enum { ST_NST, ST_NFE, ST_NSETUPS, ST_NJE, ST_NNI, ST_NCFN, ST_NETF, ST_QLAST,
       ST_NPTS, ST_NFQE, ST_NETFQ, ST_NINTERP, ST_NREBUILD, ST_RETRIES, ST_RESERVED0, ST_RESERVED1,
       ST_COUNT };

/* ------------------------------------------------------------------------- */
/* Deterministic pow: pure IEEE +,-,*,/ so that the HIP kernel (which restates  */
/* the same operation sequence) produces bit-identical step-size factors.       */
/* CVODES calls libm pow() here (SUNRpowerR); accuracy needed is ~1e-3.          */
/* ------------------------------------------------------------------------- */
static double det_log(double x)
{
    union { double d; uint64_t u; } v;
    v.d = x;
    int e = (int)((v.u >> 52) & 0x7ff);
    if (e == 0) { v.d = x * 18014398509481984.0; e = (int)((v.u >> 52) & 0x7ff) - 54; }
    e -= 1023;
    v.u = (v.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = v.d;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 1.0 / 23.0;
    p = FMA(p, z, 1.0 / 21.0);
    p = FMA(p, z, 1.0 / 19.0);
    p = FMA(p, z, 1.0 / 17.0);
    p = FMA(p, z, 1.0 / 15.0);
    p = FMA(p, z, 1.0 / 13.0);
    p = FMA(p, z, 1.0 / 11.0);
    p = FMA(p, z, 1.0 / 9.0);
    p = FMA(p, z, 1.0 / 7.0);
    p = FMA(p, z, 1.0 / 5.0);
    p = FMA(p, z, 1.0 / 3.0);
    p = FMA(p, z, 1.0);
    return FMA((double)e, 0.6931471805599453, 2.0 * s * p);
}

static double det_exp(double w)
{
    if (w > 700.0) w = 700.0;
    if (w < -700.0) w = -700.0;
    double kf = floor(FMA(w, 1.4426950408889634, 0.5));
    double r = FMA(-kf, 1.90821492927058770002e-10, FMA(-kf, 0.693147180369123816490, w));
    double p = 1.0 / 6227020800.0;
    p = FMA(p, r, 1.0 / 479001600.0);
    p = FMA(p, r, 1.0 / 39916800.0);
    p = FMA(p, r, 1.0 / 3628800.0);
    p = FMA(p, r, 1.0 / 362880.0);
    p = FMA(p, r, 1.0 / 40320.0);
    p = FMA(p, r, 1.0 / 5040.0);
    p = FMA(p, r, 1.0 / 720.0);
    p = FMA(p, r, 1.0 / 120.0);
    p = FMA(p, r, 1.0 / 24.0);
    p = FMA(p, r, 1.0 / 6.0);
    p = FMA(p, r, 0.5);
    p = FMA(p, r, 1.0);
    p = FMA(p, r, 1.0);
    union { double d; uint64_t u; } v;
    v.u = (uint64_t)((int64_t)kf + 1023) << 52;
    return p * v.d;
}

/* SUNRpowerR */
static double rpower_r(double base, double expo)
{
    if (base <= 0.0) return 0.0;
    return det_exp(expo * det_log(base));
}

/* SUNRpowerI */
static double rpower_i(double base, int expo)
{
    double prod = 1.0;
    int n = expo < 0 ? -expo : expo;
    for (int i = 1; i <= n; i++) prod *= base;
    if (expo < 0) prod = 1.0 / prod;
    return prod;
}

double orc_det_pow(double x, double y) { return rpower_r(x, y); }

/* ------------------------------------------------------------------------- */
/* stored forward trajectory (CVODES adjoint "data points", CV_POLYNOMIAL)      */
/* ------------------------------------------------------------------------- */
typedef struct {
    int np, cap;
    double *t;
    double *y;          /* [np][NS] */
    double *yd;         /* [np][NS] y'(t) at the point (CV_HERMITE only) */
    int *order;
    int hermite;        /* interpolation type: 0 = CV_POLYNOMIAL, 1 = CV_HERMITE */
    double tinitial, tfinal;
    /* interpolation cache (ca_mem->ca_ilast, ca_IMnewData, ca_T, ca_Y) */
    int ilast, newdata;
    double T[QMAX + 1];
    double Y[QMAX + 1][NSD];
    /* the backward wrappers (cvArhs, cvArhsQ, cvLsJacBWrapper) each re-interpolate y(t); at an
       unchanged t that is the same value again, so it is evaluated once (as the HIP kernel does) */
    int have_last;
    double last_t, last_y[NSD];
    long n_interp, n_rebuild;
} traj_t;

static int traj_push(traj_t *tr, double t, const double *y, const double *yd, int order, int max_pts)
{
    if (max_pts > 0 && tr->np >= max_pts) return -1;
    if (tr->np == tr->cap) {
        int ncap = tr->cap ? 2 * tr->cap : 128;
        tr->t = (double *)realloc(tr->t, sizeof(double) * ncap);
        tr->y = (double *)realloc(tr->y, sizeof(double) * ncap * NSD);
        tr->yd = (double *)realloc(tr->yd, sizeof(double) * ncap * NSD);
        tr->order = (int *)realloc(tr->order, sizeof(int) * ncap);
        tr->cap = ncap;
    }
    tr->t[tr->np] = t;
    for (int i = 0; i < NS; i++) tr->y[(size_t)tr->np * NSD + i] = y[i];
    for (int i = 0; i < NS; i++) tr->yd[(size_t)tr->np * NSD + i] = yd ? yd[i] : 0.0;
    tr->order[tr->np] = order;
    tr->np++;
    return 0;
}

/* CVAfindIndex (forward direction of integration assumed: tfinal > tinitial) */
static int traj_find_index(traj_t *tr, double t, int *indx, int *newpoint)
{
    *newpoint = 0;
    if (tr->newdata) { tr->ilast = tr->np - 1; *newpoint = 1; tr->newdata = 0; }
    int ilast = tr->ilast;
    int to_left = (t - tr->t[ilast - 1]) < 0.0;
    int to_right = (t - tr->t[ilast]) > 0.0;
    if (to_left) {
        *newpoint = 1;
        *indx = ilast;
        for (;;) {
            if (*indx == 0) break;
            if ((t - tr->t[*indx - 1]) <= 0.0) (*indx)--;
            else break;
        }
        tr->ilast = (*indx == 0) ? 1 : *indx;
        if (*indx == 0) {
            if (fabs(t - tr->t[0]) > FUZZ_FACTOR_ADJ * UROUND) return CV_GETY_BADT;
        }
    } else if (to_right) {
        *newpoint = 1;
        *indx = ilast;
        for (;;) {
            if (*indx >= tr->np - 1) break;      /* guard: CVODES relies on t <= tfinal */
            if ((t - tr->t[*indx]) > 0.0) (*indx)++;
            else break;
        }
        if ((t - tr->t[*indx]) > FUZZ_FACTOR_ADJ * UROUND * (fabs(tr->tfinal) + 1.0)) return CV_GETY_BADT;
        tr->ilast = *indx;
    } else {
        *indx = ilast;
    }
    return CV_SUCCESS;
}

/* CVApolynomialGetY */
static int traj_get_y(traj_t *tr, double t, double *y)
{
    int indx, newpoint;
    if (tr->have_last && t == tr->last_t) {
        for (int i = 0; i < NS; i++) y[i] = tr->last_y[i];
        return CV_SUCCESS;
    }
    tr->n_interp++;
    int flag = traj_find_index(tr, t, &indx, &newpoint);
    if (flag != CV_SUCCESS) return flag;
    tr->have_last = 1;
    tr->last_t = t;
    if (indx == 0) {
        for (int i = 0; i < NS; i++) y[i] = tr->last_y[i] = tr->y[i];
        return CV_SUCCESS;
    }
    if (tr->hermite) {
        /* CVAhermiteGetY: cubic Hermite on [t0, t1] = [t[indx-1], t[indx]] from y, y' at both ends;
           Y[0] = y1 - y0 - delta y0',  Y[1] = delta (y1' + y0') - 2 (y1 - y0), rebuilt when the index moves */
        const double t0 = tr->t[indx - 1], t1 = tr->t[indx];
        const double delta = t1 - t0;
        const double *y0 = tr->y + (size_t)(indx - 1) * NSD, *yd0 = tr->yd + (size_t)(indx - 1) * NSD;
        if (newpoint) {
            tr->n_rebuild++;
            const double *y1 = tr->y + (size_t)indx * NSD, *yd1 = tr->yd + (size_t)indx * NSD;
            for (int i = 0; i < NS; i++) {
                const double dy = y1[i] - y0[i];
                tr->Y[0][i] = FMA(-delta, yd0[i], dy);
                tr->Y[1][i] = FMA(delta, yd1[i] + yd0[i], -2.0 * dy);
            }
        }
        const double factor1 = t - t0;
        double factor2 = factor1 / delta;
        factor2 = factor2 * factor2;
        const double factor3 = factor2 * (t - t1) / delta;
        for (int i = 0; i < NS; i++) {
            double acc = FMA(factor1, yd0[i], y0[i]);
            acc = FMA(factor2, tr->Y[0][i], acc);
            acc = FMA(factor3, tr->Y[1][i], acc);
            y[i] = tr->last_y[i] = acc;
        }
        return CV_SUCCESS;
    }
    double dt = fabs(tr->t[indx] - tr->t[indx - 1]);
    int base = indx;
    int order = tr->order[base];
    if (indx < order) base += order - indx;
    if (newpoint) {
        tr->n_rebuild++;
        for (int j = 0; j <= order; j++) {
            tr->T[j] = tr->t[base - j];
            for (int i = 0; i < NS; i++) tr->Y[j][i] = tr->y[(size_t)(base - j) * NSD + i];
        }
        for (int i = 1; i <= order; i++) {
            for (int j = order; j >= i; j--) {
                double factor = dt / (tr->T[j] - tr->T[j - i]);
                for (int k = 0; k < NS; k++)
                    tr->Y[j][k] = factor * (tr->Y[j][k] - tr->Y[j - 1][k]);
            }
        }
    }
    /* CVODES divides by dt per term; one reciprocal per call (as the HIP kernel does) differs
       from that only in the last bits of an interpolated y(t) */
    double cvals[QMAX + 1];
    const double inv_dt = 1.0 / dt;
    cvals[0] = 1.0;
    for (int i = 0; i < order; i++) cvals[i + 1] = cvals[i] * (t - tr->T[i]) * inv_dt;
    for (int k = 0; k < NS; k++) {
        double acc = cvals[0] * tr->Y[0][k];
        for (int i = 1; i <= order; i++) acc = FMA(cvals[i], tr->Y[i][k], acc);
        y[k] = tr->last_y[k] = acc;
    }
    return CV_SUCCESS;
}

/* ------------------------------------------------------------------------- */
/* integrator memory                                                           */
/* ------------------------------------------------------------------------- */
typedef struct {
    /* problem data */
    const double *ps, *pr;
    int backward;            /* 0: forward ODE, 1: adjoint ODE + quadrature */
    traj_t *tr;              /* backward: interpolation source; forward: store target (may be NULL) */
    /* tolerances */
    double rtol, atol[NSD];
    int quadr, errconQ;
    double rtolQ, atolQ;
    int mxstep;
    /* Nordsieck state */
    double zn[QMAX + 1][NSD];
    double znQ[QMAX + 1][NQD];
    double ewt[NSD], acor[NSD], tempv[NSD], ftemp[NSD], y[NSD];
    double ewtQ[NQD], acorQ[NQD], tempvQ[NQD], yQ[NQD];
    double ytmp[NSD];        /* interpolated forward state in the backward wrappers */
    double tn, h, hprime, hscale, eta, etamax, hu, next_h;
    int q, qprime, L, qwait, qu, next_q;
    double tau[LMAXP], tq[6], l[LMAXP];
    double rl1, gamma, gammap, gamrat, crate, delp, acnrm, saved_tq5, tolsf;
    double etaq, etaqm1, etaqp1;
    int tstopset;
    double tstop, tretlast;
    long nst, nfe, nje, nsetups, nni, ncfn, netf, nfQe, netfQ, nstlp, nstlj;
    /* inequality constraints (CVodeSetConstraints, solver.py:230-233, 569-572) */
    int constraints_set;
    double constraints[NSD];
    /* forward sensitivities (CVodeSensInit, EE tolerances, errconS = 1; solver.py:360-392) */
    int sensi, ism;                          /* ism: 0 = CV_SIMULTANEOUS, 1 = CV_STAGGERED */
    double pbar[NQD];
    double znS[QMAX + 1][NQD][NSD];
    double ewtS[NQD][NSD], acorS[NQD][NSD], tempvS[NQD][NSD], ftempS[NQD][NSD], yS[NQD][NSD];
    double Jtmp[NSD * NSD];                  /* Jacobian of the sensitivity right-hand side */
    double crateS, delpS, acnrmS;
    long nfSe, nniS, ncfnS, netfS, nsetupsS;
    /* linear solver */
    double A[NSD * NSD], savedJ[NSD * NSD];
    int piv[NSD];
    double inv_piv[NSD];
    int jcur, nls_jcur;
} cvmem;

/* ---- callbacks as CVODES would see them (forward: user fns; backward: cvArhs etc.) ---- */
static int cv_f(cvmem *m, double t, const double *y, double *out)
{
    m->nfe++;
    if (!m->backward) return sa_rhs(t, y, m->ps, m->pr, out);
    if (traj_get_y(m->tr, t, m->ytmp) != CV_SUCCESS) return -1;
    return sa_adj_rhs(t, m->ytmp, y, m->ps, m->pr, out);
}

static int cv_fQ(cvmem *m, double t, const double *y, double *out)
{
    m->nfQe++;
    if (traj_get_y(m->tr, t, m->ytmp) != CV_SUCCESS) return -1;
    return sa_quad_rhs(t, m->ytmp, y, m->ps, m->pr, out);
}

static int cv_jac(cvmem *m, double t, const double *y, double *J)
{
    if (!m->backward) return sa_jac(t, y, m->ps, m->pr, J);
    if (traj_get_y(m->tr, t, m->ytmp) != CV_SUCCESS) return -1;
    return sa_adj_jac(t, m->ytmp, m->ps, m->pr, J);
}

/* Sensitivity right-hand side for all parameters at once (symode/problem.py:557-583):
   out[is] = J(t,y) yS[is] + df/dp_is.  The reference leaves the order of the dot products to BLAS;
   here: left-to-right FMA chain over j, then the explicit part is added. */
static int cv_fS(cvmem *m, double t, const double *y, double yS[NQD][NSD], double out[NQD][NSD])
{
    m->nfSe++;
    double dp[NQD * NSD];
    for (int i = 0; i < NS * NS; i++) m->Jtmp[i] = 0.0;
    int rc = sa_jac(t, y, m->ps, m->pr, m->Jtmp);
    if (rc != 0) return rc;
    rc = sa_dydp(t, y, m->ps, m->pr, dp);
    int bad = 0;
    for (int is = 0; is < NQ; is++)
        for (int i = 0; i < NS; i++) {
            double acc = m->Jtmp[(size_t)0 * NS + i] * yS[is][0];
            for (int j = 1; j < NS; j++) acc = FMA(m->Jtmp[(size_t)j * NS + i], yS[is][j], acc);
            acc = acc + dp[is * NS + i];
            out[is][i] = acc;
            bad |= !(acc * 0.0 == 0.0);
        }
    return (rc != 0 || bad) ? 1 : 0;
}

/* ---- vector kernels ---- */
/* WRMS norm.  The sum of squares is taken as a balanced binary tree over the components padded with
   zeros to the next power of two (leaf i = (x_i w_i)^2) instead of CVODES' left-to-right loop: that
   is the association a cross-lane butterfly reduction produces, so the cooperative HIP kernel (one
   lane per component) and the thread-per-instance kernel round identically to this oracle. */
static double wrms(const double *x, const double *w, int n)
{
    if (n == 0) return 0.0;
    int P = 1;
    while (P < n) P <<= 1;
    double buf[P];
    for (int i = 0; i < P; i++) {
        double prod = (i < n) ? x[i] * w[i] : 0.0;
        buf[i] = prod * prod;
    }
    for (int m = 1; m < P; m <<= 1)
        for (int i = 0; i < P; i += 2 * m) buf[i] = buf[i] + buf[i + m];
    return sqrt(buf[0] / n);
}

static double quad_update_norm(cvmem *m, double old_nrm, const double *xQ, const double *wQ)
{
    double qnrm = wrms(xQ, wQ, NQ);
    return old_nrm > qnrm ? old_nrm : qnrm;
}

static int ewt_set(cvmem *m, const double *ycur, double *w)
{
    for (int i = 0; i < NS; i++) {
        double v = FMA(m->rtol, fabs(ycur[i]), m->atol[i]);
        if (v <= 0.0) return -1;
        w[i] = 1.0 / v;
    }
    return 0;
}

static int ewtQ_set(cvmem *m, const double *qcur, double *w)
{
    for (int i = 0; i < NQ; i++) {
        double v = FMA(m->rtolQ, fabs(qcur[i]), m->atolQ);
        if (v <= 0.0) return -1;
        w[i] = 1.0 / v;
    }
    return 0;
}

/* cvSensEwtSetEE: weights of pbar*yS with the state tolerances, scaled back by pbar */
static int sens_ewt_set(cvmem *m, double yScur[NQD][NSD], double w[NQD][NSD])
{
    for (int is = 0; is < NQ; is++) {
        const double pb = m->pbar[is];
        for (int i = 0; i < NS; i++) {
            double v = FMA(m->rtol, fabs(pb * yScur[is][i]), m->atol[i]);
            if (v <= 0.0) return -1;
            w[is][i] = pb * (1.0 / v);
        }
    }
    return 0;
}

/* cvSensUpdateNorm / N_VWrmsNorm of a sensitivity wrapper: max over the vectors */
static double sens_update_norm(double old_nrm, double xS[NQD][NSD], double wS[NQD][NSD])
{
    double nrm = old_nrm;
    for (int is = 0; is < NQ; is++) {
        double snrm = wrms(xS[is], wS[is], NS);
        if (snrm > nrm) nrm = snrm;
    }
    return nrm;
}

/* ---- dense LU (SUNDIALS denseGETRF / denseGETRS, column-major) ---- */
static int dense_getrf(double *a, int n, int *p, double *inv_piv)
{
    for (int k = 0; k < n; k++) {
        double *col_k = a + (size_t)k * n;
        int l = k;
        for (int i = k + 1; i < n; i++)
            if (fabs(col_k[i]) > fabs(col_k[l])) l = i;
        p[k] = l;
        if (col_k[l] == 0.0) return k + 1;
        if (l != k) {
            for (int i = 0; i < n; i++) {
                double tmp = a[(size_t)i * n + l];
                a[(size_t)i * n + l] = a[(size_t)i * n + k];
                a[(size_t)i * n + k] = tmp;
            }
        }
        double mult = 1.0 / col_k[k];
        inv_piv[k] = mult;           /* reused by the solves instead of dividing by the pivot again */
        for (int i = k + 1; i < n; i++) col_k[i] *= mult;
        for (int j = k + 1; j < n; j++) {
            double *col_j = a + (size_t)j * n;
            double a_kj = col_j[k];
            if (a_kj != 0.0)
                for (int i = k + 1; i < n; i++) col_j[i] = FMA(-a_kj, col_k[i], col_j[i]);
        }
    }
    return 0;
}

static void dense_getrs(const double *a, int n, const int *p, const double *inv_piv, double *b)
{
    for (int k = 0; k < n; k++) {
        int pk = p[k];
        if (pk != k) { double tmp = b[k]; b[k] = b[pk]; b[pk] = tmp; }
    }
    for (int k = 0; k < n - 1; k++) {
        const double *col_k = a + (size_t)k * n;
        for (int i = k + 1; i < n; i++) b[i] = FMA(-col_k[i], b[k], b[i]);
    }
    for (int k = n - 1; k > 0; k--) {
        const double *col_k = a + (size_t)k * n;
        b[k] *= inv_piv[k];
        for (int i = 0; i < k; i++) b[i] = FMA(-col_k[i], b[k], b[i]);
    }
    if (n > 0) b[0] *= inv_piv[0];
}

/* ------------------------------------------------------------------------- */
/* CVodeInit / CVodeReInit                                                     */
/* ------------------------------------------------------------------------- */
static void cv_reinit(cvmem *m, double t0, const double *y0, const double *q0)
{
    m->tn = t0;
    m->q = 1; m->L = 2; m->qwait = m->L; m->etamax = ETAMX1;
    m->qu = 0; m->hu = 0.0; m->tolsf = 1.0;
    for (int i = 0; i < NS; i++) m->zn[0][i] = y0[i];
    if (m->quadr) for (int i = 0; i < NQ; i++) m->znQ[0][i] = q0[i];
    m->nst = m->nfe = m->ncfn = m->netf = m->nni = m->nsetups = 0;
    m->nje = 0; m->nstlp = 0; m->nstlj = 0; m->nfQe = m->netfQ = 0;
    m->h = 0.0; m->next_h = 0.0; m->next_q = 0;
    m->hprime = 0.0; m->hscale = 0.0; m->eta = 1.0;
    m->qprime = 1;
    m->gamma = m->gammap = 0.0; m->gamrat = 1.0; m->crate = 1.0; m->delp = 0.0;
    m->acnrm = 0.0; m->saved_tq5 = 0.0;
    m->jcur = 0; m->nls_jcur = 0;
    for (int i = 0; i < LMAXP; i++) { m->tau[i] = 0.0; m->l[i] = 0.0; }
    for (int i = 0; i < 6; i++) m->tq[i] = 0.0;
    m->tretlast = t0;
}

/* CVodeSensReInit: znS[0] = yS0, counters */
static void cv_sens_reinit(cvmem *m, const double *yS0 /* [NQ][NS] */)
{
    for (int is = 0; is < NQ; is++)
        for (int i = 0; i < NS; i++) m->znS[0][is][i] = yS0[is * NS + i];
    m->nfSe = m->nniS = m->ncfnS = m->netfS = m->nsetupsS = 0;
    m->crateS = 1.0; m->delpS = 0.0; m->acnrmS = 0.0;
}

/* ------------------------------------------------------------------------- */
/* cvHin and helpers                                                           */
/* ------------------------------------------------------------------------- */
static double cv_upper_bound_h0(cvmem *m, double tdist)
{
    double hub_inv = 0.0;
    {
        double temp1[NSD];
        ewt_set(m, m->zn[0], temp1);
        for (int i = 0; i < NS; i++) {
            double t2 = fabs(m->zn[0][i]);
            double t1 = 1.0 / temp1[i];
            t1 = FMA(HUB_FACTOR, t2, t1);
            double v = fabs(m->zn[1][i]) / t1;
            if (v > hub_inv) hub_inv = v;
        }
    }
    if (m->quadr && m->errconQ) {
        double tempQ[NQD];
        ewtQ_set(m, m->znQ[0], tempQ);
        double hubQ_inv = 0.0;
        for (int i = 0; i < NQ; i++) {
            double t2 = fabs(m->znQ[0][i]);
            double t1 = 1.0 / tempQ[i];
            t1 = FMA(HUB_FACTOR, t2, t1);
            double v = fabs(m->znQ[1][i]) / t1;
            if (v > hubQ_inv) hubQ_inv = v;
        }
        if (hubQ_inv > hub_inv) hub_inv = hubQ_inv;
    }
    if (m->sensi) {          /* errconS */
        double wS[NQD][NSD];
        sens_ewt_set(m, m->znS[0], wS);
        for (int is = 0; is < NQ; is++)
            for (int i = 0; i < NS; i++) {
                double t2 = fabs(m->znS[0][is][i]);
                double t1 = 1.0 / wS[is][i];
                t1 = FMA(HUB_FACTOR, t2, t1);
                double v = fabs(m->znS[1][is][i]) / t1;
                if (v > hub_inv) hub_inv = v;
            }
    }
    double hub = HUB_FACTOR * tdist;
    if (hub * hub_inv > 1.0) hub = 1.0 / hub_inv;
    return hub;
}

static int cv_ydd_norm(cvmem *m, double hg, double *yddnrm)
{
    for (int i = 0; i < NS; i++) m->y[i] = FMA(hg, m->zn[1][i], m->zn[0][i]);
    if (m->sensi)
        for (int is = 0; is < NQ; is++)
            for (int i = 0; i < NS; i++) m->yS[is][i] = FMA(hg, m->znS[1][is][i], m->znS[0][is][i]);
    int retval = cv_f(m, m->tn + hg, m->y, m->tempv);
    if (retval < 0) return CV_RHSFUNC_FAIL;
    if (retval > 0) return RHSFUNC_RECVR;
    if (m->sensi) {
        retval = cv_fS(m, m->tn + hg, m->y, m->yS, m->tempvS);
        if (retval < 0) return CV_SRHSFUNC_FAIL;
        if (retval > 0) return SRHSFUNC_RECVR;
    }
    if (m->quadr && m->errconQ) {
        retval = cv_fQ(m, m->tn + hg, m->y, m->tempvQ);
        if (retval < 0) return CV_QRHSFUNC_FAIL;
        if (retval > 0) return QRHSFUNC_RECVR;
    }
    for (int i = 0; i < NS; i++) {
        m->tempv[i] = m->tempv[i] - m->zn[1][i];
        m->tempv[i] = (1.0 / hg) * m->tempv[i];
    }
    *yddnrm = wrms(m->tempv, m->ewt, NS);
    if (m->quadr && m->errconQ) {
        for (int i = 0; i < NQ; i++) {
            m->tempvQ[i] = m->tempvQ[i] - m->znQ[1][i];
            m->tempvQ[i] = (1.0 / hg) * m->tempvQ[i];
        }
        *yddnrm = quad_update_norm(m, *yddnrm, m->tempvQ, m->ewtQ);
    }
    if (m->sensi) {
        for (int is = 0; is < NQ; is++)
            for (int i = 0; i < NS; i++) {
                m->tempvS[is][i] = m->tempvS[is][i] - m->znS[1][is][i];
                m->tempvS[is][i] = (1.0 / hg) * m->tempvS[is][i];
            }
        *yddnrm = sens_update_norm(*yddnrm, m->tempvS, m->ewtS);
    }
    return CV_SUCCESS;
}

static int cv_hin(cvmem *m, double tout)
{
    double tdiff = tout - m->tn;
    if (tdiff == 0.0) return CV_TOO_CLOSE;
    int sign = (tdiff > 0.0) ? 1 : -1;
    double tdist = fabs(tdiff);
    double tround = UROUND * fmax(fabs(m->tn), fabs(tout));
    if (tdist < 2.0 * tround) return CV_TOO_CLOSE;

    double hlb = HLB_FACTOR * tround;
    double hub = cv_upper_bound_h0(m, tdist);
    double hg = sqrt(hlb * hub);
    if (hub < hlb) {
        m->h = (sign == -1) ? -hg : hg;
        return CV_SUCCESS;
    }
    int hnewOK = 0;
    double hs = hg, hnew = hg, yddnrm = 0.0;
    (void)hnewOK;
    for (int count1 = 1; count1 <= HIN_MAX_ITERS; count1++) {
        int hgOK = 0;
        for (int count2 = 1; count2 <= HIN_MAX_ITERS; count2++) {
            double hgs = hg * sign;
            int retval = cv_ydd_norm(m, hgs, &yddnrm);
            if (retval < 0) return CV_RHSFUNC_FAIL;
            if (retval == CV_SUCCESS) { hgOK = 1; break; }
            hg *= 0.2;
        }
        if (!hgOK) {
            if (count1 <= 2) return CV_REPTD_RHSFUNC_ERR;
            hnew = hs;
            break;
        }
        hs = hg;
        hnew = (yddnrm * hub * hub > 2.0) ? sqrt(2.0 / yddnrm) : sqrt(hg * hub);
        if (count1 == HIN_MAX_ITERS) break;
        double hrat = hnew / hg;
        if ((hrat > 0.5) && (hrat < 2.0)) break;
        if ((count1 > 1) && (hrat > 2.0)) { hnew = hg; break; }
        hg = hnew;
    }
    double h0 = H_BIAS * hnew;
    if (h0 < hlb) h0 = hlb;
    if (h0 > hub) h0 = hub;
    if (sign == -1) h0 = -h0;
    m->h = h0;
    return CV_SUCCESS;
}

/* ------------------------------------------------------------------------- */
/* step machinery                                                              */
/* ------------------------------------------------------------------------- */
static void cv_rescale(cvmem *m)
{
    double factor = m->eta;
    for (int j = 1; j <= m->q; j++) {
        for (int i = 0; i < NS; i++) m->zn[j][i] *= factor;
        if (m->quadr) for (int i = 0; i < NQ; i++) m->znQ[j][i] *= factor;
        if (m->sensi)
            for (int is = 0; is < NQ; is++) for (int i = 0; i < NS; i++) m->znS[j][is][i] *= factor;
        factor *= m->eta;
    }
    m->h = m->hscale * m->eta;
    m->next_h = m->h;
    m->hscale = m->h;
}

static void cv_increase_bdf(cvmem *m)
{
    for (int i = 0; i <= QMAX; i++) m->l[i] = 0.0;
    double alpha1 = 1.0, prod = 1.0, xiold = 1.0, alpha0 = -1.0, hsum = m->hscale;
    m->l[2] = 1.0;
    if (m->q > 1) {
        for (int j = 1; j < m->q; j++) {
            hsum += m->tau[j + 1];
            double xi = hsum / m->hscale;
            prod *= xi;
            alpha0 -= 1.0 / (j + 1);
            alpha1 += 1.0 / xi;
            for (int i = j + 2; i >= 2; i--) m->l[i] = FMA(m->l[i], xiold, m->l[i - 1]);
            xiold = xi;
        }
    }
    double A1 = (-alpha0 - alpha1) / prod;
    int L = m->L;
    for (int i = 0; i < NS; i++) m->zn[L][i] = A1 * m->zn[QMAX][i];
    for (int j = 2; j <= m->q; j++)
        for (int i = 0; i < NS; i++) m->zn[j][i] = FMA(m->l[j], m->zn[L][i], m->zn[j][i]);
    if (m->quadr) {
        for (int i = 0; i < NQ; i++) m->znQ[L][i] = A1 * m->znQ[QMAX][i];
        for (int j = 2; j <= m->q; j++)
            for (int i = 0; i < NQ; i++) m->znQ[j][i] = FMA(m->l[j], m->znQ[L][i], m->znQ[j][i]);
    }
    if (m->sensi)
        for (int is = 0; is < NQ; is++) {
            for (int i = 0; i < NS; i++) m->znS[L][is][i] = A1 * m->znS[QMAX][is][i];
            for (int j = 2; j <= m->q; j++)
                for (int i = 0; i < NS; i++)
                    m->znS[j][is][i] = FMA(m->l[j], m->znS[L][is][i], m->znS[j][is][i]);
        }
}

static void cv_decrease_bdf(cvmem *m)
{
    for (int i = 0; i <= QMAX; i++) m->l[i] = 0.0;
    m->l[2] = 1.0;
    double hsum = 0.0;
    for (int j = 1; j <= m->q - 2; j++) {
        hsum += m->tau[j];
        double xi = hsum / m->hscale;
        for (int i = j + 2; i >= 2; i--) m->l[i] = FMA(m->l[i], xi, m->l[i - 1]);
    }
    for (int j = 2; j < m->q; j++)
        for (int i = 0; i < NS; i++) m->zn[j][i] = FMA(-m->l[j], m->zn[m->q][i], m->zn[j][i]);
    if (m->quadr)
        for (int j = 2; j < m->q; j++)
            for (int i = 0; i < NQ; i++) m->znQ[j][i] = FMA(-m->l[j], m->znQ[m->q][i], m->znQ[j][i]);
    if (m->sensi)
        for (int is = 0; is < NQ; is++)
            for (int j = 2; j < m->q; j++)
                for (int i = 0; i < NS; i++)
                    m->znS[j][is][i] = FMA(-m->l[j], m->znS[m->q][is][i], m->znS[j][is][i]);
}

static void cv_adjust_order(cvmem *m, int deltaq)
{
    if ((m->q == 2) && (deltaq != 1)) return;
    if (deltaq == 1) cv_increase_bdf(m);
    else if (deltaq == -1) cv_decrease_bdf(m);
}

static void cv_adjust_params(cvmem *m)
{
    if (m->qprime != m->q) {
        cv_adjust_order(m, m->qprime - m->q);
        m->q = m->qprime;
        m->L = m->q + 1;
        m->qwait = m->L;
    }
    cv_rescale(m);
}

static void cv_predict(cvmem *m)
{
    m->tn += m->h;
    if (m->tstopset) {
        if ((m->tn - m->tstop) * m->h > 0.0) m->tn = m->tstop;
    }
    for (int k = 1; k <= m->q; k++)
        for (int j = m->q; j >= k; j--) {
            for (int i = 0; i < NS; i++) m->zn[j - 1][i] = m->zn[j - 1][i] + m->zn[j][i];
            if (m->quadr) for (int i = 0; i < NQ; i++) m->znQ[j - 1][i] = m->znQ[j - 1][i] + m->znQ[j][i];
            if (m->sensi)
                for (int is = 0; is < NQ; is++)
                    for (int i = 0; i < NS; i++) m->znS[j - 1][is][i] = m->znS[j - 1][is][i] + m->znS[j][is][i];
        }
}

static void cv_restore(cvmem *m, double saved_t)
{
    m->tn = saved_t;
    for (int k = 1; k <= m->q; k++)
        for (int j = m->q; j >= k; j--) {
            for (int i = 0; i < NS; i++) m->zn[j - 1][i] = m->zn[j - 1][i] - m->zn[j][i];
            if (m->quadr) for (int i = 0; i < NQ; i++) m->znQ[j - 1][i] = m->znQ[j - 1][i] - m->znQ[j][i];
            if (m->sensi)
                for (int is = 0; is < NQ; is++)
                    for (int i = 0; i < NS; i++) m->znS[j - 1][is][i] = m->znS[j - 1][is][i] - m->znS[j][is][i];
        }
}

static void cv_set_tq_bdf(cvmem *m, double hsum, double alpha0, double alpha0_hat,
                          double xi_inv, double xistar_inv)
{
    int q = m->q;
    double A1 = 1.0 - alpha0_hat + alpha0;
    double A2 = FMA((double)q, A1, 1.0);
    m->tq[2] = fabs(A1 / (alpha0 * A2));
    m->tq[5] = fabs(A2 * xistar_inv / (m->l[q] * xi_inv));
    if (m->qwait == 1) {
        if (q > 1) {
            double C = xistar_inv / m->l[q];
            double A3 = alpha0 + 1.0 / q;
            double A4 = alpha0_hat + xi_inv;
            double Cpinv = (1.0 - A4 + A3) / A3;
            m->tq[1] = fabs(C * Cpinv);
        } else m->tq[1] = 1.0;
        hsum += m->tau[q];
        xi_inv = m->h / hsum;
        double A5 = alpha0 - (1.0 / (q + 1));
        double A6 = alpha0_hat - xi_inv;
        double Cppinv = (1.0 - A6 + A5) / A2;
        m->tq[3] = fabs(Cppinv / (xi_inv * (q + 2) * A5));
    }
    m->tq[4] = m->tq[2] * 10.0;       /* 1/tq[4] of CVODES (= tq[2]/nlscoef): the test multiplies */
}

static void cv_set_bdf(cvmem *m)
{
    int q = m->q;
    double alpha0, alpha0_hat, xi_inv, xistar_inv, hsum;
    m->l[0] = m->l[1] = xi_inv = xistar_inv = 1.0;
    for (int i = 2; i <= q; i++) m->l[i] = 0.0;
    alpha0 = alpha0_hat = -1.0;
    hsum = m->h;
    if (q > 1) {
        for (int j = 2; j < q; j++) {
            hsum += m->tau[j - 1];
            xi_inv = m->h / hsum;
            alpha0 -= 1.0 / j;
            for (int i = j; i >= 1; i--) m->l[i] = FMA(m->l[i - 1], xi_inv, m->l[i]);
        }
        alpha0 -= 1.0 / q;
        xistar_inv = -m->l[1] - alpha0;
        hsum += m->tau[q - 1];
        xi_inv = m->h / hsum;
        alpha0_hat = -m->l[1] - xi_inv;
        for (int i = q; i >= 1; i--) m->l[i] = FMA(m->l[i - 1], xistar_inv, m->l[i]);
    }
    cv_set_tq_bdf(m, hsum, alpha0, alpha0_hat, xi_inv, xistar_inv);
}

static void cv_set(cvmem *m)
{
    cv_set_bdf(m);
    m->rl1 = 1.0 / m->l[1];
    m->gamma = m->h * m->rl1;
    if (m->nst == 0) m->gammap = m->gamma;
    m->gamrat = (m->nst > 0) ? m->gamma / m->gammap : 1.0;
}

/* ---- linear solver interface (cvLsSetup / cvLsSolve with SUNLinSol_Dense) ---- */
static int cv_lsetup(cvmem *m, int convfail, const double *ypred)
{
    double dgamma = fabs((m->gamma / m->gammap) - 1.0);
    int jbad = (m->nst == 0) || (m->nst > m->nstlj + MSBJ) ||
               ((convfail == CV_FAIL_BAD_J) && (dgamma < CVLS_DGMAX)) ||
               (convfail == CV_FAIL_OTHER);
    if (!jbad) {
        m->jcur = 0;
        for (int i = 0; i < NS * NS; i++) m->A[i] = m->savedJ[i];
    } else {
        m->nje++;
        m->nstlj = m->nst;
        m->jcur = 1;
        for (int i = 0; i < NS * NS; i++) m->A[i] = 0.0;
        int retval = cv_jac(m, m->tn, ypred, m->A);
        if (retval < 0) return -1;
        if (retval > 0) return 1;
        for (int i = 0; i < NS * NS; i++) m->savedJ[i] = m->A[i];
    }
    /* SUNMatScaleAddI(-gamma, A) */
    double c = -m->gamma;
    for (int j = 0; j < NS; j++) {
        for (int i = 0; i < NS; i++) {
            if (i == j) m->A[(size_t)j * NS + i] = FMA(c, m->A[(size_t)j * NS + i], 1.0);
            else m->A[(size_t)j * NS + i] *= c;
        }
    }
    int ier = dense_getrf(m->A, NS, m->piv, m->inv_piv);
    return ier > 0 ? 1 : 0;
}

static int cv_nls_lsetup(cvmem *m, int jbad, int *convfail)
{
    if (jbad) *convfail = CV_FAIL_BAD_J;
    int retval = cv_lsetup(m, *convfail, m->y);
    m->nsetups++;
    m->nls_jcur = m->jcur;
    m->gamrat = 1.0;
    m->gammap = m->gamma;
    m->crate = 1.0;
    m->crateS = 1.0;
    m->nstlp = m->nst;
    if (retval < 0) return CV_LSETUP_FAIL;
    if (retval > 0) return NLS_CONV_RECVR;
    return CV_SUCCESS;
}

static void cv_lsolve(cvmem *m, double *b)
{
    dense_getrs(m->A, NS, m->piv, m->inv_piv, b);
    if (m->gamrat != 1.0) {
        double s = 2.0 / (1.0 + m->gamrat);
        for (int i = 0; i < NS; i++) b[i] *= s;
    }
}

static int cv_nls_residual(cvmem *m, const double *ycor, double *res)
{
    for (int i = 0; i < NS; i++) m->y[i] = m->zn[0][i] + ycor[i];
    int retval = cv_f(m, m->tn, m->y, m->ftemp);
    if (retval < 0) return CV_RHSFUNC_FAIL;
    if (retval > 0) return RHSFUNC_RECVR;
    for (int i = 0; i < NS; i++) {
        res[i] = FMA(m->rl1, m->zn[1][i], ycor[i]);
        res[i] = FMA(-m->gamma, m->ftemp[i], res[i]);
    }
    return CV_SUCCESS;
}

static int cv_nls_conv_test(cvmem *m, int curiter, const double *delta, const double *ycor)
{
    double del = wrms(delta, m->ewt, NS);
    if (curiter > 0) m->crate = fmax(CRDOWN * m->crate, del / m->delp);
    double dcon = del * fmin(1.0, m->crate) * m->tq[4];
    if (dcon <= 1.0) {
        m->acnrm = (curiter == 0) ? del : wrms(ycor, m->ewt, NS);
        return CV_SUCCESS;
    }
    if ((curiter >= 1) && (del > RDIV * m->delp)) return NLS_CONV_RECVR;
    m->delp = del;
    return NLS_CONTINUE;
}

/* residual of the combined (state + sensitivities) system: cvNlsResidualSensSim */
static int cv_nls_residual_sens(cvmem *m, double resS[NQD][NSD])
{
    for (int is = 0; is < NQ; is++)
        for (int i = 0; i < NS; i++) m->yS[is][i] = m->znS[0][is][i] + m->acorS[is][i];
    int retval = cv_fS(m, m->tn, m->y, m->yS, m->ftempS);
    if (retval < 0) return CV_SRHSFUNC_FAIL;
    if (retval > 0) return SRHSFUNC_RECVR;
    for (int is = 0; is < NQ; is++)
        for (int i = 0; i < NS; i++) {
            resS[is][i] = FMA(m->rl1, m->znS[1][is][i], m->acorS[is][i]);
            resS[is][i] = FMA(-m->gamma, m->ftempS[is][i], resS[is][i]);
        }
    return CV_SUCCESS;
}

/* cvNls + SUNNonlinSolSolve_Newton; with ism = CV_SIMULTANEOUS the iteration runs on the wrapper
   vector (y, yS_1..yS_Ns): one linear solve per member with the same factorisation, norms = max over
   the members (N_VWrmsNorm_SensWrapper) */
static int cv_nls(cvmem *m, int nflag)
{
    const int sim = m->sensi && m->ism == 0;
    int convfail = ((nflag == FIRST_CALL) || (nflag == PREV_ERR_FAIL)) ? CV_NO_FAILURES : CV_FAIL_OTHER;
    int callSetup = (nflag == PREV_CONV_FAIL) || (nflag == PREV_ERR_FAIL) || (m->nst == 0) ||
                    (m->nst >= m->nstlp + MSBP) || (fabs(m->gamrat - 1.0) > DGMAX);
    for (int i = 0; i < NS; i++) m->acor[i] = 0.0;
    if (sim) for (int is = 0; is < NQ; is++) for (int i = 0; i < NS; i++) m->acorS[is][i] = 0.0;
    double delta[NSD];
    double deltaS[NQD][NSD];
    int jbad = 0, retval;
    for (;;) {
        retval = cv_nls_residual(m, m->acor, delta);
        if (retval != CV_SUCCESS) break;
        if (sim) {
            retval = cv_nls_residual_sens(m, deltaS);
            if (retval != CV_SUCCESS) break;
        }
        if (callSetup) {
            retval = cv_nls_lsetup(m, jbad, &convfail);
            if (retval != CV_SUCCESS) break;
        }
        int curiter = 0;
        for (;;) {
            m->nni++;
            for (int i = 0; i < NS; i++) delta[i] = -1.0 * delta[i];
            cv_lsolve(m, delta);
            for (int i = 0; i < NS; i++) m->acor[i] = m->acor[i] + delta[i];
            if (sim) {
                for (int is = 0; is < NQ; is++) {
                    for (int i = 0; i < NS; i++) deltaS[is][i] = -1.0 * deltaS[is][i];
                    cv_lsolve(m, deltaS[is]);
                    for (int i = 0; i < NS; i++) m->acorS[is][i] = m->acorS[is][i] + deltaS[is][i];
                }
                /* cvNlsConvTestSensSim */
                double del = sens_update_norm(wrms(delta, m->ewt, NS), deltaS, m->ewtS);
                if (curiter > 0) m->crate = fmax(CRDOWN * m->crate, del / m->delp);
                double dcon = del * fmin(1.0, m->crate) * m->tq[4];
                if (dcon <= 1.0) {
                    m->acnrm = (curiter == 0) ? del
                                              : sens_update_norm(wrms(m->acor, m->ewt, NS), m->acorS, m->ewtS);
                    retval = CV_SUCCESS;
                } else if ((curiter >= 1) && (del > RDIV * m->delp)) retval = NLS_CONV_RECVR;
                else { m->delp = del; retval = NLS_CONTINUE; }
            } else {
                retval = cv_nls_conv_test(m, curiter, delta, m->acor);
            }
            if (retval == CV_SUCCESS) { m->nls_jcur = 0; break; }
            if (retval != NLS_CONTINUE) break;
            curiter++;
            if (curiter >= NLS_MAXCOR) { retval = NLS_CONV_RECVR; break; }
            retval = cv_nls_residual(m, m->acor, delta);
            if (retval != CV_SUCCESS) break;
            if (sim) {
                retval = cv_nls_residual_sens(m, deltaS);
                if (retval != CV_SUCCESS) break;
            }
        }
        if (retval == CV_SUCCESS) break;
        if ((retval > 0) && !m->nls_jcur) {
            callSetup = 1;
            jbad = 1;
            for (int i = 0; i < NS; i++) m->acor[i] = 0.0;
            if (sim) for (int is = 0; is < NQ; is++) for (int i = 0; i < NS; i++) m->acorS[is][i] = 0.0;
            continue;
        }
        break;
    }
    if (retval != CV_SUCCESS) return retval;
    for (int i = 0; i < NS; i++) m->y[i] = m->zn[0][i] + m->acor[i];
    if (sim)
        for (int is = 0; is < NQ; is++)
            for (int i = 0; i < NS; i++) m->yS[is][i] = m->znS[0][is][i] + m->acorS[is][i];
    return CV_SUCCESS;
}

/* cvStgrNls: Newton iteration on the sensitivity systems alone (ism = CV_STAGGERED), state fixed at
   the converged y; own convergence-rate estimate; one retry with a fresh Jacobian */
static int cv_stgr_nls(cvmem *m)
{
    int callSetup = 0, jbad = 0, convfail = CV_FAIL_OTHER, retval;
    double deltaS[NQD][NSD];
    for (int is = 0; is < NQ; is++) for (int i = 0; i < NS; i++) m->acorS[is][i] = 0.0;
    for (;;) {
        retval = cv_nls_residual_sens(m, deltaS);
        if (retval != CV_SUCCESS) break;
        if (callSetup) {
            retval = cv_nls_lsetup(m, jbad, &convfail);
            m->nsetupsS++;
            if (retval != CV_SUCCESS) break;
        }
        int curiter = 0;
        for (;;) {
            m->nniS++;
            for (int is = 0; is < NQ; is++) {
                for (int i = 0; i < NS; i++) deltaS[is][i] = -1.0 * deltaS[is][i];
                cv_lsolve(m, deltaS[is]);
                for (int i = 0; i < NS; i++) m->acorS[is][i] = m->acorS[is][i] + deltaS[is][i];
            }
            /* cvNlsConvTestSensStg */
            double del = sens_update_norm(0.0, deltaS, m->ewtS);
            if (curiter > 0) m->crateS = fmax(CRDOWN * m->crateS, del / m->delpS);
            double dcon = del * fmin(1.0, m->crateS) * m->tq[4];
            if (dcon <= 1.0) {
                if (m->sensi) m->acnrmS = (curiter == 0) ? del : sens_update_norm(0.0, m->acorS, m->ewtS);
                retval = CV_SUCCESS;
                m->nls_jcur = 0;
                break;
            }
            if ((curiter >= 1) && (del > RDIV * m->delpS)) { retval = NLS_CONV_RECVR; break; }
            m->delpS = del;
            curiter++;
            if (curiter >= NLS_MAXCOR) { retval = NLS_CONV_RECVR; break; }
            retval = cv_nls_residual_sens(m, deltaS);
            if (retval != CV_SUCCESS) break;
        }
        if (retval == CV_SUCCESS) break;
        if ((retval > 0) && !m->nls_jcur) {
            callSetup = 1;
            jbad = 1;
            for (int is = 0; is < NQ; is++) for (int i = 0; i < NS; i++) m->acorS[is][i] = 0.0;
            continue;
        }
        break;
    }
    if (retval != CV_SUCCESS) return retval;
    for (int is = 0; is < NQ; is++)
        for (int i = 0; i < NS; i++) m->yS[is][i] = m->znS[0][is][i] + m->acorS[is][i];
    return CV_SUCCESS;
}

/* N_VConstrMask entry: 1 where the constraint c on x is violated */
static int constr_violated(double c, double x)
{
    if (c == 2.0) return !(x > 0.0);
    if (c == 1.0) return !(x >= 0.0);
    if (c == -1.0) return !(x <= 0.0);
    if (c == -2.0) return !(x < 0.0);
    return 0;
}

/* cvCheckConstraints: small violations are absorbed into the correction, large ones shrink the step */
static int cv_check_constraints(cvmem *m)
{
    double mm[NSD], v[NSD];
    int any = 0;
    for (int i = 0; i < NS; i++) { mm[i] = constr_violated(m->constraints[i], m->y[i]) ? 1.0 : 0.0; any |= (mm[i] != 0.0); }
    if (!any) return CV_SUCCESS;
    for (int i = 0; i < NS; i++) {
        double a = (fabs(m->constraints[i]) >= 1.5) ? 1.0 : 0.0;     /* strict constraints aim 0.1 tolerances inside */
        double tmp = (a * m->constraints[i]) / m->ewt[i];
        tmp = FMA(-0.1, tmp, m->y[i]);
        v[i] = tmp * mm[i];
    }
    double vnorm = wrms(v, m->ewt, NS);
    if (vnorm * m->tq[4] <= 1.0) {       /* CVODES: vnorm <= tq[4]; tq[4] is stored inverted here */
        for (int i = 0; i < NS; i++) m->acor[i] = m->acor[i] - v[i];
        return CV_SUCCESS;
    }
    double minq = 1e308;                  /* N_VMinQuotient(zn[0], mm * (zn[0] - y)) */
    for (int i = 0; i < NS; i++) {
        double d = mm[i] * (m->zn[0][i] - m->y[i]);
        if (d != 0.0) { double qv = m->zn[0][i] / d; if (qv < minq) minq = qv; }
    }
    m->eta = 0.9 * minq;
    m->eta = fmax(m->eta, 0.1);
    return CONSTR_RECVR;
}

static int cv_handle_nflag(cvmem *m, int *nflagPtr, double saved_t, int *ncfPtr, long *ncfnPtr)
{
    int nflag = *nflagPtr;
    if (nflag == CV_SUCCESS) return DO_ERROR_TEST;
    (*ncfnPtr)++;
    cv_restore(m, saved_t);
    if (nflag < 0) return nflag;
    (*ncfPtr)++;
    m->etamax = 1.0;
    if (*ncfPtr == MXNCF) {      /* hmin = 0 */
        if (nflag == NLS_CONV_RECVR) return CV_CONV_FAILURE;
        if (nflag == RHSFUNC_RECVR) return CV_REPTD_RHSFUNC_ERR;
        if (nflag == QRHSFUNC_RECVR) return CV_REPTD_QRHSFUNC_ERR;
        if (nflag == SRHSFUNC_RECVR) return CV_REPTD_SRHSFUNC_ERR;
        if (nflag == CONSTR_RECVR) return CV_CONSTR_FAIL;
    }
    if (nflag != CONSTR_RECVR) m->eta = ETACF;   /* max(ETACF, hmin/|h|), hmin = 0; CONSTR_RECVR: eta set by the check */
    *nflagPtr = PREV_CONV_FAIL;
    cv_rescale(m);
    return PREDICT_AGAIN;
}

static int cv_do_error_test(cvmem *m, int *nflagPtr, double saved_t, double acor_nrm,
                            int *nefPtr, long *netfPtr, double *dsmPtr)
{
    double dsm = acor_nrm * m->tq[2];
    *dsmPtr = dsm;
    if (dsm <= 1.0) return CV_SUCCESS;
    (*nefPtr)++;
    (*netfPtr)++;
    *nflagPtr = PREV_ERR_FAIL;
    cv_restore(m, saved_t);
    if (*nefPtr == MXNEF) return CV_ERR_FAILURE;
    m->etamax = 1.0;
    if (*nefPtr <= MXNEF1) {
        m->eta = 1.0 / (rpower_r(BIAS2 * dsm, 1.0 / m->L) + ADDON);
        m->eta = fmax(ETAMIN, m->eta);
        if (*nefPtr >= SMALL_NEF) m->eta = fmin(m->eta, ETAMXF);
        cv_rescale(m);
        return TRY_AGAIN;
    }
    if (m->q > 1) {
        m->eta = ETAMIN;
        cv_adjust_order(m, -1);
        m->L = m->q;
        m->q--;
        m->qwait = m->L;
        cv_rescale(m);
        return TRY_AGAIN;
    }
    m->eta = ETAMIN;
    m->h *= m->eta;
    m->next_h = m->h;
    m->hscale = m->h;
    m->qwait = LONG_WAIT;
    int retval = cv_f(m, m->tn, m->zn[0], m->tempv);
    if (retval < 0) return CV_RHSFUNC_FAIL;
    if (retval > 0) return CV_UNREC_RHSFUNC_ERR;
    for (int i = 0; i < NS; i++) m->zn[1][i] = m->h * m->tempv[i];
    if (m->sensi) {
        retval = cv_fS(m, m->tn, m->zn[0], m->znS[0], m->tempvS);
        if (retval < 0) return CV_SRHSFUNC_FAIL;
        if (retval > 0) return CV_UNREC_SRHSFUNC_ERR;
        for (int is = 0; is < NQ; is++)
            for (int i = 0; i < NS; i++) m->znS[1][is][i] = m->h * m->tempvS[is][i];
    }
    if (m->quadr) {
        retval = cv_fQ(m, m->tn, m->zn[0], m->tempvQ);
        if (retval < 0) return CV_QRHSFUNC_FAIL;
        if (retval > 0) return CV_QRHSFUNC_FAIL - 3; /* CV_UNREC_QRHSFUNC_ERR */
        for (int i = 0; i < NQ; i++) m->znQ[1][i] = m->h * m->tempvQ[i];
    }
    return TRY_AGAIN;
}

static int cv_quad_nls(cvmem *m)
{
    int retval = cv_fQ(m, m->tn, m->y, m->acorQ);
    if (retval < 0) return CV_QRHSFUNC_FAIL;
    if (retval > 0) return QRHSFUNC_RECVR;
    for (int i = 0; i < NQ; i++) {
        m->acorQ[i] = FMA(m->h, m->acorQ[i], -m->znQ[1][i]);
        m->acorQ[i] = m->rl1 * m->acorQ[i];
        m->yQ[i] = m->znQ[0][i] + m->acorQ[i];
    }
    return CV_SUCCESS;
}

static void cv_complete_step(cvmem *m)
{
    m->nst++;
    m->hu = m->h;
    m->qu = m->q;
    for (int i = m->q; i >= 2; i--) m->tau[i] = m->tau[i - 1];
    if ((m->q == 1) && (m->nst > 1)) m->tau[2] = m->tau[1];
    m->tau[1] = m->h;
    for (int j = 0; j <= m->q; j++)
        for (int i = 0; i < NS; i++) m->zn[j][i] = FMA(m->l[j], m->acor[i], m->zn[j][i]);
    if (m->quadr)
        for (int j = 0; j <= m->q; j++)
            for (int i = 0; i < NQ; i++) m->znQ[j][i] = FMA(m->l[j], m->acorQ[i], m->znQ[j][i]);
    if (m->sensi)
        for (int is = 0; is < NQ; is++)
            for (int j = 0; j <= m->q; j++)
                for (int i = 0; i < NS; i++)
                    m->znS[j][is][i] = FMA(m->l[j], m->acorS[is][i], m->znS[j][is][i]);
    m->qwait--;
    if ((m->qwait == 1) && (m->q != QMAX)) {
        if (m->sensi)
            for (int is = 0; is < NQ; is++) for (int i = 0; i < NS; i++) m->znS[QMAX][is][i] = m->acorS[is][i];
        for (int i = 0; i < NS; i++) m->zn[QMAX][i] = m->acor[i];
        if (m->quadr && m->errconQ) for (int i = 0; i < NQ; i++) m->znQ[QMAX][i] = m->acorQ[i];
        m->saved_tq5 = m->tq[5];
    }
}

static void cv_set_eta(cvmem *m)
{
    if (m->eta < THRESH) {
        m->eta = 1.0;
        m->hprime = m->h;
    } else {
        m->eta = fmin(m->eta, m->etamax);
        /* eta /= max(1, |h|*hmax_inv*eta) with hmax_inv = 0 */
        m->eta /= fmax(1.0, fabs(m->h) * 0.0 * m->eta);
        m->hprime = m->h * m->eta;
    }
}

static double cv_compute_etaqm1(cvmem *m)
{
    m->etaqm1 = 0.0;
    if (m->q > 1) {
        double ddn = wrms(m->zn[m->q], m->ewt, NS);
        if (m->quadr && m->errconQ) ddn = quad_update_norm(m, ddn, m->znQ[m->q], m->ewtQ);
        if (m->sensi) ddn = sens_update_norm(ddn, m->znS[m->q], m->ewtS);
        ddn = ddn * m->tq[1];
        m->etaqm1 = 1.0 / (rpower_r(BIAS1 * ddn, 1.0 / m->q) + ADDON);
    }
    return m->etaqm1;
}

static double cv_compute_etaqp1(cvmem *m)
{
    m->etaqp1 = 0.0;
    if (m->q != QMAX) {
        if (m->saved_tq5 == 0.0) return m->etaqp1;
        double cquot = (m->tq[5] / m->saved_tq5) * rpower_i(m->h / m->tau[2], m->L);
        for (int i = 0; i < NS; i++) m->tempv[i] = FMA(-cquot, m->zn[QMAX][i], m->acor[i]);
        double dup = wrms(m->tempv, m->ewt, NS);
        if (m->quadr && m->errconQ) {
            for (int i = 0; i < NQ; i++) m->tempvQ[i] = FMA(-cquot, m->znQ[QMAX][i], m->acorQ[i]);
            dup = quad_update_norm(m, dup, m->tempvQ, m->ewtQ);
        }
        if (m->sensi) {
            for (int is = 0; is < NQ; is++)
                for (int i = 0; i < NS; i++)
                    m->tempvS[is][i] = FMA(-cquot, m->znS[QMAX][is][i], m->acorS[is][i]);
            dup = sens_update_norm(dup, m->tempvS, m->ewtS);
        }
        dup = dup * m->tq[3];
        m->etaqp1 = 1.0 / (rpower_r(BIAS3 * dup, 1.0 / (m->L + 1)) + ADDON);
    }
    return m->etaqp1;
}

static void cv_choose_eta(cvmem *m)
{
    double etam = fmax(m->etaqm1, fmax(m->etaq, m->etaqp1));
    if (etam < THRESH) {
        m->eta = 1.0;
        m->qprime = m->q;
        return;
    }
    if (etam == m->etaq) {
        m->eta = m->etaq;
        m->qprime = m->q;
    } else if (etam == m->etaqm1) {
        m->eta = m->etaqm1;
        m->qprime = m->q - 1;
    } else {
        m->eta = m->etaqp1;
        m->qprime = m->q + 1;
        for (int i = 0; i < NS; i++) m->zn[QMAX][i] = m->acor[i];
        if (m->quadr && m->errconQ) for (int i = 0; i < NQ; i++) m->znQ[QMAX][i] = m->acorQ[i];
        if (m->sensi)
            for (int is = 0; is < NQ; is++) for (int i = 0; i < NS; i++) m->znS[QMAX][is][i] = m->acorS[is][i];
    }
}

static void cv_prepare_next_step(cvmem *m, double dsm)
{
    if (m->etamax == 1.0) {
        m->qwait = m->qwait > 2 ? m->qwait : 2;
        m->qprime = m->q;
        m->hprime = m->h;
        m->eta = 1.0;
        return;
    }
    m->etaq = 1.0 / (rpower_r(BIAS2 * dsm, 1.0 / m->L) + ADDON);
    if (m->qwait != 0) {
        m->eta = m->etaq;
        m->qprime = m->q;
        cv_set_eta(m);
        return;
    }
    m->qwait = 2;
    cv_compute_etaqm1(m);
    cv_compute_etaqp1(m);
    cv_choose_eta(m);
    cv_set_eta(m);
}

static int cv_step(cvmem *m)
{
    double saved_t = m->tn, dsm = 0.0, dsmQ = 0.0;
    int ncf = 0, nef = 0, nefQ = 0, ncfS = 0, nefS = 0;
    int nflag = FIRST_CALL, kflag, eflag;
    if ((m->nst > 0) && (m->hprime != m->h)) cv_adjust_params(m);
    for (;;) {
        cv_predict(m);
        cv_set(m);
        nflag = cv_nls(m, nflag);
        if (nflag == CV_SUCCESS && m->constraints_set) nflag = cv_check_constraints(m);
        kflag = cv_handle_nflag(m, &nflag, saved_t, &ncf, &m->ncfn);
        if (kflag == PREDICT_AGAIN) continue;
        if (kflag != DO_ERROR_TEST) return kflag;
        eflag = cv_do_error_test(m, &nflag, saved_t, m->acnrm, &nef, &m->netf, &dsm);
        if (eflag == TRY_AGAIN) continue;
        if (eflag != CV_SUCCESS) return eflag;
        if (m->sensi && m->ism == 1) {      /* CV_STAGGERED: sensitivities after the state passed */
            ncf = nef = 0;
            /* f at the converged y (cvStep: needed by the sensitivity right-hand side); a recoverable
               failure is treated as a convergence failure: predict again, exactly as CVODES does */
            int retval = cv_f(m, m->tn, m->y, m->ftemp);
            if (retval < 0) return CV_RHSFUNC_FAIL;
            if (retval > 0) { nflag = PREV_CONV_FAIL; continue; }
            nflag = cv_stgr_nls(m);
            kflag = cv_handle_nflag(m, &nflag, saved_t, &ncfS, &m->ncfnS);
            if (kflag == PREDICT_AGAIN) continue;
            if (kflag != DO_ERROR_TEST) return kflag;
            double dsmS = 0.0;
            m->acnrmS = sens_update_norm(0.0, m->acorS, m->ewtS);
            eflag = cv_do_error_test(m, &nflag, saved_t, m->acnrmS, &nefS, &m->netfS, &dsmS);
            if (eflag == TRY_AGAIN) continue;
            if (eflag != CV_SUCCESS) return eflag;
            if (dsmS > dsm) dsm = dsmS;
        }
        if (m->quadr) {
            ncf = nef = 0;
            nflag = cv_quad_nls(m);
            kflag = cv_handle_nflag(m, &nflag, saved_t, &ncf, &m->ncfn);
            if (kflag == PREDICT_AGAIN) continue;
            if (kflag != DO_ERROR_TEST) return kflag;
            if (m->errconQ) {
                double acnrmQ = wrms(m->acorQ, m->ewtQ, NQ);
                eflag = cv_do_error_test(m, &nflag, saved_t, acnrmQ, &nefQ, &m->netfQ, &dsmQ);
                if (eflag == TRY_AGAIN) continue;
                if (eflag != CV_SUCCESS) return eflag;
                if (dsmQ > dsm) dsm = dsmQ;
            }
        }
        break;
    }
    cv_complete_step(m);
    cv_prepare_next_step(m, dsm);
    m->etamax = (m->nst <= SMALL_NST) ? ETAMX2 : ETAMX3;
    for (int i = 0; i < NS; i++) m->acor[i] = m->tq[2] * m->acor[i];
    if (m->quadr) for (int i = 0; i < NQ; i++) m->acorQ[i] = m->tq[2] * m->acorQ[i];
    if (m->sensi)
        for (int is = 0; is < NQ; is++) for (int i = 0; i < NS; i++) m->acorS[is][i] = m->tq[2] * m->acorS[is][i];
    return CV_SUCCESS;
}

/* CVodeGetDky with k = 0 (5.x form: linear combination of s^j zn[j], j = q..0) */
static int cv_get_dky0(cvmem *m, double t, double *dky, double *dkyQ)
{
    double tfuzz = FUZZ_FACTOR * UROUND * (fabs(m->tn) + fabs(m->hu));
    if (m->hu < 0.0) tfuzz = -tfuzz;
    double tp = m->tn - m->hu - tfuzz;
    double tn1 = m->tn + tfuzz;
    if ((t - tp) * (t - tn1) > 0.0) return CV_BAD_T;
    double s = (t - m->tn) / m->h;
    double cvals[QMAX + 1];
    int nvec = 0;
    for (int j = m->q; j >= 0; j--) {
        double c = 1.0;
        for (int i = 0; i < j; i++) c *= s;
        cvals[nvec++] = c;
    }
    for (int i = 0; i < NS; i++) {
        double acc = cvals[0] * m->zn[m->q][i];
        for (int v = 1; v < nvec; v++) acc = FMA(cvals[v], m->zn[m->q - v][i], acc);
        dky[i] = acc;
    }
    if (dkyQ) {
        for (int i = 0; i < NQ; i++) {
            double acc = cvals[0] * m->znQ[m->q][i];
            for (int v = 1; v < nvec; v++) acc = FMA(cvals[v], m->znQ[m->q - v][i], acc);
            dkyQ[i] = acc;
        }
    }
    return CV_SUCCESS;
}

/* CVodeGetSensDky with k = 0, all parameters */
static int cv_get_sens_dky0(cvmem *m, double t, double *dkyS /* [NQ][NS] */)
{
    double tfuzz = FUZZ_FACTOR * UROUND * (fabs(m->tn) + fabs(m->hu));
    if (m->hu < 0.0) tfuzz = -tfuzz;
    double tp = m->tn - m->hu - tfuzz;
    double tn1 = m->tn + tfuzz;
    if ((t - tp) * (t - tn1) > 0.0) return CV_BAD_T;
    double s = (t - m->tn) / m->h;
    double cvals[QMAX + 1];
    int nvec = 0;
    for (int j = m->q; j >= 0; j--) {
        double c = 1.0;
        for (int i = 0; i < j; i++) c *= s;
        cvals[nvec++] = c;
    }
    for (int is = 0; is < NQ; is++)
        for (int i = 0; i < NS; i++) {
            double acc = cvals[0] * m->znS[m->q][is][i];
            for (int v = 1; v < nvec; v++) acc = FMA(cvals[v], m->znS[m->q - v][is][i], acc);
            dkyS[is * NS + i] = acc;
        }
    return CV_SUCCESS;
}

/* First-call block of CVode(): f(t0,y0), h0, scale zn[1]. */
static int cv_first_call(cvmem *m, double tout)
{
    if (m->constraints_set) {           /* cvInitialSetup: y0 must satisfy the constraints */
        if (m->sensi && m->ism == 0) return CV_ILL_INPUT;
        for (int i = 0; i < NS; i++) if (constr_violated(m->constraints[i], m->zn[0][i])) return CV_ILL_INPUT;
    }
    if (ewt_set(m, m->zn[0], m->ewt) != 0) return CV_ILL_INPUT;
    if (m->quadr && m->errconQ) if (ewtQ_set(m, m->znQ[0], m->ewtQ) != 0) return CV_ILL_INPUT;
    if (m->sensi) if (sens_ewt_set(m, m->znS[0], m->ewtS) != 0) return CV_ILL_INPUT;
    int retval = cv_f(m, m->tn, m->zn[0], m->zn[1]);
    if (retval < 0) return CV_RHSFUNC_FAIL;
    if (retval > 0) return CV_FIRST_RHSFUNC_ERR;
    if (m->sensi) {
        retval = cv_fS(m, m->tn, m->zn[0], m->znS[0], m->znS[1]);
        if (retval < 0) return CV_SRHSFUNC_FAIL;
        if (retval > 0) return CV_FIRST_SRHSFUNC_ERR;
    }
    if (m->quadr) {
        retval = cv_fQ(m, m->tn, m->zn[0], m->znQ[1]);
        if (retval < 0) return CV_QRHSFUNC_FAIL;
        if (retval > 0) return CV_FIRST_QRHSFUNC_ERR;
    }
    if (m->tstopset) {
        if ((m->tstop - m->tn) * (tout - m->tn) <= 0.0) return CV_ILL_INPUT;
    }
    double tout_hin = tout;
    if (m->tstopset && (tout - m->tn) * (tout - m->tstop) > 0.0) tout_hin = m->tstop;
    int hflag = cv_hin(m, tout_hin);
    if (hflag != CV_SUCCESS) return hflag;
    if (m->tstopset) {
        if ((m->tn + m->h - m->tstop) * m->h > 0.0)
            m->h = (m->tstop - m->tn) * (1.0 - 4.0 * UROUND);
    }
    m->hscale = m->h;
    m->hprime = m->h;
    for (int i = 0; i < NS; i++) m->zn[1][i] = m->h * m->zn[1][i];
    if (m->quadr) for (int i = 0; i < NQ; i++) m->znQ[1][i] = m->h * m->znQ[1][i];
    if (m->sensi)
        for (int is = 0; is < NQ; is++) for (int i = 0; i < NS; i++) m->znS[1][is][i] = m->h * m->znS[1][is][i];
    return CV_SUCCESS;
}

/* Pre-step block of the CVode() internal loop: ewt refresh + too-much-accuracy test. */
static int cv_pre_step(cvmem *m)
{
    if (m->nst > 0) {
        if (ewt_set(m, m->zn[0], m->ewt) != 0) return CV_ILL_INPUT;
        if (m->quadr && m->errconQ) if (ewtQ_set(m, m->znQ[0], m->ewtQ) != 0) return CV_ILL_INPUT;
        if (m->sensi) if (sens_ewt_set(m, m->znS[0], m->ewtS) != 0) return CV_ILL_INPUT;
    }
    double nrm = wrms(m->zn[0], m->ewt, NS);
    if (m->quadr && m->errconQ) nrm = quad_update_norm(m, nrm, m->znQ[0], m->ewtQ);
    if (m->sensi) nrm = sens_update_norm(nrm, m->znS[0], m->ewtS);
    m->tolsf = UROUND * nrm;
    if (m->tolsf > 1.0) { m->tolsf *= 2.0; return CV_TOO_MUCH_ACC; }
    m->tolsf = 1.0;
    return CV_SUCCESS;
}

/* CVode(cv_mem, tout, yout, &tret, CV_NORMAL) incl. tstop logic. */
static int cv_cvode_normal(cvmem *m, double tout, double *yout, double *youtQ, double *tret)
{
    if (m->nst == 0) {
        m->tretlast = *tret = m->tn;
        int ier = cv_first_call(m, tout);
        if (ier != CV_SUCCESS) return ier;
    } else {
        double troundoff = FUZZ_FACTOR * UROUND * (fabs(m->tn) + fabs(m->h));
        if (m->tstopset) {
            if ((m->tn - m->tstop) * m->h > 0.0) return CV_ILL_INPUT;
        }
        if ((m->tn - tout) * m->h >= 0.0) {
            m->tretlast = *tret = tout;
            if (cv_get_dky0(m, tout, yout, youtQ) != CV_SUCCESS) return CV_ILL_INPUT;
            return CV_SUCCESS;
        }
        if (m->tstopset) {
            if (fabs(m->tn - m->tstop) <= troundoff) {
                if (cv_get_dky0(m, m->tstop, yout, youtQ) != CV_SUCCESS) return CV_ILL_INPUT;
                m->tretlast = *tret = m->tstop;
                m->tstopset = 0;
                return CV_TSTOP_RETURN;
            }
            if ((m->tn + m->hprime - m->tstop) * m->h > 0.0) {
                m->hprime = (m->tstop - m->tn) * (1.0 - 4.0 * UROUND);
                m->eta = m->hprime / m->h;
            }
        }
    }
    long nstloc = 0;
    for (;;) {
        m->next_h = m->h;
        m->next_q = m->q;
        int ier = cv_pre_step(m);
        if (ier == CV_ILL_INPUT) { m->tretlast = *tret = m->tn; return ier; }
        if ((m->mxstep > 0) && (nstloc >= m->mxstep)) {
            m->tretlast = *tret = m->tn;
            for (int i = 0; i < NS; i++) yout[i] = m->zn[0][i];
            return CV_TOO_MUCH_WORK;
        }
        if (ier != CV_SUCCESS) { m->tretlast = *tret = m->tn; return ier; }
        int kflag = cv_step(m);
        if (kflag != CV_SUCCESS) {
            m->tretlast = *tret = m->tn;
            for (int i = 0; i < NS; i++) yout[i] = m->zn[0][i];
            return kflag;
        }
        nstloc++;
        if (m->tstopset) {
            double troundoff = FUZZ_FACTOR * UROUND * (fabs(m->tn) + fabs(m->h));
            if (fabs(m->tn - m->tstop) <= troundoff) m->tn = m->tstop;
        }
        if ((m->tn - tout) * m->h >= 0.0) {
            m->tretlast = *tret = tout;
            cv_get_dky0(m, tout, yout, youtQ);
            m->next_q = m->qprime;
            m->next_h = m->hprime;
            return CV_SUCCESS;
        }
        if (m->tstopset) {
            double troundoff = FUZZ_FACTOR * UROUND * (fabs(m->tn) + fabs(m->h));
            if (fabs(m->tn - m->tstop) <= troundoff) {
                cv_get_dky0(m, m->tstop, yout, youtQ);
                m->tretlast = *tret = m->tstop;
                m->tstopset = 0;
                return CV_TSTOP_RETURN;
            }
            if ((m->tn + m->hprime - m->tstop) * m->h > 0.0) {
                m->hprime = (m->tstop - m->tn) * (1.0 - 4.0 * UROUND);
                m->eta = m->hprime / m->h;
            }
        }
    }
}

/* One CVode(..., CV_ONE_STEP) call as issued by CVodeF (no tstop, mxstep never hit). */
static int cv_cvode_one_step(cvmem *m, double tout)
{
    if (m->nst == 0) {
        int ier = cv_first_call(m, tout);
        if (ier != CV_SUCCESS) return ier;
    }
    m->next_h = m->h;
    m->next_q = m->q;
    int ier = cv_pre_step(m);
    if (ier != CV_SUCCESS) return ier;
    return cv_step(m);
}

/* ------------------------------------------------------------------------- */
/* public configuration / drivers                                              */
/* ------------------------------------------------------------------------- */
typedef struct {
    double rtol;
    double atol[NSD];
    double rtolB, atolB, rtolQB, atolQB;
    int mxstep;             /* CVODES default 500 */
    int max_retries_fwd;    /* sunode: 5 */
    int max_retries_bwd;    /* sunode: 50 */
    int max_traj_points;    /* 0 = unbounded; mirrors the device arena capacity */
    int constraints_set, hermite;   /* hermite: CVodeAdjInit(..., CV_HERMITE) instead of CV_POLYNOMIAL */
    double constraints[NSD];    /* CVodeSetConstraints: 0 none, +-1 (>= / <= 0), +-2 (> / < 0); forward problem only */
    int no_errconQB;            /* 1: CVodeSetQuadErrConB(false) -- the quadratures ride along without a say in the
                                   step / order control (sunode sets true, solver.py:615; the DVODE pin of the backward
                                   controller, tools/make_golden_dvode_backward.py, needs the pure adjoint system) */
    int reserved_pad;
} orc_config;

typedef struct {
    int B;
    traj_t *traj;           /* [B] */
} orc_batch;

static void fill_nan(double *p, size_t n)
{
    for (size_t i = 0; i < n; i++) p[i] = NAN;
}

static void export_stats(const cvmem *m, const traj_t *tr, int64_t *st)
{
    st[ST_NST] += m->nst; st[ST_NFE] += m->nfe; st[ST_NSETUPS] += m->nsetups; st[ST_NJE] += m->nje;
    st[ST_NNI] += m->nni; st[ST_NCFN] += m->ncfn; st[ST_NETF] += m->netf; st[ST_QLAST] = m->qu;
    st[ST_NFQE] += m->nfQe; st[ST_NETFQ] += m->netfQ;
    if (tr) { st[ST_NPTS] = tr->np; st[ST_NINTERP] = tr->n_interp; st[ST_NREBUILD] = tr->n_rebuild; }
}

/* sunode Solver.solve (solver.py:467-527): CVodeReInit + CVode(NORMAL) per tval, <=max_retries */
static int solve_plain_one(const orc_config *cfg, const double *y0, const double *ps, const double *pr,
                           double t0, const double *tvals, int n_t, double *y_out, int64_t *st)
{
    cvmem *m = (cvmem *)calloc(1, sizeof(cvmem));
    m->ps = ps; m->pr = pr; m->backward = 0; m->tr = NULL;
    m->rtol = cfg->rtol; for (int i = 0; i < NS; i++) m->atol[i] = cfg->atol[i];
    m->quadr = 0; m->errconQ = 0; m->mxstep = cfg->mxstep; m->tstopset = 0;
    m->constraints_set = cfg->constraints_set; for (int i = 0; i < NS; i++) m->constraints[i] = cfg->constraints[i];
    cv_reinit(m, t0, y0, NULL);
    int status = CV_SUCCESS;
    double ybuf[NSD], tret;
    for (int k = 0; k < n_t && status == CV_SUCCESS; k++) {
        double t = tvals[k];
        /* solver.py:505 writes row 0 here (it assumes tvals[0] == t0); row k is the sane reading of a
           repeated t0 and identical in the only case the reference supports */
        if (t == t0) { for (int i = 0; i < NS; i++) y_out[(size_t)k * NS + i] = y0[i]; continue; }
        int retval = CV_TOO_MUCH_WORK, retry;
        for (retry = 0; retry < cfg->max_retries_fwd; retry++) {
            retval = cv_cvode_normal(m, t, ybuf, NULL, &tret);
            if (retval == CV_SUCCESS) break;
            if (retval != CV_TOO_MUCH_WORK) break;
            st[ST_RETRIES]++;
        }
        if (retval != CV_SUCCESS) { status = retval; break; }
        for (int i = 0; i < NS; i++) y_out[(size_t)k * NS + i] = ybuf[i];
    }
    export_stats(m, NULL, st);
    free(m);
    return status;
}

/* sunode Solver.solve with sens_mode (solver.py:467-527, 360-392): CVodeReInit + CVodeSensReInit, then
   CVode(NORMAL) + CVodeGetSens per tval */
static int solve_sens_one(const orc_config *cfg, int ism, const double *pbar, const double *y0, const double *ps,
                          const double *pr, const double *sens0, double t0, const double *tvals, int n_t,
                          double *y_out, double *sens_out, int64_t *st)
{
    cvmem *m = (cvmem *)calloc(1, sizeof(cvmem));
    m->ps = ps; m->pr = pr; m->backward = 0; m->tr = NULL;
    m->rtol = cfg->rtol; for (int i = 0; i < NS; i++) m->atol[i] = cfg->atol[i];
    m->quadr = 0; m->errconQ = 0; m->mxstep = cfg->mxstep; m->tstopset = 0;
    m->sensi = 1; m->ism = ism;
    m->constraints_set = cfg->constraints_set; for (int i = 0; i < NS; i++) m->constraints[i] = cfg->constraints[i];
    for (int is = 0; is < NQ; is++) m->pbar[is] = pbar ? fabs(pbar[is]) : 1.0;
    cv_reinit(m, t0, y0, NULL);
    cv_sens_reinit(m, sens0);
    int status = CV_SUCCESS;
    double ybuf[NSD], tret;
    for (int k = 0; k < n_t && status == CV_SUCCESS; k++) {
        double t = tvals[k];
        if (t == t0) {
            for (int i = 0; i < NS; i++) y_out[(size_t)k * NS + i] = y0[i];
            for (int j = 0; j < NQ * NS; j++) sens_out[(size_t)k * NQ * NS + j] = sens0[j];
            continue;
        }
        int retval = CV_TOO_MUCH_WORK, retry;
        for (retry = 0; retry < cfg->max_retries_fwd; retry++) {
            retval = cv_cvode_normal(m, t, ybuf, NULL, &tret);
            if (retval == CV_SUCCESS) break;
            if (retval != CV_TOO_MUCH_WORK) break;
            st[ST_RETRIES]++;
        }
        if (retval != CV_SUCCESS) { status = retval; break; }
        for (int i = 0; i < NS; i++) y_out[(size_t)k * NS + i] = ybuf[i];
        if (cv_get_sens_dky0(m, t, sens_out + (size_t)k * NQ * NS) != CV_SUCCESS) { status = CV_BAD_T; break; }
    }
    export_stats(m, NULL, st);
    /* sensitivity counters ride in the quadrature / interpolation slots of the adjoint path */
    st[ST_NFQE] = m->nfSe; st[ST_NETFQ] = m->netfS; st[ST_NINTERP] = m->nniS; st[ST_NREBUILD] = m->ncfnS;
    free(m);
    return status;
}

/* sunode AdjointSolver.solve_forward (solver.py:682-721): CVodeReInit + CVodeAdjReInit + CVodeF per tval */
static int solve_forward_one(const orc_config *cfg, const double *y0, const double *ps, const double *pr,
                             double t0, const double *tvals, int n_t, double *y_out, traj_t *tr, int64_t *st)
{
    cvmem *m = (cvmem *)calloc(1, sizeof(cvmem));
    m->ps = ps; m->pr = pr; m->backward = 0; m->tr = tr;
    m->rtol = cfg->rtol; for (int i = 0; i < NS; i++) m->atol[i] = cfg->atol[i];
    m->quadr = 0; m->errconQ = 0; m->mxstep = cfg->mxstep; m->tstopset = 0;
    m->constraints_set = cfg->constraints_set; for (int i = 0; i < NS; i++) m->constraints[i] = cfg->constraints[i];
    cv_reinit(m, t0, y0, NULL);
    tr->np = 0; tr->n_interp = tr->n_rebuild = 0; tr->hermite = cfg->hermite;
    int status = CV_SUCCESS, first = 1;
    for (int k = 0; k < n_t && status == CV_SUCCESS; k++) {
        double tout = tvals[k];
        if (tout == t0) { for (int i = 0; i < NS; i++) y_out[(size_t)k * NS + i] = y0[i]; continue; }   /* solver.py:707, row k */
        if (first) {
            tr->tinitial = m->tn;
            tr->hermite = cfg->hermite;
            if (cfg->hermite) {          /* CVAhermiteStorePnt at nst == 0: y' = f(t0, y0), evaluated outside the counters */
                double yd0[NSD];
                long keep = m->nfe;
                if (cv_f(m, m->tn, m->zn[0], yd0) != 0) { status = CV_RHSFUNC_FAIL; m->nfe = keep; break; }
                m->nfe = keep;
                traj_push(tr, m->tn, m->zn[0], yd0, 0, 0);
            } else
            traj_push(tr, m->tn, m->zn[0], NULL, 0, 0);
            tr->tfinal = m->tn;
            first = 0;
        } else if ((m->tn - tout) * m->h >= 0.0) {
            if (cv_get_dky0(m, tout, y_out + (size_t)k * NS, NULL) != CV_SUCCESS) status = CV_BAD_T;
            continue;
        }
        for (;;) {
            int flag = cv_cvode_one_step(m, tout);
            if (flag < 0) { status = flag; break; }
            double ydp[NSD];            /* CVAhermiteStorePnt: y' = zn[1] / h */
            for (int i = 0; i < NS; i++) ydp[i] = (1.0 / m->h) * m->zn[1][i];
            if (traj_push(tr, m->tn, m->zn[0], cfg->hermite ? ydp : NULL, m->qu, cfg->max_traj_points) != 0) { status = CV_TOO_MUCH_WORK; break; }
            tr->tfinal = m->tn;
            if ((m->tn - tout) * m->h >= 0.0) {
                cv_get_dky0(m, tout, y_out + (size_t)k * NS, NULL);
                m->tretlast = tout;
                break;
            }
        }
    }
    tr->newdata = 1;
    export_stats(m, tr, st);
    free(m);
    return status;
}

/* sunode AdjointSolver.solve_backward (solver.py:723-784) on top of CVodeB semantics. */
static int solve_backward_one(const orc_config *cfg, const double *ps, const double *pr,
                              double t0, double tend, const double *tvals, int n_t,
                              const double *grads, double *grad_out, double *lamda_out,
                              traj_t *tr, int64_t *st, double *lamda_all, double *quad_all)
{
    cvmem *m = (cvmem *)calloc(1, sizeof(cvmem));
    m->ps = ps; m->pr = pr; m->backward = 1; m->tr = tr;
    m->rtol = cfg->rtolB; for (int i = 0; i < NS; i++) m->atol[i] = cfg->atolB;
    m->quadr = 1; m->errconQ = cfg->no_errconQB ? 0 : 1; m->rtolQ = cfg->rtolQB; m->atolQ = cfg->atolQB;
    m->mxstep = cfg->mxstep; m->tstopset = 0;
    double lam[NSD], quad[NQD], quad_out[NQD], tret;
    for (int i = 0; i < NS; i++) lam[i] = 0.0;
    for (int i = 0; i < NQ; i++) { quad[i] = 0.0; quad_out[i] = 0.0; }
    int status = CV_SUCCESS;
    int first_call = 1;
    /* ts = [t0] + reversed(tvals) + [tend]; interval i = (ts[i+1], ts[i]); grads reversed, then None */
    for (int iv = 0; iv <= n_t && status == CV_SUCCESS; iv++) {
        double t_upper = (iv == 0) ? t0 : tvals[n_t - iv];
        double t_lower = (iv == n_t) ? tend : tvals[n_t - 1 - iv];
        if (t_lower < t_upper) {
            cv_reinit(m, t_upper, lam, quad);          /* CVodeReInitB + CVodeQuadReInitB */
            /* CVodeB: legality checks */
            if (first_call) {
                if ((t_upper - tr->tinitial) < 0.0 || (tr->tfinal - t_upper) < 0.0) { status = CV_BAD_TB0; break; }
                first_call = 0;
            }
            if ((t_lower - tr->tinitial) < 0.0 || (tr->tfinal - t_lower) < 0.0) {
                double tfuzz = 100.0 * UROUND * (fabs(tr->tinitial) + fabs(tr->tfinal));
                if ((t_lower - tr->tinitial) < -tfuzz || (tr->tfinal - t_lower) < -tfuzz) { status = CV_ILL_INPUT; break; }
            }
            int retval = CV_TOO_MUCH_WORK, retry;
            for (retry = 0; retry < cfg->max_retries_bwd; retry++) {
                m->tstopset = 1; m->tstop = tr->tinitial;        /* CVodeSetStopTime(cvB, t0_) */
                retval = cv_cvode_normal(m, t_lower, lam, quad_out, &tret);
                /* sunode accepts only retval == 0 (solver.py:761-766).  CV_TSTOP_RETURN cannot
                   precede reaching tBout because tstop = tinitial <= t_lower and the tout test
                   comes first inside CVode. */
                if (retval == CV_SUCCESS) break;
                if (retval != CV_TOO_MUCH_WORK) break;
                st[ST_RETRIES]++;
            }
            export_stats(m, tr, st);
            if (retval != CV_SUCCESS) { status = retval; break; }
            for (int i = 0; i < NQ; i++) quad[i] = quad_out[i];
        }
        if (iv < n_t) {
            const double *g = grads + (size_t)(n_t - 1 - iv) * NS;
            for (int i = 0; i < NS; i++) lam[i] -= g[i];
            /* solver.py:778-781: lamda_all_out[-i] / quad_all_out[-i] (row 0 for the first jump) */
            const size_t row = (iv == 0) ? 0 : (size_t)(n_t - iv);
            if (lamda_all) for (int i = 0; i < NS; i++) lamda_all[row * NS + i] = lam[i];
            if (quad_all) for (int i = 0; i < NQ; i++) quad_all[row * NQ + i] = quad[i];
        }
    }
    for (int i = 0; i < NQ; i++) grad_out[i] = quad_out[i];
    for (int i = 0; i < NS; i++) lamda_out[i] = lam[i];
    free(m);
    return status;
}

/* ---- exported C entry points (ctypes) ---- */
int orc_sizes(int *n_states, int *n_sub, int *n_rem, int *n_stats)
{
    *n_states = NS; *n_sub = NQ; *n_rem = NR; *n_stats = ST_COUNT;
    return 0;
}

int orc_config_size(void) { return (int)sizeof(orc_config); }

/* evaluate the five generated callbacks at one point (golden-vector checks) */
int orc_eval(double t, const double *y, const double *lam, const double *ps, const double *pr,
             double *rhs, double *jac, double *adj, double *quad, double *adjjac, int *codes)
{
    codes[0] = sa_rhs(t, y, ps, pr, rhs);
    codes[1] = sa_jac(t, y, ps, pr, jac);
    codes[2] = sa_adj_rhs(t, y, lam, ps, pr, adj);
    codes[3] = sa_quad_rhs(t, y, lam, ps, pr, quad);
    codes[4] = sa_adj_jac(t, y, ps, pr, adjjac);
    return 0;
}

orc_batch *orc_batch_new(int B)
{
    orc_batch *b = (orc_batch *)calloc(1, sizeof(orc_batch));
    b->B = B;
    b->traj = (traj_t *)calloc((size_t)B, sizeof(traj_t));
    return b;
}

void orc_batch_free(orc_batch *b)
{
    if (!b) return;
    for (int i = 0; i < b->B; i++) { free(b->traj[i].t); free(b->traj[i].y); free(b->traj[i].yd); free(b->traj[i].order); }
    free(b->traj);
    free(b);
}

int orc_traj_len(const orc_batch *b, int i) { return b->traj[i].np; }

int orc_traj_get(const orc_batch *b, int i, double *t, double *y, int *order)
{
    const traj_t *tr = &b->traj[i];
    for (int k = 0; k < tr->np; k++) {
        t[k] = tr->t[k]; order[k] = tr->order[k];
        for (int j = 0; j < NS; j++) y[(size_t)k * NS + j] = tr->y[(size_t)k * NSD + j];
    }
    return tr->np;
}

/* rem_stride = 0: pr shared by the whole batch; = NR: per-instance. */
int orc_solve_batch(const orc_config *cfg, int B, const double *y0, const double *ps, const double *pr,
                    int rem_stride, double t0, const double *tvals, int n_t,
                    double *y_out, int32_t *status, int64_t *stats, int nthreads)
{
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 0 ? nthreads : 1)
    for (int b = 0; b < B; b++) {
        int64_t *st = stats + (size_t)b * ST_COUNT;
        memset(st, 0, sizeof(int64_t) * ST_COUNT);
        double *yo = y_out + (size_t)b * n_t * NS;
        status[b] = solve_plain_one(cfg, y0 + (size_t)b * NS, ps + (size_t)b * NQ,
                                    pr + (size_t)b * rem_stride, t0, tvals, n_t, yo, st);
        if (status[b] != CV_SUCCESS) fill_nan(yo, (size_t)n_t * NS);
    }
    return 0;
}

int orc_solve_sens_batch(const orc_config *cfg, int ism, const double *pbar, int B, const double *y0,
                         const double *ps, const double *pr, int rem_stride, const double *sens0, double t0,
                         const double *tvals, int n_t, double *y_out, double *sens_out, int32_t *status,
                         int64_t *stats, int nthreads)
{
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 0 ? nthreads : 1)
    for (int b = 0; b < B; b++) {
        int64_t *st = stats + (size_t)b * ST_COUNT;
        memset(st, 0, sizeof(int64_t) * ST_COUNT);
        double *yo = y_out + (size_t)b * n_t * NS;
        double *so = sens_out + (size_t)b * n_t * NQ * NS;
        status[b] = solve_sens_one(cfg, ism, pbar, y0 + (size_t)b * NS, ps + (size_t)b * NQ,
                                   pr + (size_t)b * rem_stride, sens0 + (size_t)b * NQ * NS, t0, tvals, n_t,
                                   yo, so, st);
        if (status[b] != CV_SUCCESS) { fill_nan(yo, (size_t)n_t * NS); fill_nan(so, (size_t)n_t * NQ * NS); }
    }
    return 0;
}

int orc_solve_forward_batch(orc_batch *bt, const orc_config *cfg, int B, const double *y0, const double *ps,
                            const double *pr, int rem_stride, double t0, const double *tvals, int n_t,
                            double *y_out, int32_t *status, int64_t *stats, int nthreads)
{
    if (B > bt->B) return -1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 0 ? nthreads : 1)
    for (int b = 0; b < B; b++) {
        int64_t *st = stats + (size_t)b * ST_COUNT;
        memset(st, 0, sizeof(int64_t) * ST_COUNT);
        double *yo = y_out + (size_t)b * n_t * NS;
        status[b] = solve_forward_one(cfg, y0 + (size_t)b * NS, ps + (size_t)b * NQ,
                                      pr + (size_t)b * rem_stride, t0, tvals, n_t, yo, &bt->traj[b], st);
        if (status[b] != CV_SUCCESS) {
            fill_nan(yo, (size_t)n_t * NS);
            bt->traj[b].np = 0;         /* like the device arena: a failed forward pass leaves nothing to run backward on */
        }
    }
    return 0;
}

/* grads_stride = 0: one [n_t, n] cotangent block shared by the batch; = n_t*n: per instance. */
int orc_solve_backward_batch(orc_batch *bt, const orc_config *cfg, int B, const double *ps, const double *pr,
                             int rem_stride, double t0, double tend, const double *tvals, int n_t,
                             const double *grads, long grads_stride, double *grad_out, double *lamda_out,
                             int32_t *status, int64_t *stats, int nthreads, double *lamda_all, double *quad_all)
{
    if (B > bt->B) return -1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 0 ? nthreads : 1)
    for (int b = 0; b < B; b++) {
        int64_t *st = stats + (size_t)b * ST_COUNT;
        memset(st, 0, sizeof(int64_t) * ST_COUNT);
        traj_t *tr = &bt->traj[b];
        if (tr->np < 2) {          /* forward pass failed or never ran: CV_NO_FWD-like */
            status[b] = -102;
            fill_nan(grad_out + (size_t)b * NQ, NQ);
            fill_nan(lamda_out + (size_t)b * NS, NS);
            if (lamda_all) fill_nan(lamda_all + (size_t)b * n_t * NS, (size_t)n_t * NS);
            if (quad_all) fill_nan(quad_all + (size_t)b * n_t * NQ, (size_t)n_t * NQ);
            continue;
        }
        tr->newdata = 1;
        tr->have_last = 0;
        tr->n_interp = tr->n_rebuild = 0;
        status[b] = solve_backward_one(cfg, ps + (size_t)b * NQ, pr + (size_t)b * rem_stride, t0, tend,
                                       tvals, n_t, grads + (size_t)b * grads_stride,
                                       grad_out + (size_t)b * NQ, lamda_out + (size_t)b * NS, tr, st,
                                       lamda_all ? lamda_all + (size_t)b * n_t * NS : NULL,
                                       quad_all ? quad_all + (size_t)b * n_t * NQ : NULL);
        if (status[b] != CV_SUCCESS) {
            if (lamda_all) fill_nan(lamda_all + (size_t)b * n_t * NS, (size_t)n_t * NS);
            if (quad_all) fill_nan(quad_all + (size_t)b * n_t * NQ, (size_t)n_t * NQ);
        }
        if (status[b] != CV_SUCCESS) {
            fill_nan(grad_out + (size_t)b * NQ, NQ);
            fill_nan(lamda_out + (size_t)b * NS, NS);
        }
    }
    return 0;
}
