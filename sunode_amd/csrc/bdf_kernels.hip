/*
 * bdf_kernels.hip -- batched variable-order BDF(1-5)/Newton integrator + adjoint for gfx950.
 *
 * One integrator per parameter draw, one draw per lane ("thread-per-instance"): the whole
 * per-instance CVODES state -- Nordsieck array zn[6][n] (+ quadrature znQ[6][p]), error weights,
 * Newton matrix I - gamma*J with its LU, saved Jacobian, BDF coefficient vectors l/tau/tq,
 * divided-difference table of the stored forward trajectory -- lives in VGPRs; every loop over
 * the order q is unrolled with predicates so no array is indexed dynamically (no scratch).
 * At batch 65 536 there are exactly 1024 wavefronts = one per SIMD of the 256 CUs, so the
 * 512-VGPR budget per lane is free to use.
 *
 * Divergence control: the main loop iterates over step ATTEMPTS (predict / Newton / error
 * test), not over steps, so a lane that rejects a step simply retries in the next iteration
 * while its neighbours move on; the backward pass re-converges the wave once per observation
 * interval where sunode restarts the integrator (solver.py:756-757).
 *
 * Replaces, for a whole batch at once, the work the reference delegates to CVODES through
 *   Solver.solve                   /root/reference/sunode/solver.py:467-527
 *   AdjointSolver.solve_forward    /root/reference/sunode/solver.py:682-721
 *   AdjointSolver.solve_backward   /root/reference/sunode/solver.py:723-784
 * (CVode / CVodeF / CVodeB with SUNLinSol_Dense + analytic Jacobians, CV_POLYNOMIAL
 * interpolation, backward quadratures with error control).
 *
 * Compiled once per problem against the generated callback header:
 *   hipcc --offload-arch=gfx950 --genco -O3 -ffp-contract=off -DSA_PROBLEM_HEADER='"..."'
 * -ffp-contract=off + the deterministic pow below keep the step/order bookkeeping
 * reproducible bit-for-bit against the CPU restatement used by the tests.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SA_FN static __device__ __forceinline__
/* lane families of the generated callbacks (symode/codegen.py find_lane_families): this mapping has ONE lane per
   instance and keeps the callback inputs in register arrays, so the loop over the M members of a family is unrolled at
   compile time -- every index stays a constant (same expression text as in every other mapping) */
template <int I> struct sa_fam_ic { static constexpr int value = I; };
template <int B, int E, class F>
static __device__ __forceinline__ void sa_fam_for(F &&f) { if constexpr (B < E) { f(sa_fam_ic<B>{}); sa_fam_for<B + 1, E>(f); } }
constexpr int sa_tau_c(int f, int j) { return j == 0 ? f : (j == f ? 0 : j); }
#define SA_FAM_BEGIN(M) sa_fam_for<0, (M)>([&](auto sa_fic_) __attribute__((always_inline)) { constexpr int sa_f = decltype(sa_fic_)::value;
#define SA_F sa_f
#define SA_TAU(j) sa_tau_c(sa_f, (j))
#define SA_FAM_STORE(S0, value) { const double v_ = (value); out[(S0) + sa_f] = v_; chk += v_ * 0.0; }
#define SA_FAM_END });
#include SA_PROBLEM_HEADER
#include "sa_device_abi.h"

#include "sa_common.h"

/* Optional phase timing (SA_KERNEL_DEFINES=-DSA_ABLATE_PROFILE, tools/profile_lv.py): clock deltas per segment of
   the attempt loop, written over stats slots 8..15 of the backward kernel.  PHASE(m, k) closes the segment that ENDS
   at mark k (static slot, so the counters stay in registers); a lane idling while its neighbours run a divergent
   block charges that time to the next mark it reaches. */
#ifdef SA_ABLATE_PROFILE
#define PROF_DECL long long prof[8]; long long prof_last; int prof_cur;
#define PHASE(m, k) do { long long now_ = __builtin_readcyclecounter(); (m).prof[k] += now_ - (m).prof_last; \
                         (m).prof_last = now_; } while (0)
#else
#define PROF_DECL
#define PHASE(m, k) do { } while (0)
#endif
/* -DSA_ABLATE_PROFILE -DSA_INTERP_PROFILE: the interpolation of the backward kernel split further (slots 1..5: index
   search | the rebuild's point loads until they have arrived | table arithmetic | table + touches stored | evaluation);
   every other mark of the attempt loop is booked on slot 0, the interval ends stay on slot 7 (tools/profile_lv.py) */
/* -DSA_SEARCH_COUNT: the index search's dependent global loads are counted INSTEAD of the interpolations / rebuilds
   (stats slots 11 / 12: loads of moves to the left / to the right) -- a diagnostic build, tools/profile_lv.py */
#ifdef SA_SEARCH_COUNT
#define SEARCH_COUNT(c) (c)++
#define INTERP_COUNT(c) do { } while (0)
#else
#define SEARCH_COUNT(c) do { } while (0)
#define INTERP_COUNT(c) (c)++
#endif
/* Index search of the backward interpolation (interp_y): CVAfindIndex walks from the last index one stored point at a
   time, and on this device every further point is a DEPENDENT global load (~1 us) that the whole wavefront waits for.
   Where the backward steps are long against the forward steps (Robertson: 801 + 258 such loads per instance in 1 235
   attempts, some lane of the 64 in nearly every iteration; profiles/r06_interp_search.txt) that walk was 21 % of the
   backward kernel.  SA_SEARCH_CACHE: the same index without the walk -- times from the current table, the remembered
   right neighbour, then galloping + section search with independent probes (search_left / search_right below).
   Measured beside it and not kept (same file): 1 / 2 / 8 probes per round, a contiguous first round, 8 / 16 further times
   parked in LDS at every rebuild (fewer walks, but the extra scattered loads of every rebuild cost more).
   On for the compact-record builds (three states and more); the table-record builds (two states: 13 backward steps per
   stored point, 8 + 9 such loads per instance) keep the plain walk, which is 1 % faster there.  Hermite builds: off
   (no divided-difference table to take the times from). */
#ifndef SA_SEARCH_CACHE
#if defined(SA_COMPACT_TRAJ) && !defined(SA_HERMITE)
#define SA_SEARCH_CACHE 1
#else
#define SA_SEARCH_CACHE 0
#endif
#endif
#if defined(SA_ABLATE_PROFILE) && defined(SA_INTERP_PROFILE)
#define IPH(m, k) PHASE(m, k)
#else
#define IPH(m, k) do { } while (0)
#endif

/* ------------------------------------------------------------------------------------ */
/* per-lane integrator state                                                              */
/* ------------------------------------------------------------------------------------ */
/* The saved Jacobian (cvLsSetup reuses it while jbad is false) is read and written only at matrix set-ups: from three
   states on it waits in LDS, [entry][lane] (conflict-free), instead of in n*n register pairs. */
#ifndef SA_SJ_LDS
#define SA_SJ_LDS (NS >= 3)
#endif
/* The divided-difference table of the current interpolation index: an LDS column per lane (the default from three
   states on: 20 + 6n register pairs are worth more there) or, for two states, 20 doubles in REGISTERS -- the backward
   kernel of Lotka-Volterra has 120 registers to spare at its one wavefront per SIMD, and a lone wavefront waits out
   every LDS round trip of the 20 reads per interpolation (round 5). */
#ifndef SA_TAB_REGS
#define SA_TAB_REGS (NS <= 2)
#endif
#if SA_TAB_REGS
#define LT_(m, f) (m).tabr[f]
#else
#define LT_(m, f) (m).ltab[(f) * 64]
#endif
#if SA_SJ_LDS
__shared__ double s_savedJ[NS * NS * 64];
#define SAVEDJ(m, i) s_savedJ[(i) * 64 + threadIdx.x]
#else
#define SAVEDJ(m, i) (m).savedJ[i]
#endif

template <bool BWD>
struct Cv {
    /* Invariant: columns j > q of zn / znQ are exactly zero (CVODES leaves stale data there and
       parks the saved correction Delta_n in zn[qmax]; here that lives in zsave / zsaveQ).  With
       zero columns above the order, predict / restore / rescale / correct / interpolate run over all
       six columns without per-lane predicates: adding or scaling a zero column is exact. */
    double zn[QMAX + 1][NSD];
    double znQ[QMAX + 1][NQD];
    double zsave[NSD], zsaveQ[NQD];
    double ewt[NSD], acor[NSD], tempv[NSD], ftemp[NSD], y[NSD];
#ifdef SA_CONSTRAINTS
    double cons[NSD];                 /* CVodeSetConstraints vector (all zero: none) */
    int constr;
#endif
    double ewtQ[NQD], acorQ[NQD], tempvQ[NQD];
    double ytmp[NSD];                 /* interpolated forward state (backward only) */
    double atol[NSD];
    double rtol, rtolQ, atolQ;
    double tn, h, hprime, hscale, eta, etamax, hu;
    int q, qprime, L, qwait, qu;
    double tau[7], tq[6], l[7];
    double rl1, gamma, gammap, gamrat, crate, delp, acnrm, saved_tq5;
    double etaq, etaqm1, etaqp1;
    double tstop;                     /* backward: forward t0 (CVodeSetStopTime in CVodeB) */
    int nst, nfe, nje, nsetups, nni, ncfn, netf, nfQe, netfQ, nstlp, nstlj;
    double A[NSD * NSD];
#if !SA_SJ_LDS
    double savedJ[NSD * NSD];
#endif
    int piv[NSD];
    double inv_piv[NSD];
    int jcur, nls_jcur;
    double ps[NQD];
    double prl[NR <= 32 ? NRD : 1];   /* remaining parameters held per lane (small NR) */
    const double *prg;            /* or read through a (typically shared) global pointer (large NR) */
    /* trajectory interpolation (backward) */
    const double *traj;               /* lane's first record; see "stored forward trajectory" */
    int64_t trow;                     /* step pitch in doubles */
    int cur_idx;                      /* index whose divided-difference table is in use */
    double pf[4];                     /* in-flight prefetch touches (never consumed as data) */
#ifdef SA_HERMITE                     /* CV_HERMITE: cubic on [t0,t1] from y, y' at both ends */
    double h_t0, h_t1, h_y0[NSD], h_yd0[NSD], h_Y0[NSD], h_Y1[NSD];
    double f0[NSD];                   /* f(t0, y0) of the first stored point */
#endif
    double *ltab;                     /* this lane's column of the LDS table copy: ltab[field * 64] */
#if SA_TAB_REGS
    double tabr[8 + 6 * NS];          /* ... or the table in registers (see SA_TAB_REGS) */
#endif
    double tlo2;                      /* t[ilast-2] */
#if SA_SEARCH_CACHE
    double thi2; int thi2_idx;        /* t[thi2_idx], the time right of the bracket after a move to the left (-1: unknown) */
#endif
    int np;
    double tfinal;
    int ilast, newdata, have_last;
    double last_t;
    double tlo, thi;                  /* t[ilast-1], t[ilast]: bracketing times kept in registers */
    int n_interp, n_rebuild;
#ifdef SA_SENS
    /* forward sensitivities (Solver(sens_mode=...), reference solver.py:360-392): one Nordsieck array per
       differentiated parameter, kept in registers like the state's (columns above q at zero, the saved correction
       in zsaveS); only the SA_SENS build of the forward problem carries them */
    double sv[SV_COUNT][NQD][NSD];    /* SV(m, vector, parameter, component): SV_ZN0..5, SV_ZSAVE, SV_EWT, ... (sa_common.h) */
    double pbar[NQD], crateS, delpS, acnrmS;
    int sensi, ism, nfSe, nniS, ncfnS, netfS, nsetupsS;
#endif
    PROF_DECL
};

#ifdef SA_SENS
#define SENS_ON(m) (!BWD && (m).sensi)
#define SFOR_S(is, i) SFOR(is, 0, NQ) SFOR(i, 0, NS)
#define SEND_S SEND SEND
#endif

/* ---- stored forward trajectory ------------------------------------------------------------
 * CVODES (CV_POLYNOMIAL) stores (t_n, y_n, q_n) after every forward step and, in the backward
 * wrappers, rebuilds a Newton divided-difference table through the q+1 points ending at the bracketing
 * index whenever that index changes (CVApolynomialGetY).  Doing that rebuild inside the backward
 * kernel is the worst case for a thread-per-instance mapping: ~88 rebuilds per instance against
 * ~1500 loop iterations per wave means SOME lane rebuilds in nearly every iteration, so the whole wave
 * pays 15 divides + dependent loads every time.  Instead the FORWARD kernel, whose lanes all store
 * a point in the same iteration, builds the table of every index once and stores it:
 *
 *   record (step s, instance) = TREC = 8 + 6n doubles: {order, dt, T[0..5], Y[0..5][n]}
 *     T[j] = t_{s-j},  Y = scaled divided differences through points s..s-order,  dt = |t_s - t_{s-1}|
 *   traj[(s * stride + inst) * TREC + f], instance index fastest.
 *
 * The backward kernel then only COPIES a table: when a lane's table index moves (~every 13th
 * attempt) it copies the 8 + 6n doubles of the new record from HBM/L2 into its private column of an
 * LDS array ltab[field][lane] (10 KB per wave, conflict-free: the bank depends on the lane only), and
 * every interpolation evaluates the polynomial from LDS.  No table occupies VGPRs across
 * iterations, nothing is rebuilt in the divergent part of the loop, and the index search of the
 * common move (one index to the left) needs no load at all (t[idx-1], t[idx-2] ride along in
 * registers).  Same values as CVODES computes on demand, hence bit-identical to the oracle.
 */
#if defined(SA_COMPACT_TRAJ) && !defined(SA_HERMITE)
/* -DSA_COMPACT_TRAJ (AdjointSolver(compact_trajectory=True)): the arena holds what CVODES itself stores per step and
 * SURVEY 8(d) counts -- {order, t, y[n]}, n + 2 doubles instead of 8 + 6n -- and the BACKWARD kernel rebuilds the
 * divided-difference table into its LDS column when the index moves (the same operations in the same order the forward
 * kernel performs otherwise: build_table below), from the order + 1 points ending at the index: one contiguous block of
 * the instance's records.  5.2x (n = 3) ... 5.8x (n = 16) less arena and forward write traffic, no table build and no
 * point history in the forward kernel; the price is the rebuild in the divergent part of the backward loop (the
 * reason the table records exist).  Measured A/B: profiles/r03_compact_trajectory.txt. */
#define SA_COMPACT 1
#define TREC (NS + 2)
#define TREC_T 1
#define TREC_Y 2
#else
#define SA_COMPACT 0
#define TREC (8 + 6 * NS)
#define TREC_T 2
#define TREC_Y 8
#endif
#define TTAB (8 + 6 * NS)            /* the table in LDS: {order, dt, T[6], Y[6][n]} */
#define LT(m, f) LT_(m, f)

template <bool BWD>
DEV double point_time(const Cv<BWD> &m, int s) { return m.traj[(int64_t)s * m.trow + TREC_T]; }

/* Newton divided differences of CVApolynomialGetY over the points hT[j], hY[j] (j = 0 newest .. order), scaled by
   dt^j: factor = dt / (T[j] - T[j-i]), Y[j] = factor * (Y[j] - Y[j-1]) -- the oracle's operation order */
DEV void build_table(int order, double dt, const double (&hT)[QMAX + 1], double (&Y)[QMAX + 1][NSD])
{
    SFOR(i, 1, (QMAX) + 1) {
        SFOR_DOWN(j, QMAX, 1) {
            if constexpr (j >= i) {
                if (j <= order) {
                    double factor = SA_TABLE_DIV(dt, hT[j] - hT[j - i]);
                    SFOR(k, 0, NS) Y[j][k] = factor * (Y[j][k] - Y[j - 1][k]); SEND
                }
            }
        } SEND
    } SEND
}

#if defined(SA_ABLATE_PROFILE) && defined(SA_INTERP_PROFILE)
/* the values have ARRIVED (loads) / are COMPUTED before the clock is read */
DEV void prof_pin(const double (&hT)[QMAX + 1], const double (&Y)[QMAX + 1][NSD])
{
#pragma unroll
    for (int j = 0; j <= QMAX; j++) {
        asm volatile("" :: "v"(hT[j]));
#pragma unroll
        for (int k = 0; k < NS; k++) asm volatile("" :: "v"(Y[j][k]));
    }
}
#endif

#if SA_SEARCH_CACHE
/* The index CVAfindIndex's walk to the LEFT ends at, without walking: the first k in [0, hi] with (t - t[k]) <= 0, given
   that the comparison holds at hi (thv = t[hi] on entry).  The stored times increase strictly, so the walk's
   comparisons are monotone in k and any probing order finds the same index: SA_SEARCH_PROBES independent loads per round
   (one memory latency per round, not per probe) -- galloping to the left of hi (distances 2, 4, 8, ...), then a section
   search of the bracket.  On return thv = t[k] and, for k > 0, tlv = t[k-1] (both were probed on the way). */
#ifndef SA_SEARCH_PROBES
#define SA_SEARCH_PROBES 4
#endif
template <bool BWD>
DEV int search_left(Cv<BWD> &m, double t, int hi, double &thv, double &tlv)
{
    constexpr int NP = SA_SEARCH_PROBES;
    int lo = -1;
    int64_t step = 2;
    while (lo < 0 && hi > 0) {
        int k[NP];
        double v[NP];
        SFOR(j, 0, NP) {
            const int64_t kk = (int64_t)hi - (step << j);
            k[j] = kk > 0 ? (int)kk : 0;
            v[j] = point_time(m, k[j]); SEARCH_COUNT(m.n_interp);
        } SEND
        bool stop = false;
        SFOR(j, 0, NP) {                      /* nearest first */
            if (!stop) {
                if ((t - v[j]) <= 0.0) { hi = k[j]; thv = v[j]; }
                else { lo = k[j]; tlv = v[j]; stop = true; }
            }
        } SEND
        step = (step < ((int64_t)1 << 40)) ? (step << NP) : step;
    }
    while (hi - lo > 1) {
        const int n = hi - lo, lo0 = lo, hi0 = hi;
        int k[NP];
        double v[NP];
        SFOR(j, 0, NP) {
            int kk = lo0 + (int)(((int64_t)n * (j + 1)) / (NP + 1));
            kk = kk < lo0 + 1 ? lo0 + 1 : kk;
            kk = kk > hi0 - 1 ? hi0 - 1 : kk;
            k[j] = kk;
            v[j] = point_time(m, kk); SEARCH_COUNT(m.n_interp);
        } SEND
        bool found = false;
        SFOR(j, 0, NP) {                      /* ascending */
            if (!found) {
                if ((t - v[j]) <= 0.0) { hi = k[j]; thv = v[j]; found = true; }
                else { lo = k[j]; tlv = v[j]; }
            }
        } SEND
    }
    return hi;
}
/* ... to the RIGHT: the first k in (lo, np-1] with (t - t[k]) <= 0, or np-1 when there is none (the walk stops at the
   last point); (t - t[lo]) > 0 on entry with tlv = t[lo].  On return thv = t[k], tlv = t[k-1]. */
template <bool BWD>
DEV int search_right(Cv<BWD> &m, double t, int lo, double &tlv, double &thv)
{
    constexpr int NP = SA_SEARCH_PROBES;
    const int last = m.np - 1;
    int hi = -1;
    int64_t step = 1;
    bool ran_off = false;
    while (hi < 0) {
        int k[NP];
        double v[NP];
        SFOR(j, 0, NP) {
            const int64_t kk = (int64_t)lo + (step << j);
            k[j] = kk < last ? (int)kk : last;
            v[j] = point_time(m, k[j]); SEARCH_COUNT(m.n_rebuild);
        } SEND
        bool stop = false;
        SFOR(j, 0, NP) {                      /* nearest first */
            if (!stop) {
                if ((t - v[j]) > 0.0) {
                    lo = k[j]; tlv = v[j];
                    if (k[j] == last) { hi = last; thv = v[j]; stop = true; ran_off = true; }
                } else { hi = k[j]; thv = v[j]; stop = true; }
            }
        } SEND
        step = (step < ((int64_t)1 << 40)) ? (step << NP) : step;
    }
    if (ran_off) { lo = last - 1; tlv = point_time(m, lo); }    /* (t beyond the last stored point) */
    while (hi - lo > 1) {
        const int n = hi - lo, lo0 = lo, hi0 = hi;
        int k[NP];
        double v[NP];
        SFOR(j, 0, NP) {
            int kk = lo0 + (int)(((int64_t)n * (j + 1)) / (NP + 1));
            kk = kk < lo0 + 1 ? lo0 + 1 : kk;
            kk = kk > hi0 - 1 ? hi0 - 1 : kk;
            k[j] = kk;
            v[j] = point_time(m, kk); SEARCH_COUNT(m.n_rebuild);
        } SEND
        bool found = false;
        SFOR(j, 0, NP) {                      /* ascending */
            if (!found) {
                if ((t - v[j]) > 0.0) { lo = k[j]; tlv = v[j]; }
                else { hi = k[j]; thv = v[j]; found = true; }
            }
        } SEND
    }
    return hi;
}
#endif

/* CVAfindIndex + CVApolynomialGetY (forward integration direction), with the wrappers'
   repeated interpolation at an unchanged t evaluated once. */
template <bool BWD>
DEV int interp_y(Cv<BWD> &m, double t)
{
    if (m.have_last && t == m.last_t) return CV_SUCCESS;
#ifdef SA_ABLATE_INTERP          /* timing experiment: no trajectory access at all */
    m.have_last = 1; m.last_t = t;
    SFOR(i, 0, NS) m.ytmp[i] = 1.0 + 0.001 * t; SEND
    return CV_SUCCESS;
#endif
    INTERP_COUNT(m.n_interp);
    IPH(m, 0);
    int newpoint = 0, indx;
    if (m.newdata) {
        m.ilast = m.np - 1; newpoint = 1; m.newdata = 0;
        m.tlo = point_time(m, m.ilast - 1); m.thi = point_time(m, m.ilast);
        m.tlo2 = (m.ilast >= 2) ? point_time(m, m.ilast - 2) : m.tlo;
    }
    const int ilast = m.ilast;
    bool to_left = (t - m.tlo) < 0.0;
    bool to_right = (t - m.thi) > 0.0;
    indx = ilast;
    if (to_left) {
        newpoint = 1;
        double tprev = m.tlo;                 /* t[indx-1] */
        double tcur = m.thi;                  /* t[indx]   */
#if SA_SEARCH_CACHE
        /* CVAfindIndex walks one index at a time; here the walk costs no dependent global load for the first QMAX
           indices (their times are in the table of the current index, T[j] = t[cur_idx - j], read once) and continues
           as a galloping + binary search (search_left) -- the same index, t[indx-1] and t[indx] as the walk's */
        const bool tab_ok = (m.cur_idx == ilast);
        const double tb3 = LT(m, 5), tb4 = LT(m, 6), tb5 = LT(m, 7);
        double tright = m.thi2;
        bool far = false;
        for (;;) {
            if (indx == 0) break;
            if ((t - tprev) <= 0.0) {
                indx--;
                tright = tcur;                /* t[indx+1] */
                tcur = tprev;
                if (indx > 0) {
                    const int back = ilast - indx + 1;            /* t[indx-1] = t[ilast - back] */
                    if (back == 2) tprev = m.tlo2;
                    else if (tab_ok && back <= QMAX) tprev = (back == 3) ? tb3 : ((back == 4) ? tb4 : tb5);
                    else { far = true; break; }
                }
            } else break;
        }
        int right_idx = indx + 1;
        if (far) {                            /* t <= t[indx] = tcur, t[indx-1] not at hand */
            const int k = search_left(m, t, indx, tcur, tprev);
            if (k != indx) right_idx = -1;
            indx = k;
        }
        m.thi2 = tright; m.thi2_idx = right_idx;
#else
        for (;;) {
            if (indx == 0) break;
            if ((t - tprev) <= 0.0) {
                indx--;
                tcur = tprev;
                if (indx > 0) {
                    if (indx == ilast - 1) tprev = m.tlo2;
                    else { tprev = point_time(m, indx - 1); SEARCH_COUNT(m.n_interp); }
                }
            } else break;
        }
#endif
        m.ilast = (indx == 0) ? 1 : indx;
        if (indx == 0) {
            /* tcur = t[0]; CVODES leaves ilast = 1 here */
            m.tlo = tcur; m.thi = point_time(m, 1);
            if (fabs(t - m.tlo) > FUZZ_FACTOR_ADJ * UROUND) return CV_GETY_BADT;
        } else {
            m.tlo = tprev; m.thi = tcur;
        }
    } else if (to_right) {
        newpoint = 1;
        double tcur = m.thi;                  /* t[indx] */
        double tprev = m.tlo;
#if SA_SEARCH_CACHE
        /* the first index to the right: t[ilast+1] is remembered from the last move to the left (the retry of a
           rejected attempt steps back over it); further: galloping + binary search (search_right) */
        if (indx < m.np - 1) {                /* ((t - tcur) > 0 holds: to_right) */
            indx++;
            tprev = tcur;
            if (indx == m.thi2_idx) tcur = m.thi2;
            else { tcur = point_time(m, indx); SEARCH_COUNT(m.n_rebuild); }
            if (indx < m.np - 1 && (t - tcur) > 0.0) { tprev = tcur; indx = search_right(m, t, indx, tprev, tcur); }
        }
        m.thi2_idx = -1;                      /* t[indx+1] is not known after a move to the right */
#else
        for (;;) {
            if (indx >= m.np - 1) break;
            if ((t - tcur) > 0.0) {
                indx++;
                tprev = tcur;
                tcur = point_time(m, indx); SEARCH_COUNT(m.n_rebuild);
            } else break;
        }
#endif
        m.ilast = indx;
        m.tlo = tprev; m.thi = tcur;
        if ((t - m.thi) > FUZZ_FACTOR_ADJ * UROUND * (fabs(m.tfinal) + 1.0)) return CV_GETY_BADT;
    }
    m.have_last = 1;
    m.last_t = t;
    IPH(m, 1);
    if (indx == 0) {
        SFOR(i, 0, NS) m.ytmp[i] = m.traj[TREC_Y + i]; SEND   /* record 0: Y[0] = y(t0) */
        return CV_SUCCESS;
    }
#ifdef SA_HERMITE
    {   /* CVAhermiteGetY (see the oracle; same arithmetic as bdf_wave.hip / bdf_mem.hip) */
        if (newpoint) {
            INTERP_COUNT(m.n_rebuild);
            m.cur_idx = indx;
            const double *r0 = m.traj + (int64_t)(indx - 1) * m.trow, *r1 = m.traj + (int64_t)indx * m.trow;
            m.h_t0 = r0[2]; m.h_t1 = r1[2];
            const double delta = m.h_t1 - m.h_t0;
            SFOR(c, 0, NS) {
                m.h_y0[c] = r0[8 + c]; m.h_yd0[c] = r0[8 + NS + c];
                const double y1 = r1[8 + c], yd1 = r1[8 + NS + c];
                const double dy = y1 - m.h_y0[c];
                m.h_Y0[c] = FMA(-delta, m.h_yd0[c], dy);
                m.h_Y1[c] = FMA(delta, yd1 + m.h_yd0[c], -2.0 * dy);
            } SEND
            if (indx == m.ilast) m.tlo2 = (indx >= 2) ? point_time(m, indx - 2) : m.tlo;
        }
        const double delta = m.h_t1 - m.h_t0;
        const double factor1 = t - m.h_t0;
        double factor2 = factor1 / delta;
        factor2 = factor2 * factor2;
        const double factor3 = factor2 * (t - m.h_t1) / delta;
        SFOR(c, 0, NS) {
            double acc = FMA(factor1, m.h_yd0[c], m.h_y0[c]);
            acc = FMA(factor2, m.h_Y0[c], acc);
            acc = FMA(factor3, m.h_Y1[c], acc);
            m.ytmp[c] = acc;
        } SEND
        return CV_SUCCESS;
    }
#endif
    if (newpoint) {
        INTERP_COUNT(m.n_rebuild);
        m.cur_idx = indx;                     /* the table CVODES would rebuild now */
        const double *r = m.traj + (int64_t)indx * m.trow;
        /* the touches issued at the previous move have long landed: retire them (keeps their
           destination registers reserved until now, i.e. the loads were never waited for early) */
        asm volatile("" :: "v"(m.pf[0]), "v"(m.pf[1]), "v"(m.pf[2]), "v"(m.pf[3]));
#if SA_COMPACT
        {   /* rebuild: the order + 1 points ending at indx (all loads in flight together; unused columns zero).
               (Measured and not kept: reading the six points as ONE contiguous block of 6 (n + 2) doubles at
               constant offsets from a single base -- an instance's records are contiguous -- with a per-point
               fall-back for indx < 5: Robertson backward 68.4 -> 70.8 ms.) */
            const int order = (int)r[0];
            double hT[QMAX + 1], Y[QMAX + 1][NSD];
            SFOR(j, 0, (QMAX) + 1) {
                const double *rj = m.traj + (int64_t)(indx - j > 0 ? indx - j : 0) * m.trow;
                hT[j] = rj[TREC_T];
                SFOR(k, 0, NS) { const double v = rj[TREC_Y + k]; Y[j][k] = (j <= order) ? v : 0.0; } SEND
            } SEND
#if defined(SA_ABLATE_PROFILE) && defined(SA_INTERP_PROFILE)
            prof_pin(hT, Y);
            IPH(m, 2);
#endif
            const double dt = fabs(hT[0] - hT[1]);
            build_table(order, dt, hT, Y);
#if defined(SA_ABLATE_PROFILE) && defined(SA_INTERP_PROFILE)
            prof_pin(hT, Y);
            IPH(m, 3);
#endif
            LT(m, 0) = (double)order;
            LT(m, 1) = dt;
            SFOR(j, 0, (QMAX) + 1) LT(m, 2 + j) = hT[j]; SEND
            SFOR(j, 0, (QMAX) + 1) { SFOR(k, 0, NS) LT(m, 8 + j * NS + k) = Y[j][k]; SEND } SEND
            const double *rn = r - (indx > QMAX + 1 ? (QMAX + 1) * m.trow : 0);     /* the point the next move adds */
            m.pf[0] = rn[0]; m.pf[1] = rn[TREC - 1]; m.pf[2] = m.pf[0]; m.pf[3] = m.pf[1];
        }
#else
        SFOR(f, 0, TREC) LT(m, f) = r[f]; SEND
        {   /* touch the record of the next index to the left so that it is L2-resident when needed */
            const double *rn = r - (indx > 0 ? m.trow : 0);
            m.pf[0] = rn[0]; m.pf[1] = rn[TREC / 3]; m.pf[2] = rn[2 * TREC / 3]; m.pf[3] = rn[TREC - 1];
        }
#endif
        if (LT(m, 0) > (double)indx) return CV_GETY_BADT;    /* CVODES would shift the base; cannot occur */
        if (indx == m.ilast) m.tlo2 = LT(m, 4);              /* T[2] = t[ilast-2] for the next move */
        IPH(m, 4);
    }
    {
        /* every LDS read of the record up front, in ONE batch: with the reads inside the conditional expressions
           the compiler built a branch chain around them -- seven dependent LDS round trips per interpolation */
        double hdr[8], Yt[QMAX + 1][NSD];
        SFOR(f, 0, 8) hdr[f] = LT(m, f); SEND
        SFOR(i, 0, (QMAX) + 1) { SFOR(k, 0, NS) Yt[i][k] = LT(m, 8 + i * NS + k); SEND } SEND
        const int order = (int)hdr[0];
        const double inv_dt = SA_TABLE_DIV(1.0, hdr[1]);
        double cvals[QMAX + 1];
        cvals[0] = 1.0;
        SFOR(i, 0, QMAX) { const double v = cvals[i] * (t - hdr[2 + i]) * inv_dt; cvals[i + 1] = (i < order) ? v : 0.0; } SEND
        SFOR(k, 0, NS) {
            double acc = cvals[0] * Yt[0][k];
            SFOR(i, 1, (QMAX) + 1) acc = FMA(cvals[i], Yt[i][k], acc); SEND
            m.ytmp[k] = acc;
        } SEND
    }
    return CV_SUCCESS;
}

/* remaining parameters: registers when few, global pointer when many (e.g. a shared 100x100 K) */
#define SA_REM_IN_REGS (NR <= 32)
template <bool BWD>
DEV const double *pr_of(const Cv<BWD> &m)
{
    if constexpr (SA_REM_IN_REGS) return m.prl;
    else return m.prg;
}
#define PR_OF(m) pr_of(m)

/* ---- callbacks as the integrator sees them (backward: y(t) must have been interpolated) ---- */
template <bool BWD>
DEV int cv_f(Cv<BWD> &m, double t, const double *y, double *out)
{
    m.nfe++;
    if (BWD) return sa_adj_rhs(t, m.ytmp, y, m.ps, PR_OF(m), out);
    return sa_rhs(t, y, m.ps, PR_OF(m), out);
}

template <bool BWD>
DEV int cv_fQ(Cv<BWD> &m, double t, const double *y, double *out)
{
    m.nfQe++;
    return sa_quad_rhs(t, m.ytmp, y, m.ps, PR_OF(m), out);
}

template <bool BWD>
DEV int cv_jac(Cv<BWD> &m, double t, const double *y, double *J)
{
    if (BWD) return sa_adj_jac(t, m.ytmp, m.ps, PR_OF(m), J);
    return sa_jac(t, y, m.ps, PR_OF(m), J);
}

#ifdef SA_SENS
/* sensitivity right-hand side for all parameters: SV(v_out)[is] = J(t,y) SV(v_in)[is] + df/dp_is (oracle cv_fS) */
template <int v_in, int v_out, bool BWD>
DEV int cv_fS(Cv<BWD> &m, double t, const double *y)
{
    m.nfSe++;
    double Jt[NSD * NSD], dp[NQD * NSD];
    int rc = sa_jac(t, y, m.ps, PR_OF(m), Jt);
    if (rc != 0) return rc;
    rc = sa_dydp(t, y, m.ps, PR_OF(m), dp);
    int bad = 0;
    SFOR_S(is, i) {
        double acc = Jt[0 * NS + i] * m.sv[v_in][is][0];
        SFOR(j, 1, NS) acc = FMA(Jt[j * NS + i], m.sv[v_in][is][j], acc); SEND
        acc = acc + dp[is * NS + i];
        m.sv[v_out][is][i] = acc;
        bad |= !(acc * 0.0 == 0.0);
    } SEND_S
    return (rc != 0 || bad) ? 1 : 0;
}
#endif

/* ---- vector kernels ---- */
/* balanced-tree sum over P = 2^k leaves (the association of a cross-lane butterfly; see the oracle) */
template <int P>
DEV double tree_sum(double (&buf)[P])
{
    if constexpr (P > 1) {
        double half[P / 2];
        SFOR(i, 0, P / 2) half[i] = buf[2 * i] + buf[2 * i + 1]; SEND
        return tree_sum<P / 2>(half);
    } else {
        return buf[0];
    }
}

constexpr int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

template <int N>
DEV double wrms(const double *x, const double *w)
{
    if constexpr (N == 0) return 0.0;
    else {
        constexpr int P = next_pow2(N);
        double leaf[P];
        SFOR(i, 0, P) {
            if constexpr (i < N) { double prod = x[i] * w[i]; leaf[i] = prod * prod; }
            else leaf[i] = 0.0;
        } SEND
        return sqrt(tree_sum<P>(leaf) / N);
    }
}

/* mean square of the weighted vector: wrms<N>(x, w) == sqrt(wms<N>(x, w)), same tree, same division */
template <int N>
DEV double wms(const double *x, const double *w)
{
    if constexpr (N == 0) return 0.0;
    else {
        constexpr int P = next_pow2(N);
        double leaf[P];
        SFOR(i, 0, P) {
            if constexpr (i < N) { double prod = x[i] * w[i]; leaf[i] = prod * prod; }
            else leaf[i] = 0.0;
        } SEND
        return tree_sum<P>(leaf) / N;
    }
}

template <bool BWD>
DEV double quad_update_norm(const Cv<BWD> &m, double old_nrm, const double *xQ)
{
    double qnrm = wrms<NQ>(xQ, m.ewtQ);
    return old_nrm > qnrm ? old_nrm : qnrm;
}

template <bool BWD>
DEV int ewt_set(const Cv<BWD> &m, const double *ycur, double *w)
{
    int bad = 0;
    SFOR(i, 0, NS) {
        double v = FMA(m.rtol, fabs(ycur[i]), m.atol[i]);
        bad |= (v <= 0.0);
        w[i] = 1.0 / v;
    } SEND
    return bad ? -1 : 0;
}

template <bool BWD>
DEV int ewtQ_set(const Cv<BWD> &m, const double *qcur, double *w)
{
    int bad = 0;
    SFOR(i, 0, NQ) {
        double v = FMA(m.rtolQ, fabs(qcur[i]), m.atolQ);
        bad |= (v <= 0.0);
        w[i] = 1.0 / v;
    } SEND
    return bad ? -1 : 0;
}


/* ---- dense LU with partial pivoting, column-major, fully unrolled (denseGETRF/GETRS) ---- */
DEV int dense_getrf(double *a, int *p, double *inv_piv)
{
    int ier = 0;
    SFOR(k, 0, NS) {
        int l = k;
        double best = fabs(a[k * NS + k]);
        SFOR(i, k + 1, NS) {
            double v = fabs(a[k * NS + i]);
            if (v > best) { best = v; l = i; }
        } SEND
        p[k] = l;
        double pivot = a[k * NS + k];
        SFOR(i, k + 1, NS) pivot = (l == i) ? a[k * NS + i] : pivot; SEND
        if (pivot == 0.0 && ier == 0) ier = k + 1;
        if (ier == 0) {
            if (l != k) {
                SFOR(c, 0, NS) {
                    double akc = a[c * NS + k];
                    double alc = akc;
                    SFOR(i, k + 1, NS) alc = (l == i) ? a[c * NS + i] : alc; SEND
                    SFOR(i, k + 1, NS) a[c * NS + i] = (l == i) ? akc : a[c * NS + i]; SEND
                    a[c * NS + k] = alc;
                } SEND
            }
            double mult = 1.0 / a[k * NS + k];
            inv_piv[k] = mult;
            SFOR(i, k + 1, NS) a[k * NS + i] *= mult; SEND
            SFOR(j, k + 1, NS) {
                double a_kj = a[j * NS + k];
                if (a_kj != 0.0) {
                    SFOR(i, k + 1, NS) a[j * NS + i] = FMA(-a_kj, a[k * NS + i], a[j * NS + i]); SEND
                }
            } SEND
        }
    } SEND
    return ier;
}

DEV void dense_getrs(const double *a, const int *p, const double *inv_piv, double *b)
{
    SFOR(k, 0, NS) {
        int pk = p[k];
        if (pk != k) {
            double bk = b[k];
            double bp = bk;
            SFOR(i, k + 1, NS) bp = (pk == i) ? b[i] : bp; SEND
            SFOR(i, k + 1, NS) b[i] = (pk == i) ? bk : b[i]; SEND
            b[k] = bp;
        }
    } SEND
    SFOR(k, 0, NS - 1) {
        SFOR(i, k + 1, NS) b[i] = FMA(-a[k * NS + i], b[k], b[i]); SEND
    } SEND
    SFOR_DOWN(k, NS - 1, (0) + 1) {
        b[k] *= inv_piv[k];
        SFOR(i, 0, k) b[i] = FMA(-a[k * NS + i], b[k], b[i]); SEND
    } SEND
    if (NS > 0) b[0] *= inv_piv[0];
}

/* ---- linear solver interface (cvLsSetup / cvLsSolve on SUNLinSol_Dense) ---- */
template <bool BWD>
DEV int cv_lsetup(Cv<BWD> &m, int convfail)
{
    double dgamma = fabs((m.gamma / m.gammap) - 1.0);
    int jbad = (m.nst == 0) || (m.nst > m.nstlj + MSBJ) ||
               ((convfail == CV_FAIL_BAD_J) && (dgamma < CVLS_DGMAX)) ||
               (convfail == CV_FAIL_OTHER);
    int jret = 0;
    if (!jbad) {
        m.jcur = 0;
        SFOR(i, 0, NS * NS) m.A[i] = SAVEDJ(m, i); SEND
    } else {
        m.nje++;
        m.nstlj = m.nst;
        m.jcur = 1;
        jret = cv_jac(m, m.tn, m.y, m.A);
        if (jret == 0) { SFOR(i, 0, NS * NS) SAVEDJ(m, i) = m.A[i]; SEND }
    }
    if (jret < 0) return -1;
    if (jret > 0) return 1;
    double c = -m.gamma;
    SFOR(j, 0, NS) {
        SFOR(i, 0, NS) {
            if constexpr (i == j) m.A[j * NS + i] = FMA(c, m.A[j * NS + i], 1.0);
            else m.A[j * NS + i] *= c;
        } SEND
    } SEND
    int ier = dense_getrf(m.A, m.piv, m.inv_piv);
    return ier > 0 ? 1 : 0;
}


/* ---- the mapping bdf_core.h needs: one lane = one instance, a vector = NS doubles in this lane's registers ---- */
#if SA_COMPACT && NS <= 3 && !defined(SA_FWD_OCC1)      /* (n = 5: 265 spill slots under the cap -- not worth it) */
#define SA_FWD_CAP 1                 /* the forward kernel runs at two wavefronts per SIMD (see sa_k_forward) */
#else
#define SA_FWD_CAP 0
#endif
#ifdef SA_POLY_CM_FORCE              /* (A/B switch: -DSA_POLY_CM_FORCE=0|1) */
#define SA_POLY_CM(BWD) (SA_POLY_CM_FORCE)
#else
#define SA_POLY_CM(BWD) (!(BWD) && SA_FWD_CAP)       /* pow coefficients from constant memory (sa_common.h) */
#endif
#define SA_STATE Cv
#define RS NSD
#define RQ NQD
#define IDX(m, r) (r)
#define wave_max(lane, x) (x)
/* (LDS parking of the cold Nordsieck columns / coefficient vectors around the Newton pass, as in the lean lane groups,
   was measured here for the capped forward kernel: not needed once the pow coefficients left the registers --
   Robertson forward 20.9 ms with it, 20.1 ms without) */
#define COLD_STORE(m)
#define COLD_LOAD(m)
#define PH_T0
#if defined(SA_ABLATE_PROFILE) && defined(SA_INTERP_PROFILE)
#define PH_ADD(m, k) PHASE(m, ((k) == 2 ? 5 : 0));
#else
#define PH_ADD(m, k) PHASE(m, k);
#endif
#define SA_RESCALE_ALWAYS 1          /* cv_attempt: the rescale runs for every lane (eta = 1: exact no-op), see there */
#define SA_PRESTEP_FUSED 1           /* cv_pre_step: weights + accuracy test as one straight line (wrms2_n / wrms2_q below) */
template <bool BWD>
DEV double wrms_n(const Cv<BWD> &, const double (&x)[RS], const double (&w)[RS]) { return wrms<NS>(x, w); }
template <bool BWD>
DEV double wrms_q(const Cv<BWD> &, const double (&x)[RQ], const double (&w)[RQ]) { return wrms<NQ>(x, w); }
template <bool BWD>
DEV double wrms2_n(const Cv<BWD> &, const double (&x)[RS], const double (&w)[RS]) { return wms<NS>(x, w); }
template <bool BWD>
DEV double wrms2_q(const Cv<BWD> &, const double (&x)[RQ], const double (&w)[RQ]) { return wms<NQ>(x, w); }
template <bool BWD>
DEV void dense_getrs(const Cv<BWD> &m, double (&b)[RS]) { dense_getrs(m.A, m.piv, m.inv_piv, b); }
#ifdef SA_SENS
#define SV(m, v, is, r) (m).sv[v][is][r]
/* (measured and not kept, profiles/r06_sens_nounroll.txt: the parameter loop as a run-time loop -- the vectors then live in
   scratch by dynamic indexing instead of spilling there, 254 registers and no spill slot -- Robertson 0.91 -> 0.51 M, LV
   23 -> 6 M sensitivity solves/s) */
#define SLOOP_BEGIN(is) SFOR(is, 0, NQ)
#define SLOOP_END SEND
#endif
#include "bdf_core.h"

template <bool BWD>
DEV void load_params(Cv<BWD> &m, const double *ps, const double *pr, int rem_stride, int inst)
{
    SFOR(i, 0, NQ) m.ps[i] = ps[(int64_t)inst * NQ + i]; SEND
    if constexpr (SA_REM_IN_REGS) {
        SFOR(i, 0, NR) m.prl[i] = pr[(int64_t)inst * rem_stride + i]; SEND
        m.prg = nullptr;
    } else {
        m.prl[0] = 0.0;
        m.prg = pr + (int64_t)inst * rem_stride;
    }
}

#define SA_NAN __builtin_bit_cast(double, (uint64_t)0x7ff8000000000000ULL)

/* Build the CVApolynomialGetY divided-difference table of the newest stored point from the point
   history (T[j], Y[j] = point s-j) and write the trajectory record.  Same operation order as the
   on-demand rebuild in the oracle: factor = dt / (T[j] - T[j-i]), Y[j] = factor * (Y[j] - Y[j-1]). */
#ifdef SA_HERMITE
/* CV_HERMITE data point {t, y, y'} in the slots r[2], r[8 + i], r[8 + n + i] of a record; y' = scale * yd
   (f(t0, y0) with scale 1 for the first point, zn[1] / h afterwards) */
DEV void store_hermite(double *r, double t, const double (&y)[NSD], const double (&yd)[NSD], double scale)
{
    r[0] = 0.0; r[1] = 1.0; r[2] = t;
    SFOR(i, 0, NS) { r[8 + i] = y[i]; r[8 + NS + i] = (scale == 1.0) ? yd[i] : scale * yd[i]; } SEND
}
#endif

DEV void store_table(double *r, int order, double dt, const double (&hT)[QMAX + 1], const double (&hY)[QMAX + 1][NSD])
{
    double Y[QMAX + 1][NSD];
    SFOR(j, 0, (QMAX) + 1) { SFOR(i, 0, NS) Y[j][i] = hY[j][i]; SEND } SEND
    build_table(order, dt, hT, Y);
    r[0] = (double)order;
    r[1] = dt;
    SFOR(j, 0, (QMAX) + 1) r[2 + j] = hT[j]; SEND
    SFOR(j, 0, (QMAX) + 1) { SFOR(i, 0, NS) r[8 + j * NS + i] = Y[j][i]; SEND } SEND
}

/* (measured, Robertson B = 262 144: streaming / non-temporal stores for the arena records make the forward kernel
   50 % SLOWER, 20.4 -> 31.1 ms -- plain stores; -DSA_ARENA_NT switches them on) */
#ifdef SA_ARENA_NT
#define ARENA_ST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define ARENA_ST(p, v) (*(p) = (v))
#endif
#if SA_COMPACT
DEV void store_point(double *r, int order, double t, const double (&y)[NSD])
{
    ARENA_ST(&r[0], (double)order);
    ARENA_ST(&r[TREC_T], t);
    SFOR(i, 0, NS) ARENA_ST(&r[TREC_Y + i], y[i]); SEND
}
#endif

/* ------------------------------------------------------------------------------------ */
/* forward kernel: Solver.solve (mode PLAIN) / AdjointSolver.solve_forward (mode ADJ_FWD)   */
/* ------------------------------------------------------------------------------------ */
/* Compact-record builds of three-state systems (long trajectories, batches of several wavefronts per SIMD) cap the
   FORWARD kernel at 256 registers = two wavefronts per SIMD (SA_FWD_CAP, defined with the mapping above): without the
   table build and the point history, with the saved Jacobian in LDS and the pow coefficients in constant memory it
   needs 254 and no scratch; the second wavefront buys latency hiding (Robertson
   B = 262 144: 25.3 -> 20.4 ms).  The backward kernel needs its 410+ registers; -DSA_FWD_OCC1 switches the cap off. */
#if SA_FWD_CAP
#ifndef SA_FWD_WAVES
#define SA_FWD_WAVES 2
#endif
#define SA_FWD_ATTR __attribute__((amdgpu_waves_per_eu(SA_FWD_WAVES, SA_FWD_WAVES)))
#else
#define SA_FWD_ATTR
#endif
extern "C" __global__ void __launch_bounds__(64) SA_FWD_ATTR sa_k_forward(sa_fwd_args a)
{
    const int inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= a.B) return;
    Cv<false> m;
    load_params(m, a.ps, a.pr, a.rem_stride, inst);
    m.rtol = a.rtol;
#ifdef SA_CONSTRAINTS
    m.constr = (a.constraints != nullptr);
    SFOR(i, 0, NS) m.cons[i] = m.constr ? a.constraints[i] : 0.0; SEND
#endif
    SFOR(i, 0, NS) m.atol[i] = a.atol[i]; SEND
    m.rtolQ = 0.0; m.atolQ = 0.0; m.tstop = 0.0;
    m.np = 0; m.tfinal = 0.0; m.ilast = 0; m.newdata = 0; m.have_last = 0;
    m.last_t = 0.0; m.n_interp = 0; m.n_rebuild = 0; m.tlo = 0.0; m.thi = 0.0;
    m.traj = nullptr; m.trow = 0; m.cur_idx = 0; m.tlo2 = 0.0; m.ltab = nullptr;
#if SA_SEARCH_CACHE
    m.thi2 = 0.0; m.thi2_idx = -1;
#endif
    m.pf[0] = m.pf[1] = m.pf[2] = m.pf[3] = 0.0;
#ifdef SA_SENS
    m.sensi = 0; m.ism = 0;
#endif

    double y0[NSD];
    SFOR(i, 0, NS) y0[i] = a.y0[(int64_t)inst * NS + i]; SEND
    { double q0_[NQD]; SFOR(i, 0, NQD) q0_[i] = 0.0; SEND cv_reinit(m, a.t0, y0, q0_); }

    /* store: CVodeF semantics (every step is a data point, no mxstep budget); wr: the points are written to the
       arena (SA_MODE_ADJ_COUNT runs the identical pass and only counts them, see sunode_amd.cpp) */
    const bool store = (a.mode != SA_MODE_PLAIN), wr = (a.mode == SA_MODE_ADJ_FWD);
    double *yo = a.y_out + (int64_t)inst * a.n_t * NS;
    double *trec = a.traj + (int64_t)inst * a.traj_istride * TREC;     /* records {order, dt, T[6], Y[6][n]} */
    const int64_t trow = a.traj_stride * TREC;
    double hT[QMAX + 1], hY[QMAX + 1][NSD];           /* the last six stored points, newest first */
    SFOR(j, 0, (QMAX) + 1) { hT[j] = 0.0; SFOR(i, 0, NS) hY[j][i] = 0.0; SEND } SEND

    int status = CV_SUCCESS, k = 0, np = 0, nstloc = 0, retries = 0, total_retries = 0, attempts = 0;
    while (k < a.n_t && a.tvals[k] == a.t0) {       /* solver.py:505,707 (row k; the reference writes row 0) */
        SFOR(i, 0, NS) yo[(int64_t)k * NS + i] = y0[i]; SEND
        k++;
    }
    bool done = (k >= a.n_t);
    StepCtl c;
    c.in_step = 0; c.redo = 0; c.nflag = FIRST_CALL; c.ncf = c.nef = c.nefQ = 0; c.convfail = 0; c.saved_t = a.t0;
    if (!done) {
        int flag = cv_first_call(m, a.tvals[k]);
        if (flag != CV_SUCCESS) { status = flag; done = true; }
        else if (store) {
            hT[0] = m.tn;
            SFOR(i, 0, NS) hY[0][i] = m.zn[0][i]; SEND
#ifdef SA_HERMITE
            if (wr) store_hermite(trec, m.tn, m.zn[0], m.f0, 1.0);
#elif SA_COMPACT
            if (wr) store_point(trec, 0, m.tn, m.zn[0]);
#else
            if (wr) store_table(trec, 0, 1.0, hT, hY);
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
                        SFOR_DOWN(j, QMAX, 1) {
                            hT[j] = hT[j - 1];
                            SFOR(i, 0, NS) hY[j][i] = hY[j - 1][i]; SEND
                        } SEND
                        hT[0] = m.tn;
                        SFOR(i, 0, NS) hY[0][i] = m.zn[0][i]; SEND
#ifdef SA_HERMITE
                        if (wr && np < a.traj_cap) store_hermite(trec + (int64_t)np * trow, m.tn, m.zn[0], m.zn[1], 1.0 / m.h);
#elif SA_COMPACT
                        if (wr && np < a.traj_cap) store_point(trec + (int64_t)np * trow, m.qu, m.tn, m.zn[0]);
#else
                        if (wr && np < a.traj_cap) store_table(trec + (int64_t)np * trow, m.qu, fabs(hT[0] - hT[1]), hT, hY);
#endif
                        np++;
                    }
                }
                while (!done && k < a.n_t) {
                    double tout = a.tvals[k];
                    if (tout == a.t0) {       /* (re-read: y0 is not worth NS register pairs across the whole loop) */
                        SFOR(i, 0, NS) yo[(int64_t)k * NS + i] = a.y0[(int64_t)inst * NS + i]; SEND
                        k++;
                    } else if ((m.tn - tout) * m.h >= 0.0) {
                        double dky[NSD], dq[NQD];
                        cv_get_dky0(m, tout, dky, dq);
                        SFOR(i, 0, NS) yo[(int64_t)k * NS + i] = dky[i]; SEND
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
#ifdef SA_TEST_PERTURB_FORWARD       /* tests/test_guard.py: a build that differs on purpose (never a default) */
    else yo[(int64_t)(a.n_t - 1) * NS] += 1.0;
#endif
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
#ifdef SA_TEST_PERTURB_NETF          /* tests/test_guard.py: a build that differs ONLY for instances with that many error-test
                                       failures -- a difference a prefix sample would never see (never a default) */
    if (status == CV_SUCCESS && st[ST_NETF] >= SA_TEST_PERTURB_NETF) yo[(int64_t)(a.n_t - 1) * NS] += 1.0;
#endif
    SFOR(i, 0, SA_N_STATS) a.stats[(int64_t)inst * SA_N_STATS + i] = st[i]; SEND
}

#ifdef SA_SENS
/* Solver(sens_mode=...).solve (reference solver.py:360-392, 497-531): the forward problem together with its
   NQ sensitivity systems, one instance per thread, everything in registers.  Same control flow as sa_k_forward
   without the trajectory; bit-identical to bdf_mem.hip's sa_k_sens (and to the oracle). */
extern "C" __global__ void __launch_bounds__(64) sa_k_sens(sa_sens_args a)
{
    const int inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= a.B) return;
    Cv<false> m;
    load_params(m, a.ps, a.pr, a.rem_stride, inst);
    m.rtol = a.rtol;
#ifdef SA_CONSTRAINTS
    m.constr = false;
    SFOR(i, 0, NS) m.cons[i] = 0.0; SEND
#endif
    SFOR(i, 0, NS) m.atol[i] = a.atol[i]; SEND
    SFOR(i, 0, NQ) m.pbar[i] = a.pbar[i]; SEND
    m.rtolQ = 0.0; m.atolQ = 0.0; m.tstop = 0.0;
    m.np = 0; m.tfinal = 0.0; m.ilast = 0; m.newdata = 0; m.have_last = 0;
    m.last_t = 0.0; m.n_interp = 0; m.n_rebuild = 0; m.tlo = 0.0; m.thi = 0.0;
    m.traj = nullptr; m.trow = 0; m.cur_idx = 0; m.tlo2 = 0.0; m.ltab = nullptr;
#if SA_SEARCH_CACHE
    m.thi2 = 0.0; m.thi2_idx = -1;
#endif
    m.pf[0] = m.pf[1] = m.pf[2] = m.pf[3] = 0.0;
    m.sensi = 1; m.ism = a.ism;

    double y0[NSD], s0[NQD][NSD];
    SFOR(i, 0, NS) y0[i] = a.y0[(int64_t)inst * NS + i]; SEND
    SFOR_S(is, i) s0[is][i] = a.sens0[((int64_t)inst * NQ + is) * NS + i]; SEND_S
    { double q0_[NQD]; SFOR(i, 0, NQD) q0_[i] = 0.0; SEND cv_reinit(m, a.t0, y0, q0_); }
    SFOR(v, 0, SV_COUNT) { SFOR_S(is, i) m.sv[v][is][i] = (v == SV_ZN0) ? s0[is][i] : 0.0; SEND_S } SEND

    double *yo = a.y_out + (int64_t)inst * a.n_t * NS;
    double *so = a.sens_out + (int64_t)inst * a.n_t * NQ * NS;
    int status = CV_SUCCESS, k = 0, nstloc = 0, retries = 0, total_retries = 0, attempts = 0;
    while (k < a.n_t && a.tvals[k] == a.t0) {
        SFOR(i, 0, NS) yo[(int64_t)k * NS + i] = y0[i]; SEND
        SFOR_S(is, i) so[((int64_t)k * NQ + is) * NS + i] = s0[is][i]; SEND_S
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
                        SFOR(i, 0, NS) yo[(int64_t)k * NS + i] = y0[i]; SEND
                        SFOR_S(is, i) so[((int64_t)k * NQ + is) * NS + i] = s0[is][i]; SEND_S
                        k++;
                    } else if ((m.tn - tout) * m.h >= 0.0) {
                        double dky[NSD], dq[NQD], dkyS[NQD][NSD];
                        cv_get_dky0(m, tout, dky, dq);
                        {   /* CVodeGetSensDky, k = 0, all parameters (t validated by cv_get_dky0) */
                            const double sx = (tout - m.tn) / m.h;
                            double pw[QMAX + 1];
                            pw[0] = 1.0;
                            SFOR(j, 1, (QMAX) + 1) pw[j] = pw[j - 1] * sx; SEND
                            SFOR_S(is, i) {
                                double acc = pw[QMAX] * m.sv[SV_ZN0 + QMAX][is][i];
                                SFOR_DOWN(j, QMAX - 1, 0) acc = FMA(pw[j], m.sv[SV_ZN0 + j][is][i], acc); SEND
                                dkyS[is][i] = acc;
                            } SEND_S
                        }
                        SFOR(i, 0, NS) yo[(int64_t)k * NS + i] = dky[i]; SEND
                        SFOR_S(is, i) so[((int64_t)k * NQ + is) * NS + i] = dkyS[is][i]; SEND_S
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

/* ------------------------------------------------------------------------------------ */
/* backward kernel: AdjointSolver.solve_backward (solver.py:723-784) over CVodeB semantics */
/* ------------------------------------------------------------------------------------ */
#ifdef SA_BWD_WAVES                  /* (experiment: cap the backward kernel's registers for SA_BWD_WAVES wavefronts per SIMD) */
#define SA_BWD_ATTR __attribute__((amdgpu_waves_per_eu(SA_BWD_WAVES, SA_BWD_WAVES)))
#else
#define SA_BWD_ATTR
#endif
extern "C" __global__ void __launch_bounds__(64) SA_BWD_ATTR sa_k_backward(sa_bwd_args a)
{
#if !SA_TAB_REGS
    __shared__ double ltab[TTAB * 64];        /* per-lane copy of the current divided-difference table */
#endif
    const int inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= a.B) return;
    int64_t st[SA_N_STATS];
    SFOR(i, 0, SA_N_STATS) st[i] = 0; SEND
    int status = CV_SUCCESS;
    const int np = a.traj_np[inst];
    if (a.fwd_status[inst] != CV_SUCCESS || np < 2) status = CV_NO_FWD;

    Cv<true> m;
    load_params(m, a.ps, a.pr, a.rem_stride, inst);
    m.rtol = a.rtolB;
    SFOR(i, 0, NS) m.atol[i] = a.atolB; SEND
    m.rtolQ = a.rtolQB; m.atolQ = a.atolQB;
    m.tstop = a.tinitial;
    m.traj = a.traj + (int64_t)inst * a.traj_istride * TREC;
    m.trow = a.traj_stride * TREC;
    m.np = np;
    m.tfinal = (status == CV_SUCCESS) ? m.traj[(int64_t)(np - 1) * m.trow + TREC_T] : a.tinitial;
    m.cur_idx = 0; m.tlo2 = 0.0;
#if SA_SEARCH_CACHE
    m.thi2 = 0.0; m.thi2_idx = -1;
#endif
    m.pf[0] = m.pf[1] = m.pf[2] = m.pf[3] = 0.0;
#ifdef SA_ABLATE_PROFILE
    SFOR(k, 0, 8) m.prof[k] = 0; SEND
    m.prof_last = __builtin_readcyclecounter(); m.prof_cur = 7;
#endif
#if SA_TAB_REGS
    m.ltab = nullptr;
#else
    m.ltab = ltab + threadIdx.x;
#endif
    SFOR(f, 0, TTAB) LT(m, f) = 0.0; SEND
    LT(m, 1) = 1.0;
    m.ilast = 0; m.newdata = 1; m.have_last = 0; m.last_t = 0.0;
    m.tlo = 0.0; m.thi = 0.0;
    m.n_interp = 0; m.n_rebuild = 0;
    SFOR(i, 0, NS) m.ytmp[i] = 0.0; SEND

    double lam[NSD], quad[NQD], quad_out[NQD];
    SFOR(i, 0, NS) lam[i] = 0.0; SEND
    SFOR(i, 0, NQ) { quad[i] = 0.0; quad_out[i] = 0.0; } SEND
    const double *g = a.grads + (int64_t)inst * a.grads_stride;
    bool first_call = true;
    int total_retries = 0, attempts = 0, wave_iters = 0;
    cv_reinit(m, a.t0, lam, quad);

    /* ts = [t0] + reversed(tvals) + [tend]; interval iv = (ts[iv+1], ts[iv]).  The wavefront walks the intervals
       together: every lane waits for the slowest at each observation, then all restart at once.
       (Measured and not kept -- profiles/r03_restart_batching.txt: the restart as a per-lane state of ONE attempt
       loop, executed when `batch` lanes wait for it or nobody is left stepping.  Bit-identical for every batch size,
       and slower for all of them: the flattened loop alone costs 10 % (batch = 64 = this schedule: Robertson backward
       68.1 -> 75.4 ms, LV 9.2 -> 10.1 ms); letting early lanes run ahead makes it worse (batch 32 / 16 / 8 / 4:
       100 / 99 / 96 / 93 ms) although 23 % of the Robertson kernel is lanes waiting at its six interval ends --
       what follows a restart (order-one steps, a matrix set-up with a fresh Jacobian in nearly every attempt while
       the step size grows) is cheap only while all 64 lanes go through it in the same iterations.) */
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
            int nstloc = 0, retries = 0, lane_iters = 0;
            StepCtl c;
            c.in_step = 0; c.redo = 0; c.nflag = FIRST_CALL; c.ncf = c.nef = c.nefQ = 0; c.convfail = 0;
            c.saved_t = t_upper;
            bool idone = (status != CV_SUCCESS);
            while (!idone) {
                PHASE(m, 0);
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
                    attempts++; lane_iters++;
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
            PHASE(m, 7);
            {   /* diagnostic: iterations the whole wave spent in this interval = max over its lanes */
                int wi = lane_iters;
                const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                SFOR(b, 0, 6) {        /* butterfly max over the 64 lanes (builtins only: no call ABI) */
                    int o = __builtin_amdgcn_ds_bpermute((lane ^ (1 << b)) << 2, wi);
                    wi = wi > o ? wi : o;
                } SEND
                wave_iters += wi;
            }
            if (status == CV_SUCCESS || m.nst > 0) accumulate_stats(m, st);
            if (status == CV_SUCCESS) { SFOR(i, 0, NQ) quad[i] = quad_out[i]; SEND }
        }
        if (iv < a.n_t && status == CV_SUCCESS) {
            const double *gi = g + (int64_t)(a.n_t - 1 - iv) * NS;
            SFOR(i, 0, NS) lam[i] -= gi[i]; SEND
            const int64_t row = (int64_t)inst * a.n_t + (iv == 0 ? 0 : a.n_t - iv);
            if (a.lamda_all) { SFOR(i, 0, NS) a.lamda_all[row * NS + i] = lam[i]; SEND }
            if (a.quad_all) { SFOR(i, 0, NQ) a.quad_all[row * NQ + i] = quad[i]; SEND }
        }
    }
    if (status != CV_SUCCESS) {
        if (a.lamda_all) for (int j = 0; j < a.n_t * NS; j++) a.lamda_all[(int64_t)inst * a.n_t * NS + j] = SA_NAN;
        if (a.quad_all) for (int j = 0; j < a.n_t * NQ; j++) a.quad_all[(int64_t)inst * a.n_t * NQ + j] = SA_NAN;
        SFOR(i, 0, NQ) quad_out[i] = SA_NAN; SEND
        SFOR(i, 0, NS) lam[i] = SA_NAN; SEND
    }
#ifdef SA_TEST_PERTURB_BACKWARD      /* tests/test_guard.py: a build whose adjoint differs on purpose (never a default) */
    lam[0] += 1.0;
#endif
    SFOR(i, 0, NQ) a.grad_out[(int64_t)inst * NQ + i] = quad_out[i]; SEND
    SFOR(i, 0, NS) a.lamda_out[(int64_t)inst * NS + i] = lam[i]; SEND
    a.status[inst] = status;
    st[ST_NPTS] = np; st[ST_NINTERP] = m.n_interp; st[ST_NREBUILD] = m.n_rebuild;
    st[ST_RETRIES] = total_retries; st[ST_ATTEMPTS] = attempts; st[ST_RESERVED1] = wave_iters;
#ifdef SA_ABLATE_PROFILE
    PHASE(m, 7);
    SFOR(k, 0, 8) st[8 + k] = m.prof[k]; SEND
#endif
    SFOR(i, 0, SA_N_STATS) a.stats[(int64_t)inst * SA_N_STATS + i] = st[i]; SEND
}

/* ------------------------------------------------------------------------------------ */
/* callback evaluation (EvalRhs op, codegen parity tests) and arithmetic probes            */
/* ------------------------------------------------------------------------------------ */
extern "C" __global__ void __launch_bounds__(64) sa_k_eval(sa_eval_args a)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.npts) return;
    double y[NSD], lam[NSD], ps[NQD], pr[SA_REM_IN_REGS ? NRD : 1];
    SFOR(k, 0, NS) { y[k] = a.y[(int64_t)i * NS + k]; lam[k] = a.lam[(int64_t)i * NS + k]; } SEND
    SFOR(k, 0, NQ) ps[k] = a.ps[(int64_t)i * NQ + k]; SEND
    const double *prp = pr;
    if constexpr (SA_REM_IN_REGS) { SFOR(k, 0, NR) pr[k] = a.pr[(int64_t)i * NR + k]; SEND }
    else { pr[0] = 0.0; prp = a.pr + (int64_t)i * NR; }
    const double t = a.t[i];
    double o1[NSD], oj[NSD * NSD], oq[NQD];
    int c0 = sa_rhs(t, y, ps, prp, o1);
    SFOR(k, 0, NS) a.rhs[(int64_t)i * NS + k] = o1[k]; SEND
    int c1 = sa_jac(t, y, ps, prp, oj);
    SFOR(k, 0, NS * NS) a.jac[(int64_t)i * NS * NS + k] = oj[k]; SEND
    int c2 = sa_adj_rhs(t, y, lam, ps, prp, o1);
    SFOR(k, 0, NS) a.adj[(int64_t)i * NS + k] = o1[k]; SEND
    int c3 = sa_quad_rhs(t, y, lam, ps, prp, oq);
    SFOR(k, 0, NQ) a.quad[(int64_t)i * NQ + k] = oq[k]; SEND
    int c4 = sa_adj_jac(t, y, ps, prp, oj);
    SFOR(k, 0, NS * NS) a.adjjac[(int64_t)i * NS * NS + k] = oj[k]; SEND
    a.codes[i * 5 + 0] = c0; a.codes[i * 5 + 1] = c1; a.codes[i * 5 + 2] = c2;
    a.codes[i * 5 + 3] = c3; a.codes[i * 5 + 4] = c4;
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

/* doubles per arena record when it is not the default 8 + 6n table record (read by sa_solver_create() if present) */
#if SA_COMPACT
extern "C" __device__ __attribute__((used)) const int32_t sa_traj_rec = TREC;
#endif
/* {n_states, n_sub, n_rem, ABI version, lanes per instance} read back by sa_solver_create() */
extern "C" __device__ __attribute__((used)) const int32_t sa_meta[6] = {NS, NQ, NR, SA_DEVICE_ABI_VERSION, 1, 0};
