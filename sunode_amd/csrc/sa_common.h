/*
 * sa_common.h -- pieces shared by the thread-per-instance kernels (bdf_kernels.hip, bdf_mem.hip) and the
 * lane-group / workgroup kernels (bdf_wave.hip): sizes, CVODES constants and return codes, compile-time loops,
 * the deterministic pow, and the scalar BDF coefficient routine cvSet.
 * Included after the generated problem header (SA_N_STATES, SA_N_SUB, SA_N_REM).
 */
#ifndef SA_COMMON_H
#define SA_COMMON_H

#define NS SA_N_STATES
#define NQ SA_N_SUB
#define NR SA_N_REM
#define NSD (NS > 0 ? NS : 1)
#define NQD (NQ > 0 ? NQ : 1)
#define NRD (NR > 0 ? NR : 1)
#define DEV static __device__ __forceinline__
/* explicit fused multiply-add at the same places as the CPU oracle (no compiler contraction) */
#define FMA(a, b, c) __builtin_fma((a), (b), (c))

/*
 * Compile-time loops.  Every per-lane array (Nordsieck columns, LU, coefficient vectors...) must
 * be scalar-replaced into VGPRs by the FIRST SROA pass, i.e. before instcombine gets a chance to
 * turn a select chain over array elements into a dynamically indexed scratch load.  `#pragma
 * unroll` unrolls too late for that, so loops over array indices are expanded by template
 * recursion (always-inlined lambdas): no loop and no variable subscript ever reaches the IR.
 */
template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, class F>
DEV void sfor(F &&f)
{
    if constexpr (B < E) { f(IC<B>{}); sfor<B + 1, E>(f); }
}
template <int B, int E, class F>
DEV void sfor_down(F &&f)          /* B, B-1, ..., E (inclusive) */
{
    if constexpr (B >= E) { f(IC<B>{}); sfor_down<B - 1, E>(f); }
}
#define SFOR(var, B, E) sfor<(B), (E)>([&](auto var##_ic) __attribute__((always_inline)) { constexpr int var = decltype(var##_ic)::value;
#define SFOR_DOWN(var, B, E) sfor_down<(B), (E)>([&](auto var##_ic) __attribute__((always_inline)) { constexpr int var = decltype(var##_ic)::value;
#define SEND });

/* loops over the slots of a state / quadrature vector in bdf_core.h: compile-time unless the mapping says otherwise
   (bdf_mem.hip: run-time loops over n components in HBM) */
#ifndef VFOR
#define VFOR(r) SFOR(r, 0, RS)
#define VEND SEND
#define QFOR(r) SFOR(r, 0, RQ)
#define QEND SEND
#endif

/* CVODES return codes (16_cvodes.h:45-106) */
#define CV_SUCCESS 0
#define CV_TSTOP_RETURN 1
#define CV_TOO_MUCH_WORK (-1)
#define CV_TOO_MUCH_ACC (-2)
#define CV_ERR_FAILURE (-3)
#define CV_CONV_FAILURE (-4)
#define CV_LSETUP_FAIL (-6)
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
#define CV_UNREC_QRHSFUNC_ERR (-34)
#define CV_SRHSFUNC_FAIL (-41)
#define CV_FIRST_SRHSFUNC_ERR (-42)
#define CV_REPTD_SRHSFUNC_ERR (-43)
#define CV_UNREC_SRHSFUNC_ERR (-44)
#define CV_NO_FWD (-102)
#define CV_BAD_TB0 (-104)
#define CV_GETY_BADT (-107)

/* CVODES constants */
#define QMAX 5
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

#define FIRST_CALL 101
#define PREV_CONV_FAIL 102
#define PREV_ERR_FAIL 103
#define RHSFUNC_RECVR 9
#define QRHSFUNC_RECVR 11
#define SRHSFUNC_RECVR 12
#define CONSTR_RECVR 10
#define CV_CONSTR_FAIL (-15)
#define NLS_CONV_RECVR 902
#define CV_NO_FAILURES 0
#define CV_FAIL_BAD_J 1
#define CV_FAIL_OTHER 2

/* forward-sensitivity vectors of an instance (one NQ x n block each): the six Nordsieck columns, the saved correction
   and the work vectors of the corrector; SV(m, vector, parameter, slot) in the kernels (bdf_core.h) */
enum { SV_ZN0 = 0, SV_ZSAVE = 6, SV_EWT = 7, SV_ACOR = 8, SV_TEMPV = 9, SV_FTEMP = 10, SV_Y = 11, SV_DELTA = 12, SV_COUNT = 13 };

enum { ST_NST, ST_NFE, ST_NSETUPS, ST_NJE, ST_NNI, ST_NCFN, ST_NETF, ST_QLAST,
       ST_NPTS, ST_NFQE, ST_NETFQ, ST_NINTERP, ST_NREBUILD, ST_RETRIES, ST_ATTEMPTS, ST_RESERVED1 };

/* Is `c` true for ANY active lane of the wavefront?  One VALU compare + one scalar test.  The controller uses it for
   wave-uniform shortcuts around straight-line blocks whose results only SOME lanes keep (order decision, tq[1] / tq[3]):
   when no lane needs them the whole block is jumped over by a scalar branch, when one does every lane runs the same
   straight-line code as before -- values identical either way (round 5: a lone wavefront pays ~4 cycles for every
   instruction it issues, needed by its lanes or not). */
DEV bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0; }
/* The forward-sensitivity builds keep the plain straight-line forms: measured with the shortcuts LV 25.1 -> 24.5 M,
   Robertson 0.91 -> 0.84 M, SEIR 43.9 -> 43.1 k sensitivity solves/s (their kernels live at the register limit:
   Robertson's sa_k_sens 1136 -> 1397 spill slots); adjoint / plain builds: LV +4 %, Robertson +2 %, network100 +3 %. */
#if defined(SA_SENS) && !defined(SA_SENS_SHORTCUT)
#define SA_SHORTCUT(c) true
#else
#define SA_SHORTCUT(c) wave_any(c)
#endif

/* a / b for operands far from the exponent limits (cvSet's step-size ratios and BDF coefficients, det_log's reduced argument: all O(1)): the
   instruction sequence of the compiler's IEEE division (v_rcp_f64, two Newton steps on the reciprocal, quotient,
   residual correction) without its range scaling -- v_div_scale leaves such operands unscaled, v_div_fmas then is a
   plain FMA and v_div_fixup passes the result through, so this IS the correctly rounded a / b, bit for bit
   (tests/test_gpu_parity.py::test_device_arithmetic checks it against numpy on the device). */
DEV double fdiv(double a, double b)
{
    const double r0 = __builtin_amdgcn_rcp(b);
    const double e0 = FMA(-b, r0, 1.0);
    const double r1 = FMA(r0, e0, r0);
    const double e1 = FMA(-b, r1, 1.0);
    const double r2 = FMA(r1, e1, r1);
    const double q0 = a * r2;
    const double rem = FMA(-b, q0, a);
    return FMA(rem, r2, q0);
}

/* the divisions of the interpolation table (CVApolynomialGetY: factor = dt / (t_j - t_{j-i}), 15 per table, and 1 / dt
   per evaluation): step sizes over sums of step sizes, far from every exponent limit, so the lean division above
   returns the IEEE quotient bit for bit (every parity test compares the result with the oracle's IEEE divisions).
   Measured (profiles/r06_table_fdiv.txt): SEIR +2.2 %, Robertson +0.5 %, LV +0.3 %.  -DSA_TABLE_IEEE_DIV: the plain
   division (the round-5 code). */
#ifdef SA_TABLE_IEEE_DIV
#define SA_TABLE_DIV(a, b) ((a) / (b))
#else
#define SA_TABLE_DIV(a, b) fdiv((a), (b))
#endif

/* ------------------------------------------------------------------------------------ */
/* deterministic pow (pure +,-,*,/): same operation sequence as the CPU restatement       */
/* ------------------------------------------------------------------------------------ */
/* Horner coefficients of det_log / det_exp.  As literals the compiler hoists all 26 out of the step loop into vector
   register pairs (~50 registers for the whole kernel); CM = true reads them from constant memory instead (uniform
   scalar loads feed the FMAs from scalar registers).  Measured (profiles/r03_compact_trajectory.txt, "coefficients"):
   pays where registers are the limit -- the forward kernel capped at 256 registers (Robertson: 46 spill slots and
   24 GB of scratch traffic per launch -> none, 22.7 -> 20.4 ms) and the lean lane groups (SEIR backward 360 -> 290
   spill slots, 58.5 -> 54.2 ms) -- and costs 1-2 % in the one-wavefront-per-SIMD backward kernels of the register
   mapping, which keep the literals.  Same values either way (the quotients are folded at compile time). */
#define SA_POLY_LOG {1.0 / 23.0, 1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, \
                     1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0, 1.0}
#define SA_POLY_EXP {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, \
                     1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0}
__constant__ double sa_poly_log[12] = SA_POLY_LOG;
__constant__ double sa_poly_exp[14] = SA_POLY_EXP;
template <bool CM, int I> DEV double poly_log_c() { constexpr double c[12] = SA_POLY_LOG; if constexpr (CM) return sa_poly_log[I]; else return c[I]; }
template <bool CM, int I> DEV double poly_exp_c() { constexpr double c[14] = SA_POLY_EXP; if constexpr (CM) return sa_poly_exp[I]; else return c[I]; }

template <bool CM = false>
DEV double det_log(double x)
{
    uint64_t u = __builtin_bit_cast(uint64_t, x);
    int e = (int)((u >> 52) & 0x7ff);
    if (e == 0) {
        u = __builtin_bit_cast(uint64_t, x * 18014398509481984.0);
        e = (int)((u >> 52) & 0x7ff) - 54;
    }
    e -= 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __builtin_bit_cast(double, u);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = fdiv(f, 2.0 + f);         /* |f| < 0.42: far from every exponent limit */
    double z = s * s;
    double p = poly_log_c<CM, 0>();
    SFOR(i, 1, 12) p = FMA(p, z, (poly_log_c<CM, i>())); SEND
    return FMA((double)e, 0.6931471805599453, 2.0 * s * p);
}

template <bool CM = false>
DEV double det_exp(double w)
{
    if (w > 700.0) w = 700.0;
    if (w < -700.0) w = -700.0;
    double kf = floor(FMA(w, 1.4426950408889634, 0.5));
    double r = FMA(-kf, 1.90821492927058770002e-10, FMA(-kf, 0.693147180369123816490, w));
    double p = poly_exp_c<CM, 0>();
    SFOR(i, 1, 14) p = FMA(p, r, (poly_exp_c<CM, i>())); SEND
    uint64_t bits = (uint64_t)((int64_t)kf + 1023) << 52;
    return p * __builtin_bit_cast(double, bits);
}

template <bool CM = false>
DEV double rpower_r(double base, double expo)
{
    if (base <= 0.0) return 0.0;
    return det_exp<CM>(expo * det_log<CM>(base));
}

/* the same value without the early return (select instead of branch): independent powers written one after the
   other stay in one basic block and the scheduler interleaves their dependent chains */
template <bool CM = false>
DEV double rpower_nb(double base, double expo)
{
    const bool nonpos = (base <= 0.0);
    const double r = det_exp<CM>(expo * det_log<CM>(nonpos ? 1.0 : base));
    return nonpos ? 0.0 : r;
}

/* N_VConstrMask entry: true where the inequality constraint c (0, +-1: >= / <= 0, +-2: > / < 0) on x fails */
DEV bool constr_violated(double c, double x)
{
    bool v = false;
    v = (c == 2.0) ? !(x > 0.0) : v;
    v = (c == 1.0) ? !(x >= 0.0) : v;
    v = (c == -1.0) ? !(x <= 0.0) : v;
    v = (c == -2.0) ? !(x < 0.0) : v;
    return v;
}

/* 1/k for k = 1..7, identical to the correctly rounded quotient 1.0/k */
DEV double inv_int(int k)
{
    double r = 1.0;
    r = (k == 2) ? 1.0 / 2.0 : r;
    r = (k == 3) ? 1.0 / 3.0 : r;
    r = (k == 4) ? 1.0 / 4.0 : r;
    r = (k == 5) ? 1.0 / 5.0 : r;
    r = (k == 6) ? 1.0 / 6.0 : r;
    r = (k == 7) ? 1.0 / 7.0 : r;
    return r;
}

/* dynamic pick from a small register array without dynamic indexing */
template <int N>
DEV double pick(const double (&a)[N], int idx)
{
    double r = a[0];
    SFOR(k, 1, N) r = (idx == k) ? a[k] : r; SEND
    return r;
}

/* cvSetBDF + cvSetTqBDF + the tail of cvSet */
template <class M>
DEV void cv_set(M &m)
{
    /* Straight-line (select-based) form of cvSetBDF/cvSetTqBDF: a lone wavefront per SIMD is bound by
       dependent-instruction latency, so one long basic block the scheduler can interleave beats a
       chain of short per-lane predicated blocks.  Values are identical to the branching form. */
    const int q = m.q;
    const bool gt1 = q > 1;
    double alpha0 = -1.0, alpha0_hat = -1.0, xi_inv = 1.0, xistar_inv = 1.0, hsum = m.h;
    m.l[0] = m.l[1] = 1.0;
    SFOR(i, 2, (QMAX) + 1) m.l[i] = 0.0; SEND
    SFOR(j, 2, QMAX) {
        const bool on = j < q;
        hsum = on ? hsum + m.tau[j - 1] : hsum;
        const double xi = fdiv(m.h, hsum);
        xi_inv = on ? xi : xi_inv;
        alpha0 = on ? alpha0 - 1.0 / j : alpha0;
        SFOR_DOWN(i, j, 1) { const double v = FMA(m.l[i - 1], xi_inv, m.l[i]); m.l[i] = on ? v : m.l[i]; } SEND
    } SEND
    {
        const double a0 = alpha0 - inv_int(q);
        alpha0 = gt1 ? a0 : alpha0;
        const double xs = -m.l[1] - alpha0;
        xistar_inv = gt1 ? xs : xistar_inv;
        const double hs = hsum + pick(m.tau, q - 1);
        hsum = gt1 ? hs : hsum;
        const double xi = fdiv(m.h, hsum);
        xi_inv = gt1 ? xi : xi_inv;
        const double ah = -m.l[1] - xi_inv;
        alpha0_hat = gt1 ? ah : alpha0_hat;
        SFOR_DOWN(i, QMAX, 1) {
            const double v = FMA(m.l[i - 1], xistar_inv, m.l[i]);
            m.l[i] = (gt1 && i <= q) ? v : m.l[i];
        } SEND
    }
    {
        const double lq = pick(m.l, q);
        const double A1 = 1.0 - alpha0_hat + alpha0;
        const double A2 = FMA((double)q, A1, 1.0);
        m.tq[2] = fabs(fdiv(A1, alpha0 * A2));
        m.tq[5] = fabs(fdiv(A2 * xistar_inv, lq * xi_inv));
        const bool w1 = (m.qwait == 1);
        if (SA_SHORTCUT(w1)) {       /* tq[1] / tq[3] feed the order decision of the NEXT step: four divisions nobody reads otherwise */
            const double C = fdiv(xistar_inv, lq);
            const double A3 = alpha0 + inv_int(q);
            const double A4 = alpha0_hat + xi_inv;
            const double Cpinv = fdiv(1.0 - A4 + A3, A3);
            const double tq1 = gt1 ? fabs(C * Cpinv) : 1.0;
            m.tq[1] = w1 ? tq1 : m.tq[1];
            const double hs = hsum + pick(m.tau, q);
            const double xi3 = fdiv(m.h, hs);
            const double A5 = alpha0 - inv_int(q + 1);
            const double A6 = alpha0_hat - xi3;
            const double Cppinv = fdiv(1.0 - A6 + A5, A2);
            const double tq3 = fabs(fdiv(Cppinv, xi3 * (q + 2) * A5));
            m.tq[3] = w1 ? tq3 : m.tq[3];
        }
        m.tq[4] = m.tq[2] * 10.0;       /* 1/tq[4] of CVODES (= tq[2]/nlscoef): the test multiplies */
    }
    m.rl1 = fdiv(1.0, m.l[1]);
    m.gamma = m.h * m.rl1;
    m.gammap = (m.nst == 0) ? m.gamma : m.gammap;
    const double gr = fdiv(m.gamma, m.gammap);
    m.gamrat = (m.nst > 0) ? gr : 1.0;
}


#endif
