/*
 * sa_math.h -- deterministic elementary functions for the generated callbacks.
 *
 * This text is EMBEDDED by sunode_amd/symode/codegen.py into the generated header of every problem whose
 * right-hand side uses a transcendental function (exp / log / sin / pow with a non-integer exponent / the
 * reference's helper functions logaddexp, expit, dexpit -- /root/reference/sunode/symode/lambdify.py:59-77),
 * so the device build (hipcc, gfx950) and the host build of the same header (the test oracle, gcc) execute ONE
 * sequence of IEEE-754 operations: +, -, *, /, floor, fabs, explicit fma(), integer bit operations.  With
 * -ffp-contract=off on both sides the results are bit-identical by construction, which keeps the step / order
 * bookkeeping of a transcendental model comparable bit for bit between the device and the oracle -- libm's and
 * ocml's exp / log / sin / pow differ in the last place, and one different ulp in a right-hand side forks an
 * adaptive step history.
 *
 * Accuracy (tests/test_sa_math.py, against mpmath): exp, log, sin, cos < 1 ulp; log1p, expm1, tanh, sinh, cosh,
 * tan <= 4 ulp; pow <= 2 ulp for results in [1e-150, 1e150], a few ulp more towards the overflow edge (the
 * logarithm is carried to ~2^-59).  The reference prints numpy calls compiled by numba with fastmath=True
 * (lambdify.py:88), i.e. it is itself only defined up to a few ulp.
 *
 * Domain: sin / cos / tan reduce with a three-word pi/2 and are exact-reduction-accurate for |x| <= 2^50; beyond
 * that (and for +-inf) they return NaN -- a non-finite output makes the callback report a recoverable error, the
 * same path a NaN from the reference's callbacks takes (symode/problem.py:266-269).
 *
 * Plain C99 + SA_FN (static inline / __device__ __forceinline__): no tables, no branches on the main paths other
 * than the special-value exits, so the lanes of a wavefront stay converged.
 */
#ifndef SA_MATH_H
#define SA_MATH_H
#define SA_HAVE_MATH 1

#define SAM_LN2_HI 0.6931471803691238      /* ln 2 with 21 trailing zero bits: k * SAM_LN2_HI is exact for |k| < 2^20 */
#define SAM_LN2_LO 1.9082149292705877e-10
#define SAM_INV_LN2 1.4426950408889634
#define SAM_INF (__builtin_huge_val())
#define SAM_NAN (__builtin_nan(""))

SA_FN double sam_from_bits(unsigned long long u) { double d; __builtin_memcpy(&d, &u, 8); return d; }
SA_FN unsigned long long sam_bits(double d) { unsigned long long u; __builtin_memcpy(&u, &d, 8); return u; }
/* 2^k for -1022 <= k <= 1023 */
SA_FN double sam_pow2(int k) { return sam_from_bits((unsigned long long)(k + 1023) << 52); }

/* exp(r) for |r| <= 0.35: Taylor polynomial of degree 13 (remainder 0.35^14 / 14! = 4e-18) */
SA_FN double sam_exp_poly(double r)
{
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    return p;                                   /* (exp(r) - 1 - r) / r^2 */
}

SA_FN double sa_exp(double x)
{
    if (!(x == x)) return x;
    if (x > 709.782712893384) return SAM_INF;
    if (x < -745.1332191019412) return 0.0;
    const double kf = floor(fma(x, SAM_INV_LN2, 0.5));
    const double r = fma(-kf, SAM_LN2_LO, fma(-kf, SAM_LN2_HI, x));
    const double p = fma(fma(sam_exp_poly(r), r, 1.0), r, 1.0);
    const int k = (int)kf, k1 = k / 2;
    return (p * sam_pow2(k1)) * sam_pow2(k - k1);    /* two exact-or-final scalings: subnormal results round once */
}

SA_FN double sa_expm1(double x)
{
    if (!(x == x)) return x;
    if (x > 40.0) return sa_exp(x);              /* exp(40) > 2^57: the -1 is below half an ulp */
    if (x < -40.0) return -1.0;
    const double kf = floor(fma(x, SAM_INV_LN2, 0.5));
    const double r = fma(-kf, SAM_LN2_LO, fma(-kf, SAM_LN2_HI, x));
    const double q = fma(r * r, sam_exp_poly(r), r);      /* expm1(r) */
    const double s = sam_pow2((int)kf);
    return (kf == 0.0) ? q : fma(s, q, s - 1.0);
}

/* sum_{j>=1} 2 z^j / (2j + 1), z = s^2 <= 0.0295: (log((1+s)/(1-s)) - 2s) / s */
SA_FN double sam_log_poly(double z)
{
    double p = 2.0 / 23.0;
    p = fma(p, z, 2.0 / 21.0);
    p = fma(p, z, 2.0 / 19.0);
    p = fma(p, z, 2.0 / 17.0);
    p = fma(p, z, 2.0 / 15.0);
    p = fma(p, z, 2.0 / 13.0);
    p = fma(p, z, 2.0 / 11.0);
    p = fma(p, z, 2.0 / 9.0);
    p = fma(p, z, 2.0 / 7.0);
    p = fma(p, z, 2.0 / 5.0);
    p = fma(p, z, 2.0 / 3.0);
    return p * z;
}

/* x > 0 finite -> exponent e and m in [sqrt(1/2), sqrt(2)) with x = m 2^e */
SA_FN double sam_split(double x, int *e_out)
{
    unsigned long long u = sam_bits(x);
    int e = (int)((u >> 52) & 0x7ff);
    if (e == 0) {                                /* subnormal: scale by 2^54 */
        u = sam_bits(x * 18014398509481984.0);
        e = (int)((u >> 52) & 0x7ff) - 54;
    }
    e -= 1023;
    double m = sam_from_bits((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    *e_out = e;
    return m;
}

SA_FN double sa_log(double x)
{
    if (!(x == x)) return x;
    if (x < 0.0) return SAM_NAN;
    if (x == 0.0) return -SAM_INF;
    if (x == SAM_INF) return x;
    int e;
    const double f = sam_split(x, &e) - 1.0;     /* exact */
    const double s = f / (2.0 + f);
    const double hfsq = 0.5 * f * f;
    /* log(1 + f) = 2 s + s R = f - (f^2/2 - s (f^2/2 + R)) */
    const double t = s * (hfsq + sam_log_poly(s * s));
    const double dk = (double)e;
    return fma(dk, SAM_LN2_HI, f - (hfsq - fma(dk, SAM_LN2_LO, t)));
}

SA_FN double sa_log1p(double x)
{
    if (!(x == x)) return x;
    if (x < -1.0) return SAM_NAN;
    if (x == -1.0) return -SAM_INF;
    if (x == SAM_INF) return x;
    if (fabs(x) < 5.551115123125783e-17) return x;
    const double u = 1.0 + x;
    const double c = (x >= 1.0) ? (1.0 - (u - x)) : (x - (u - 1.0));      /* what the rounding of u lost */
    return sa_log(u) + c / u;
}

/* ---- sin / cos ---- */
#define SAM_PIO2_1 1.5707963267948966
#define SAM_PIO2_2 6.123233995736766e-17
#define SAM_PIO2_3 -1.4973849048591698e-33

/* sin(r + rl), |r| <= pi/4 + eps: Taylor through r^17 */
SA_FN double sam_sin_k(double r, double rl)
{
    const double z = r * r;
    double p = -1.0 / 355687428096000.0;
    p = fma(p, z, 1.0 / 1307674368000.0);
    p = fma(p, z, -1.0 / 6227020800.0);
    p = fma(p, z, 1.0 / 39916800.0);
    p = fma(p, z, -1.0 / 362880.0);
    p = fma(p, z, 1.0 / 5040.0);
    p = fma(p, z, -1.0 / 120.0);
    p = fma(p, z, 1.0 / 6.0);
    return r - fma(r * z, p, -rl * fma(z, -0.5, 1.0));
}

/* cos(r + rl): Taylor through r^18 */
SA_FN double sam_cos_k(double r, double rl)
{
    const double z = r * r;
    double p = -1.0 / 6402373705728000.0;
    p = fma(p, z, 1.0 / 20922789888000.0);
    p = fma(p, z, -1.0 / 87178291200.0);
    p = fma(p, z, 1.0 / 479001600.0);
    p = fma(p, z, -1.0 / 3628800.0);
    p = fma(p, z, 1.0 / 40320.0);
    p = fma(p, z, -1.0 / 720.0);
    p = fma(p, z, 1.0 / 24.0);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + fma(z * z, p, -r * rl));
}

/* x = k pi/2 + (r + rl), |r + rl| <= pi/4 (+ rounding of the quotient); returns k mod 4.  |x| <= 2^50 */
SA_FN int sam_rem_pio2(double x, double *r_out, double *rl_out)
{
    const double kf = floor(fma(x, 0.6366197723675814, 0.5));
    const double t = fma(-kf, SAM_PIO2_1, x);    /* exact: a multiple of 2^-52 (2^-53 below 1) smaller than 1 */
    const double ph = kf * SAM_PIO2_2;
    const double pl = fma(kf, SAM_PIO2_2, -ph);  /* exact product = ph + pl */
    const double r0 = t - ph;
    const double bb = r0 - t;
    const double e0 = (t - (r0 - bb)) + (-ph - bb);      /* two-sum: t - ph = r0 + e0 */
    const double lo = fma(-kf, SAM_PIO2_3, e0 - pl);
    const double r = r0 + lo;
    *r_out = r;
    *rl_out = (r0 - r) + lo;
    return (int)((long long)kf & 3);
}

SA_FN double sa_sin(double x)
{
    if (!(fabs(x) <= 1125899906842624.0)) return SAM_NAN;
    double r, rl;
    const int q = sam_rem_pio2(x, &r, &rl);
    const double s = sam_sin_k(r, rl), c = sam_cos_k(r, rl);
    const double v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}

SA_FN double sa_cos(double x)
{
    if (!(fabs(x) <= 1125899906842624.0)) return SAM_NAN;
    double r, rl;
    const int q = sam_rem_pio2(x, &r, &rl);
    const double s = sam_sin_k(r, rl), c = sam_cos_k(r, rl);
    const double v = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -v : v;
}

SA_FN double sa_tan(double x)
{
    if (!(fabs(x) <= 1125899906842624.0)) return SAM_NAN;
    double r, rl;
    const int q = sam_rem_pio2(x, &r, &rl);
    const double s = sam_sin_k(r, rl), c = sam_cos_k(r, rl);
    return (q & 1) ? -c / s : s / c;
}

/* ---- hyperbolic ---- */
SA_FN double sa_tanh(double x)
{
    if (!(x == x)) return x;
    const double a = fabs(x);
    if (a > 20.0) return (x > 0.0) ? 1.0 : -1.0;
    const double t = sa_expm1(2.0 * a);
    const double v = t / (t + 2.0);
    return (x < 0.0) ? -v : ((x == 0.0) ? x : v);
}

SA_FN double sa_sinh(double x)
{
    if (!(x == x)) return x;
    const double a = fabs(x);
    double v;
    if (a > 709.0) { const double h = sa_exp(0.5 * a); v = (0.5 * h) * h; }
    else if (a < 3.725290298461914e-09) v = a;
    else { const double t = sa_expm1(a); v = 0.5 * (t + t / (t + 1.0)); }
    return (x < 0.0) ? -v : ((x == 0.0) ? x : v);
}

SA_FN double sa_cosh(double x)
{
    if (!(x == x)) return x;
    const double a = fabs(x);
    if (a > 709.0) { const double h = sa_exp(0.5 * a); return (0.5 * h) * h; }
    const double e = sa_exp(a);
    return 0.5 * e + 0.5 / e;
}

/* ---- pow ---- */
SA_FN double sa_pow(double x, double y)
{
    if (y == 0.0 || x == 1.0) return 1.0;
    if (!(x == x) || !(y == y)) return x + y;
    const double ax = fabs(x);
    double sign = 1.0;
    if (x < 0.0) {
        if (floor(y) != y) return SAM_NAN;                       /* negative base, non-integer exponent */
        const double half = 0.5 * y;
        if (fabs(y) < 9007199254740992.0 && floor(half) != half) sign = -1.0;
    }
    if (fabs(y) == SAM_INF) return ((ax > 1.0) == (y > 0.0)) ? SAM_INF : ((ax == 1.0) ? 1.0 : 0.0);
    if (ax == 0.0) return (y > 0.0) ? sign * 0.0 : sign * SAM_INF;
    if (ax == SAM_INF) return (y > 0.0) ? sign * SAM_INF : sign * 0.0;
    /* log(ax) = h + l to about 2^-59 relative: the quotient s = f / (2 + f) as a double-double, the odd series in s,
       e ln 2 in two words */
    int e;
    const double f = sam_split(ax, &e) - 1.0;
    const double d = 2.0 + f;
    const double dl = f - (d - 2.0);
    const double sh = f / d;
    const double sl = fma(-sh, dl, fma(-sh, d, f)) / d;
    const double T = 0.5 * sam_log_poly(sh * sh);                /* log(m) = 2 s (1 + T) */
    const double dk = (double)e;
    const double A = dk * SAM_LN2_HI, Bq = 2.0 * sh;
    const double h0 = A + Bq;
    const double bb = h0 - A;
    const double er = (A - (h0 - bb)) + (Bq - bb);               /* two-sum */
    const double l0 = er + fma(dk, SAM_LN2_LO, fma(Bq, T, 2.0 * sl));
    const double h = h0 + l0;
    const double l = (h0 - h) + l0;
    const double ph = y * h;
    const double pl = fma(y, h, -ph) + y * l;
    if (ph > 710.0) return sign * SAM_INF;
    if (ph < -746.0) return sign * 0.0;
    const double ev = sa_exp(ph);
    if (ev == SAM_INF || ev == 0.0) return sign * ev;
    return sign * fma(ev, pl, ev);
}

/* ---- the reference's helper functions (lambdify.py:59-77), on the functions above ---- */
SA_FN double sa_logaddexp(double a, double b)
{
    const double lo = fmin(a, b), hi = fmax(a, b);
    return hi + sa_log1p(sa_exp(lo - hi));
}
SA_FN double sa_expit(double x) { return 1.0 / (1.0 + sa_exp(-x)); }
SA_FN double sa_dexpit(double x) { return sa_expit(x) * sa_expit(-x); }
/* CardinalBSpline(4, t): the five polynomial pieces of the reference's helper, evaluated in its operation order */
SA_FN double sa_cardinal_bspline4(double t)
{
    if (t >= 0.0 && t <= 1.0) return (1.0 / 24.0) * (t * t * t * t);
    if (t >= 1.0 && t <= 2.0) return t * (t * (t * (5.0 / 6.0 - 1.0 / 6.0 * t) - 5.0 / 4.0) + 5.0 / 6.0) - 5.0 / 24.0;
    if (t >= 2.0 && t <= 3.0) return t * (t * (t * ((1.0 / 4.0) * t - 5.0 / 2.0) + 35.0 / 4.0) - 25.0 / 2.0) + 155.0 / 24.0;
    if (t >= 3.0 && t <= 4.0) return t * (t * (t * (5.0 / 2.0 - 1.0 / 6.0 * t) - 55.0 / 4.0) + 65.0 / 2.0) - 655.0 / 24.0;
    if (t >= 4.0 && t <= 5.0) return t * (t * (t * ((1.0 / 24.0) * t - 5.0 / 6.0) + 25.0 / 4.0) - 125.0 / 6.0) + 625.0 / 24.0;
    return 0.0;
}
#endif /* SA_MATH_H */
