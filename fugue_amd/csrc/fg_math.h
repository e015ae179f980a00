// fg_math.h -- scalar building blocks of the engine, usable from host (program compiler:
// constant folding / hoisting) and device (gfx950 kernels): the 17 log-densities of
// /root/reference/src/core/distribution.rs with their guard order and left-to-right
// evaluation order, the Philox4x32-10 counter RNG and the prior samplers.
//
// Build with -ffp-contract=off: the reference (Rust) never fuses a*b+c.
#pragma once
#include <math.h>
#include <stdint.h>
#include "fg_ir.h"

#ifndef FG_HD                     /* the run-time compiled kernels (fg_jit.cpp) define it themselves */
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FG_HD __host__ __device__ __forceinline__
#else
#define FG_HD inline
#endif
#endif

#define FG_LN_2PI 1.8378770664093456   /* distribution.rs:206 */
#define FG_LN_PI 1.1447298858494002    /* distribution.rs:1372 */
#define FG_LN_2 0.6931471805599453
#define FG_NEG_INF (-INFINITY)
#define FG_MIN_POSITIVE 2.2250738585072014e-308
#define FG_F64_MAX 1.7976931348623157e308
#define FG_I64_MIN (-9223372036854775807LL - 1LL)
#define FG_I64_MAX 9223372036854775807LL

FG_HD bool fg_finite(double x) { return isfinite(x); }
FG_HD double fg_lgamma(double x) { return lgamma(x); }
FG_HD long long fg_f2i_sat(double v) {   // Rust `as i64`
    if (v != v) return 0;
    if (v >= 9223372036854775807.0) return FG_I64_MAX;
    if (v <= -9223372036854775808.0) return FG_I64_MIN;
    return (long long)v;
}
FG_HD double fg_as_double(long long bits) { union { long long i; double d; } u; u.i = bits; return u.d; }
FG_HD long long fg_as_i64(double d) { union { long long i; double d; } u; u.d = d; return u.i; }
FG_HD double fg_clamp(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }  // NaN stays

// ---------------------------------------------------------------------------------------
// Hoisting: when every parameter of a site is a compile-time constant the host evaluates
// the parameter guards once and precomputes the parameter-only sub-terms, in the
// reference's own evaluation order, into h[0..4].  Returns false when the constant
// parameters are invalid (log-density identically -inf).
// ---------------------------------------------------------------------------------------
// a / b for a divisor b whose correctly rounded reciprocal y = RN(1/b) is known (computed once, by IEEE division, on
// the host or per launch).  Markstein's sequence: q0 = a y is within 1.5 ulp of a/b; r = a - q b is exact in an FMA;
// q1 = q0 + r0 y is a faithful quotient; repeating the step from a faithful quotient yields the CORRECTLY ROUNDED a/b
// (Markstein 1990, Thm 4: needs y = RN(1/b) and a significand of b that is not all ones) -- the same bits as the
// reference's `/`, in 5 full-rate instructions instead of the 13-instruction v_div_scale / v_rcp / v_div_fixup expansion.
// Valid while nothing over/underflows: fg_div_const_ok(b) bounds b, callers bound a or accept that a non-finite a gives
// NaN where IEEE gives +-inf (both are "non-finite" to every consumer on this path).  tests/cpp/test_div_const.cpp checks
// 2e7 random and adversarial (near-midpoint) quotients against `/` on the host build of this function.
FG_HD double fg_div_const(double a, double b, double y) {
    const double q0 = a * y;
    const double r0 = __builtin_fma(-q0, b, a);
    const double q1 = __builtin_fma(r0, y, q0);
    const double r1 = __builtin_fma(-q1, b, a);
    return __builtin_fma(r1, y, q1);
}
FG_HD bool fg_div_const_ok(double b) {
    unsigned long long u; __builtin_memcpy(&u, &b, 8);
    const unsigned e = (unsigned)((u >> 52) & 0x7ffu);
    return e > 1023u - 400u && e < 1023u + 400u && (u & 0xfffffffffffffull) != 0xfffffffffffffull;
}
FG_HD bool fg_hoist(uint32_t kind, double p0, double p1, double p2, double *h) {
    h[0] = h[1] = h[2] = h[3] = h[4] = 0.0;
    switch (kind) {
    case 12: /* Normal */    if (p1 <= 0.0 || !fg_finite(p1) || !fg_finite(p0)) return false; h[0] = log(p1); return true;
    case 15: /* Uniform */   if (p0 >= p1 || !fg_finite(p0) || !fg_finite(p1)) return false;
                             { double w = p1 - p0; if (w <= 0.0) return false; h[0] = -log(w); } return true;
    case 11: /* LogNormal */ if (p1 <= 0.0 || !fg_finite(p1) || !fg_finite(p0)) return false; h[0] = log(p1); return true;
    case 7:  /* Exponential*/if (p0 <= 0.0 || !fg_finite(p0)) return false; h[0] = log(p0); return true;
    case 0:  /* Bernoulli */ if (p0 < 0.0 || p0 > 1.0 || !fg_finite(p0)) return false;
                             h[0] = (p0 <= 0.0) ? FG_NEG_INF : log(p0); h[1] = (p0 >= 1.0) ? FG_NEG_INF : log(1.0 - p0); return true;
    case 1:  /* Beta */      if (p0 <= 0.0 || p1 <= 0.0 || !fg_finite(p0) || !fg_finite(p1)) return false;
                             h[0] = fg_lgamma(p0) + fg_lgamma(p1) - fg_lgamma(p0 + p1); return true;
    case 8:  /* Gamma */     if (p0 <= 0.0 || p1 <= 0.0 || !fg_finite(p0) || !fg_finite(p1)) return false;
                             h[0] = log(p1); h[1] = fg_lgamma(p0); return true;
    case 2:  /* Binomial */  if (!fg_finite(p1) || !(p1 >= 0.0 && p1 <= 1.0)) return false;
                             { double n = (double)(unsigned long long)p0; h[0] = fg_lgamma(n + 1.0);
                               h[1] = (p1 > 0.0) ? log(p1) : FG_NEG_INF; h[2] = (p1 < 1.0) ? log(1.0 - p1) : FG_NEG_INF; } return true;
    case 13: /* Poisson */   if (p0 <= 0.0 || !fg_finite(p0)) return false; h[0] = log(p0); return true;
    case 14: /* StudentT */  if (p0 <= 0.0 || p2 <= 0.0 || !fg_finite(p0) || !fg_finite(p2) || !fg_finite(p1)) return false;
                             h[0] = fg_lgamma((p0 + 1.0) / 2.0) - fg_lgamma(p0 / 2.0) - 0.5 * (log(p0) + FG_LN_PI) - log(p2); return true;
    case 4:  /* Cauchy */    if (p1 <= 0.0 || !fg_finite(p1) || !fg_finite(p0)) return false; h[0] = -FG_LN_PI - log(p1); return true;
    case 10: /* Laplace */   if (p1 <= 0.0 || !fg_finite(p1) || !fg_finite(p0)) return false; h[0] = -log(2.0 * p1); return true;
    case 16: /* Weibull */   if (p0 <= 0.0 || p1 <= 0.0 || !fg_finite(p0) || !fg_finite(p1)) return false;
                             h[1] = log(p1); h[0] = log(p0) - p0 * h[1]; return true;
    case 5:  /* ChiSquared */if (p0 <= 0.0 || !fg_finite(p0)) return false;
                             { double hk = p0 / 2.0; h[0] = -hk * FG_LN_2 - fg_lgamma(hk); } return true;
    case 9:  /* InvGamma */  if (p0 <= 0.0 || p1 <= 0.0 || !fg_finite(p0) || !fg_finite(p1)) return false;
                             h[0] = p0 * log(p1) - fg_lgamma(p0); return true;
    default: return true;    // Categorical / DiscreteUniform are handled by the compiler
    }
}

// scale = 2^k (|k| <= 500): x / scale == x * (1/scale) bit for bit, so the hot loop may multiply
FG_HD bool fg_pow2_scale(double s) { int e; return s > 0.0 && fg_finite(s) && frexp(s, &e) == 0.5 && e > -500 && e < 500; }
// index of the scale parameter of a location-scale family, or -1
FG_HD int fg_scale_param(uint32_t kind) { return (kind == 12u || kind == 11u || kind == 4u || kind == 10u) ? 1 : (kind == 14u ? 2 : -1); }

// DiscreteUniform count -> -ln(n): distribution.rs:1917-1932
FG_HD double fg_du_logp(long long lo, long long hi) {
    if (lo == FG_I64_MIN && hi == FG_I64_MAX) return -(64.0 * FG_LN_2);
    unsigned long long cnt = (unsigned long long)hi - (unsigned long long)lo + 1ull;   // < 2^64 here
    return -log((double)cnt);
}

// ---------------------------------------------------------------------------------------
// log-density of a non-Categorical distribution.  xf / xi: the value as f64 / i64
// (only the one matching the distribution's value type is read).  `hoisted` is
// wave-uniform on the device.
// ---------------------------------------------------------------------------------------
FG_HD double fg_logpdf(uint32_t kind, bool hoisted, bool pow2, double xf, long long xi, double p0, double p1, double p2,
                       const double *h, bool sh = false /* scale-only hoisting: sigma guards done on the host, h[0] valid */,
                       bool xh = false /* constant observed count: h[3] = its own term (FG_F_XHOIST) */) {
    switch (kind) {
    case 12: { /* Normal: distribution.rs:189-208 */
        if (!hoisted && ((!sh && (p1 <= 0.0 || !fg_finite(p1))) || !fg_finite(p0))) return FG_NEG_INF;
        if (!fg_finite(xf)) return FG_NEG_INF;
        double z = pow2 ? (xf - p0) * h[4] : (xf - p0) / p1;
        double ls = (hoisted || sh) ? h[0] : log(p1);
        return -0.5 * z * z - ls - 0.5 * FG_LN_2PI; }
    case 15: { /* Uniform: :309-330 */
        if (!hoisted && (p0 >= p1 || !fg_finite(p0) || !fg_finite(p1))) return FG_NEG_INF;
        if (!fg_finite(xf)) return FG_NEG_INF;
        if (xf < p0 || xf >= p1) return FG_NEG_INF;
        if (hoisted) return h[0];
        double w = p1 - p0;
        if (w <= 0.0) return FG_NEG_INF;
        return -log(w); }
    case 11: { /* LogNormal: :413-434 */
        if (!hoisted && ((!sh && (p1 <= 0.0 || !fg_finite(p1))) || !fg_finite(p0))) return FG_NEG_INF;
        if (xf <= 0.0 || !fg_finite(xf)) return FG_NEG_INF;
        double lx = log(xf);
        double z = pow2 ? (lx - p0) * h[4] : (lx - p0) / p1;
        double ls = (hoisted || sh) ? h[0] : log(p1);
        return -0.5 * z * z - lx - ls - 0.5 * FG_LN_2PI; }
    case 7: { /* Exponential: :503-518 */
        if (!hoisted && (p0 <= 0.0 || !fg_finite(p0))) return FG_NEG_INF;
        if (!fg_finite(xf)) return FG_NEG_INF;
        if (xf < 0.0) return FG_NEG_INF;
        double lr = hoisted ? h[0] : log(p0);
        return lr - p0 * xf; }
    case 0: { /* Bernoulli: :598-619 */
        if (hoisted) return xi ? h[0] : h[1];
        if (p0 < 0.0 || p0 > 1.0 || !fg_finite(p0)) return FG_NEG_INF;
        if (xi) return (p0 <= 0.0) ? FG_NEG_INF : log(p0);
        return (p0 >= 1.0) ? FG_NEG_INF : log(1.0 - p0); }
    case 1: { /* Beta: :897-956 */
        if (!hoisted && (p0 <= 0.0 || p1 <= 0.0 || !fg_finite(p0) || !fg_finite(p1))) return FG_NEG_INF;
        if (!fg_finite(xf)) return FG_NEG_INF;
        if (!(xf >= 0.0 && xf <= 1.0)) return FG_NEG_INF;
        double lb = hoisted ? h[0] : (fg_lgamma(p0) + fg_lgamma(p1) - fg_lgamma(p0 + p1));
        if (xf == 0.0) return (p0 > 1.0) ? FG_NEG_INF : ((p0 < 1.0) ? INFINITY : -lb);
        if (xf == 1.0) return (p1 > 1.0) ? FG_NEG_INF : ((p1 < 1.0) ? INFINITY : -lb);
        double ln_x = log(xf);
        double ln_1mx = log(1.0 - xf);
        return (p0 - 1.0) * ln_x + (p1 - 1.0) * ln_1mx - lb; }
    case 8: { /* Gamma(shape, rate): :1039-1068 */
        if (!hoisted && (p0 <= 0.0 || p1 <= 0.0 || !fg_finite(p0) || !fg_finite(p1))) return FG_NEG_INF;
        if (!fg_finite(xf)) return FG_NEG_INF;
        if (xf <= 0.0) return FG_NEG_INF;
        double log_rate = hoisted ? h[0] : log(p1);
        double log_x = log(xf);
        double lg = hoisted ? h[1] : fg_lgamma(p0);
        return p0 * log_rate + (p0 - 1.0) * log_x - p1 * xf - lg; }
    case 2: { /* Binomial(n, p): :1138-1165 */
        if (!hoisted && (!fg_finite(p1) || !(p1 >= 0.0 && p1 <= 1.0))) return FG_NEG_INF;
        if (xi < 0) return FG_NEG_INF;
        unsigned long long n = (unsigned long long)p0, k = (unsigned long long)xi;
        if (k > n) return FG_NEG_INF;
        if (p1 == 0.0) return (k == 0) ? 0.0 : FG_NEG_INF;
        if (p1 == 1.0) return (k == n) ? 0.0 : FG_NEG_INF;
        double lbc;
        if (xh) lbc = h[3];
        else { double lgn = hoisted ? h[0] : fg_lgamma((double)n + 1.0);
               lbc = lgn - fg_lgamma((double)k + 1.0) - fg_lgamma((double)(n - k) + 1.0); }
        double lp = hoisted ? h[1] : log(p1);
        double lq = hoisted ? h[2] : log(1.0 - p1);
        return lbc + ((double)k) * lp + ((double)(n - k)) * lq; }
    case 13: { /* Poisson: :1237-1257 */
        if (!hoisted && (p0 <= 0.0 || !fg_finite(p0))) return FG_NEG_INF;
        if (xi < 0) return FG_NEG_INF;
        if (p0 > 700.0 && xi == 0) return -p0;
        double kf = (double)xi;
        double ll = hoisted ? h[0] : log(p0);
        double lf = xh ? h[3] : fg_lgamma(kf + 1.0);
        return kf * ll - p0 - lf; }
    case 14: { /* StudentT(df, loc, scale): :1362-1381 */
        if (!hoisted && (p0 <= 0.0 || p2 <= 0.0 || !fg_finite(p0) || !fg_finite(p2) || !fg_finite(p1))) return FG_NEG_INF;
        if (!fg_finite(xf)) return FG_NEG_INF;
        double z = pow2 ? (xf - p1) * h[4] : (xf - p1) / p2;
        double pre = hoisted ? h[0]
                             : (fg_lgamma((p0 + 1.0) / 2.0) - fg_lgamma(p0 / 2.0) - 0.5 * (log(p0) + FG_LN_PI) - log(p2));
        return pre - 0.5 * (p0 + 1.0) * log1p(z * z / p0); }
    case 4: { /* Cauchy: :1451-1459 */
        if (!hoisted && ((!sh && (p1 <= 0.0 || !fg_finite(p1))) || !fg_finite(p0))) return FG_NEG_INF;
        if (!fg_finite(xf)) return FG_NEG_INF;
        double z = pow2 ? (xf - p0) * h[4] : (xf - p0) / p1;
        double pre = (hoisted || sh) ? h[0] : (-FG_LN_PI - log(p1));
        return pre - log1p(z * z); }
    case 10: { /* Laplace: :1535-1541 */
        if (!hoisted && ((!sh && (p1 <= 0.0 || !fg_finite(p1))) || !fg_finite(p0))) return FG_NEG_INF;
        if (!fg_finite(xf)) return FG_NEG_INF;
        double pre = (hoisted || sh) ? h[0] : -log(2.0 * p1);
        return pow2 ? pre - fabs(xf - p0) * h[4] : pre - fabs(xf - p0) / p1; }
    case 16: { /* Weibull(shape, scale): :1618-1644 */
        if (!hoisted && (p0 <= 0.0 || p1 <= 0.0 || !fg_finite(p0) || !fg_finite(p1))) return FG_NEG_INF;
        if (!fg_finite(xf)) return FG_NEG_INF;
        if (xf < 0.0) return FG_NEG_INF;
        if (xf == 0.0) return (p0 > 1.0) ? FG_NEG_INF : ((p0 < 1.0) ? INFINITY : -(hoisted ? h[1] : log(p1)));
        double pre = hoisted ? h[0] : (log(p0) - p0 * log(p1));
        return pre + (p0 - 1.0) * log(xf) - pow(xf / p1, p0); }
    case 5: { /* ChiSquared(k): :1699-1709 */
        if (!hoisted && (p0 <= 0.0 || !fg_finite(p0))) return FG_NEG_INF;
        if (!fg_finite(xf)) return FG_NEG_INF;
        if (xf <= 0.0) return FG_NEG_INF;
        double hk = p0 / 2.0;
        double pre = hoisted ? h[0] : (-hk * FG_LN_2 - fg_lgamma(hk));
        return pre + (hk - 1.0) * log(xf) - xf / 2.0; }
    case 9: { /* InverseGamma(shape, rate): :1789-1806 */
        if (!hoisted && (p0 <= 0.0 || p1 <= 0.0 || !fg_finite(p0) || !fg_finite(p1))) return FG_NEG_INF;
        if (!fg_finite(xf)) return FG_NEG_INF;
        if (xf <= 0.0) return FG_NEG_INF;
        double pre = hoisted ? h[0] : (p0 * log(p1) - fg_lgamma(p0));
        return pre - (p0 + 1.0) * log(xf) - p1 / xf; }
    case 6: { /* DiscreteUniform(lo, hi): :1917-1932; hoisted bounds are exact i64 bit patterns */
        long long lo = hoisted ? fg_as_i64(p0) : fg_f2i_sat(p0);
        long long hi = hoisted ? fg_as_i64(p1) : fg_f2i_sat(p1);
        if (hi < lo) return FG_NEG_INF;
        if (xi < lo || xi > hi) return FG_NEG_INF;
        return hoisted ? h[0] : fg_du_logp(lo, hi); }
    default: return NAN;
    }
}

// ---------------------------------------------------------------------------------------
// FG_GRAD_ANALYTIC for every distribution (north_star: "finite-difference (and where available analytic) gradients"; opt-in, never
// the default -- the reference has only the finite difference, hmc.rs:304-329): the directional derivative of fg_logpdf,
//     d lp = dlp/dx dx + dlp/dp0 dp0 + dlp/dp1 dp1 + dlp/dp2 dp2,
// in the parameterisations of distribution.rs (Gamma(shape, rate), InverseGamma(shape, scale), Weibull(shape, scale), StudentT(df, loc,
// scale), Exponential(rate)).  NaN where the density is not finite (outside the support, invalid parameters, an endpoint with infinite
// density): the force is then non-finite and the transition diverges, as with the finite difference of a -inf log-joint.  Discrete
// values have no dx.  fg_digamma: recurrence up to 6, then the asymptotic series (|error| < 1e-12 for x > 0).
// ---------------------------------------------------------------------------------------
FG_HD double fg_digamma(double x) {
    if (!(x > 0.0) || !fg_finite(x)) return NAN;
    double r = 0.0;
    while (x < 6.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    // 1/12, 1/120, 1/252, 1/240, 1/132, 691/32760, 1/12
    const double t = f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0 - f * (1.0 / 132.0 - f * (691.0 / 32760.0 - f / 12.0))))));
    return r + log(x) - 0.5 / x - t;
}
FG_HD double fg_dlogpdf(uint32_t kind, double xf, long long xi, double p0, double p1, double p2, double dx, double dp0, double dp1, double dp2) {
    const double hh[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    const double lp = fg_logpdf(kind, false, false, xf, xi, p0, p1, p2, hh);
    if (!fg_finite(lp)) return NAN;
    switch (kind) {
    case 12: { const double z = (xf - p0) / p1; return (-z / p1) * (dx - dp0) + ((z * z - 1.0) / p1) * dp1; }                 // Normal(mu, sigma)
    case 15: { const double w = p1 - p0; return dp0 / w - dp1 / w; }                                                          // Uniform(a, b): -ln(b - a)
    case 11: { const double lx = log(xf), z = (lx - p0) / p1;                                                                 // LogNormal(mu, sigma)
               return (-(1.0 + z / p1) / xf) * dx + (z / p1) * dp0 + ((z * z - 1.0) / p1) * dp1; }
    case 7: return -p0 * dx + (1.0 / p0 - xf) * dp0;                                                                          // Exponential(rate)
    case 0: return (xi ? 1.0 / p0 : -1.0 / (1.0 - p0)) * dp0;                                                                 // Bernoulli(p)
    case 1: { const double ps = fg_digamma(p0 + p1);                                                                          // Beta(a, b)
              return ((p0 - 1.0) / xf - (p1 - 1.0) / (1.0 - xf)) * dx + (log(xf) - fg_digamma(p0) + ps) * dp0 + (log(1.0 - xf) - fg_digamma(p1) + ps) * dp1; }
    case 8: return ((p0 - 1.0) / xf - p1) * dx + (log(p1) + log(xf) - fg_digamma(p0)) * dp0 + (p0 / p1 - xf) * dp1;           // Gamma(shape, rate)
    case 2: { const double n = (double)(unsigned long long)p0, k = (double)xi; return (k / p1 - (n - k) / (1.0 - p1)) * dp1; } // Binomial(n, p)
    case 13: return ((double)xi / p0 - 1.0) * dp0;                                                                            // Poisson(lambda)
    case 14: { const double z = (xf - p1) / p2, q = p0 + z * z;                                                               // StudentT(df, loc, scale)
               const double gx = -(p0 + 1.0) * z / (p2 * q);
               const double gdf = 0.5 * fg_digamma(0.5 * (p0 + 1.0)) - 0.5 * fg_digamma(0.5 * p0) - 0.5 / p0 - 0.5 * log1p(z * z / p0) + 0.5 * (p0 + 1.0) * z * z / (p0 * q);
               return gx * (dx - dp1) + gdf * dp0 + (((p0 + 1.0) * z * z / q - 1.0) / p2) * dp2; }
    case 4: { const double z = (xf - p0) / p1, q = 1.0 + z * z;                                                               // Cauchy(loc, scale)
              return (-2.0 * z / (p1 * q)) * (dx - dp0) + ((2.0 * z * z / q - 1.0) / p1) * dp1; }
    case 10: { const double dl = xf - p0, sg = dl > 0.0 ? 1.0 : (dl < 0.0 ? -1.0 : 0.0);                                      // Laplace(loc, scale)
               return (-sg / p1) * (dx - dp0) + ((fabs(dl) / p1 - 1.0) / p1) * dp1; }
    case 16: { const double r = xf / p1, lr = log(r), rk = pow(r, p0);                                                        // Weibull(shape k, scale lambda)
               return ((p0 - 1.0) / xf - p0 * rk / xf) * dx + (1.0 / p0 + lr - rk * lr) * dp0 + (p0 * (rk - 1.0) / p1) * dp1; }
    case 5: return ((0.5 * p0 - 1.0) / xf - 0.5) * dx + (0.5 * log(xf) - 0.5 * FG_LN_2 - 0.5 * fg_digamma(0.5 * p0)) * dp0;   // ChiSquared(k)
    case 9: return (-(p0 + 1.0) / xf + p1 / (xf * xf)) * dx + (log(p1) - fg_digamma(p0) - log(xf)) * dp0 + (p0 / p1 - 1.0 / xf) * dp1;   // InverseGamma(shape, scale)
    case 6: return 0.0;                                                                                                       // DiscreteUniform: constant in its support
    default: return NAN;
    }
}

// ---------------------------------------------------------------------------------------
// Philox4x32-10 stream: key = seed, counter = (chain, block, iteration, purpose).
// Each call to fg_rng_block yields two u64 and advances `block`.
// ---------------------------------------------------------------------------------------
struct FgStream { uint32_t k0, k1, c0, c1, c2, c3; };

FG_HD uint32_t fg_mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((unsigned long long)a * b) >> 32); }
FG_HD void fg_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t *o) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32 x 32 -> 64 product per multiplier (v_mad_u64_u32 on gfx950) instead of a mul_hi / mul_lo pair
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
FG_HD FgStream fg_stream(unsigned long long seed, uint32_t chain, uint32_t iter, uint32_t purpose) {
    FgStream s; s.k0 = (uint32_t)seed; s.k1 = (uint32_t)(seed >> 32); s.c0 = chain; s.c1 = 0; s.c2 = iter; s.c3 = purpose;
    return s;
}
FG_HD void fg_rng_block(FgStream &s, unsigned long long &a, unsigned long long &b) {
    uint32_t o[4];
    fg_philox4x32_10(s.c0, s.c1, s.c2, s.c3, s.k0, s.k1, o);
    s.c1 += 1;
    a = (unsigned long long)o[0] | ((unsigned long long)o[1] << 32);
    b = (unsigned long long)o[2] | ((unsigned long long)o[3] << 32);
}
// rand 0.8 `Standard` f64: 53 bits scaled into [0,1)
FG_HD double fg_u01_of(unsigned long long x) { return (double)(x >> 11) * 0x1.0p-53; }
FG_HD double fg_rng_u01(FgStream &s) { unsigned long long a, b; fg_rng_block(s, a, b); return fg_u01_of(a); }
// ln x and (sin, cos) of an angle in [0, 2 pi) for the normal generators below, which run once per coordinate pair per
// transition (HMC) and once per step (MH) and were a fifth of the instructions of those kernels through ocml's general-purpose
// log / sincos (table lookups, Payne-Hanek reduction, double-double arithmetic for < 1 ulp over the whole range).  These are
// the classic fdlibm kernels (e_log.c, k_sin.c, k_cos.c, the first Cody-Waite round of e_rem_pio2.c: error < 1 ulp each) for the
// inputs that occur here -- x a positive normal number, the angle below 2 pi -- in ~40 instructions apiece.  The generators'
// values are pinned by no reference test (rand_distr's ziggurat is not reproduced); the oracle computes the same
// expressions with glibc and the parity tests compare with tolerances far above an ulp.  tests/cpp/test_fast_math.cpp checks
// both against libm on the host build.
FG_HD double fg_fast_log(double x) {            // x > 0, normal
    const long long bx = fg_as_i64(x);
    int hx = (int)(bx >> 32);
    int k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int i = (hx + 0x95f64) & 0x100000;     // significand above sqrt(2): halve it
    k += i >> 20;
    const double m = fg_as_double(((long long)(hx | (i ^ 0x3ff00000)) << 32) | (bx & 0xffffffffLL));   // [sqrt(1/2), sqrt(2))
    const double f = m - 1.0, dk = (double)k, d = 2.0 + f;
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(d);
#else
    double y = (double)(float)(1.0 / d);         // a seed of the same quality as v_rcp_f64 (host build: tests only)
#endif
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    double sq = f * y;
    sq = __builtin_fma(__builtin_fma(-sq, d, f), y, sq);             // f / (2 + f)
    const double z = sq * sq, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                                        6.666666666666735130e-01);
    const double R = t2 + t1, hfsq = 0.5 * f * f;
    return dk * 6.93147180369123816490e-01 - ((hfsq - __builtin_fma(sq, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}
FG_HD void fg_fast_sincos(double x, double &sn, double &cs) {   // 0 <= x < 2 pi (any |x| < ~1e5 reduces accurately)
    const double fn = __builtin_rint(x * 6.36619772367581382433e-01);
    const double r = __builtin_fma(-fn, 1.57079632673412561417e+00, x);   // 33-bit head of pi / 2: the product is exact
    const double w = fn * 6.07710050650619224932e-11;
    const double y0 = r - w, y1 = (r - y0) - w;
    const double z = y0 * y0;
    // k_sin
    const double v = z * y0;
    const double rs = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                                     -1.98412698298579493134e-04), 8.33333333332248946124e-03);
    const double s0 = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * -1.66666666666666324348e-01);
    // k_cos
    const double w2 = z * z;
    const double rc = z * __builtin_fma(z, __builtin_fma(z, 2.48015872894767294178e-05, -1.38888888888741095749e-03), 4.16666666666666019037e-02) +
                      (w2 * w2) * __builtin_fma(z, __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07);
    const double hz = 0.5 * z, wc = 1.0 - hz;
    const double c0 = wc + (((1.0 - wc) - hz) + (z * rc - y0 * y1));
    const int n = (int)fn;
    const double sa = (n & 1) ? c0 : s0, ca = (n & 1) ? s0 : c0;
    sn = (n & 2) ? -sa : sa;
    cs = ((n + 1) & 2) ? -ca : ca;
}
FG_HD void fg_rng_normal_pair(FgStream &s, double &z0, double &z1) {   // Box-Muller, u1 in (0,1]
    unsigned long long a, b; fg_rng_block(s, a, b);
    double u1 = ((double)(a >> 11) + 1.0) * 0x1.0p-53;
    double u2 = fg_u01_of(b);
    double r = sqrt(-2.0 * fg_fast_log(u1));
    double th = 2.0 * M_PI * u2;
    double sn, cs;
    fg_fast_sincos(th, sn, cs);
    z0 = r * cs; z1 = r * sn;
}
FG_HD double fg_rng_normal(FgStream &s) { double a, b; fg_rng_normal_pair(s, a, b); return a; }
// gaussian_z: /root/reference/src/inference/mh.rs:128-132
FG_HD double fg_gaussian_z_of(unsigned long long a, unsigned long long b) {
    double u1 = fmax(fg_u01_of(a), 1e-10);
    double u2 = fg_u01_of(b);
    double sn, cs;
    fg_fast_sincos(2.0 * M_PI * u2, sn, cs);
    return sqrt(-2.0 * fg_fast_log(u1)) * cs;
}
FG_HD double fg_rng_gaussian_z(FgStream &s) { unsigned long long a, b; fg_rng_block(s, a, b); return fg_gaussian_z_of(a, b); }
// gen_range(0..n): widening multiply (bias < n * 2^-64)
FG_HD uint32_t fg_pick(unsigned long long r, uint32_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__umul64hi(r, (unsigned long long)n);
#else
    return (uint32_t)(((unsigned __int128)r * n) >> 64);
#endif
}

// ---------------------------------------------------------------------------------------
// Samplers (`Distribution::sample`, distribution.rs:183-188 ... 1899-1916).  rand_distr's
// algorithms are not pinned by any reference test; these are exact samplers that consume
// the Philox stream in a fixed order (the test oracle uses the identical order).
// ---------------------------------------------------------------------------------------
FG_HD double fg_smp_gamma(double shape, double scale, FgStream &s) {   // Marsaglia-Tsang
    double boost = 1.0, k = shape;
    if (k < 1.0) { double u = fg_rng_u01(s); boost = pow(1.0 - u, 1.0 / k); k += 1.0; }
    double d = k - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (int it = 0; it < 1000; it++) {
        double x = fg_rng_normal(s);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        double u = 1.0 - fg_rng_u01(s);
        double x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2) return d * v * scale * boost;
        if (log(u) < 0.5 * x2 + d * (1.0 - v + log(v))) return d * v * scale * boost;
    }
    return d * scale * boost;
}
FG_HD long long fg_smp_poisson(double lambda, FgStream &s) {
    if (lambda < 30.0) {
        double L = exp(-lambda), p = 1.0; long long k = 0;
        do { k++; p *= fg_rng_u01(s); } while (p > L && k < 100000);
        return k - 1;
    }
    double slam = sqrt(lambda), loglam = log(lambda);   // PTRS
    double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
    double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
    for (int it = 0; it < 100000; it++) {
        unsigned long long ra, rb; fg_rng_block(s, ra, rb);
        double U = fg_u01_of(ra) - 0.5, V = 1.0 - fg_u01_of(rb);
        double us = 0.5 - fabs(U);
        double kf = floor((2.0 * a / us + b) * U + lambda + 0.43);
        if (us >= 0.07 && V <= vr) return (long long)kf;
        if (kf < 0.0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -lambda + kf * loglam - fg_lgamma(kf + 1.0)) return (long long)kf;
    }
    return (long long)lambda;
}
FG_HD long long fg_smp_binomial(unsigned long long n, double p, FgStream &s) {
    if (p <= 0.0 || n == 0) return 0;
    if (p >= 1.0) return (long long)n;
    bool flip = p > 0.5; double q = flip ? 1.0 - p : p; double nd = (double)n;
    long long k;
    if (nd * q < 10.0) {   // BINV
        double sq = q / (1.0 - q), a = (nd + 1.0) * sq, r = pow(1.0 - q, nd);
        double u = fg_rng_u01(s); k = 0;
        while (u > r && k < (long long)n) { u -= r; k++; r *= (a / (double)k - sq); }
    } else {               // BTRS with exact lgamma acceptance
        double spq = sqrt(nd * q * (1.0 - q));
        double b = 1.15 + 2.53 * spq, a = -0.0873 + 0.0248 * b + 0.01 * q;
        double c = nd * q + 0.5, vr = 0.92 - 4.2 / b, alpha = (2.83 + 5.1 / b) * spq;
        double m = floor((nd + 1.0) * q), lpq = log(q / (1.0 - q));
        double hm = fg_lgamma(m + 1.0) + fg_lgamma(nd - m + 1.0);
        k = (long long)m;
        for (int it = 0; it < 100000; it++) {
            unsigned long long ra, rb; fg_rng_block(s, ra, rb);
            double U = fg_u01_of(ra) - 0.5, V = 1.0 - fg_u01_of(rb);
            double us = 0.5 - fabs(U);
            double kf = floor((2.0 * a / us + b) * U + c);
            if (kf < 0.0 || kf > nd) continue;
            if (us >= 0.07 && V <= vr) { k = (long long)kf; break; }
            double lv = log(V * alpha / (a / (us * us) + b));
            if (lv <= hm - fg_lgamma(kf + 1.0) - fg_lgamma(nd - kf + 1.0) + (kf - m) * lpq) { k = (long long)kf; break; }
        }
    }
    return flip ? (long long)n - k : k;
}
// Draw from a non-Categorical distribution; the result is a raw 8-byte cell
// (double bits for f64 distributions, the integer otherwise).
FG_HD long long fg_sample_dist(uint32_t kind, bool hoisted, double p0, double p1, double p2, FgStream &s) {
    switch (kind) {
    case 12: return fg_as_i64((p1 <= 0.0) ? NAN : p0 + p1 * fg_rng_normal(s));
    case 15: return fg_as_i64((p0 >= p1 || !fg_finite(p0) || !fg_finite(p1)) ? NAN : p0 + (p1 - p0) * fg_rng_u01(s));
    case 11: return fg_as_i64((p1 <= 0.0) ? NAN : exp(p0 + p1 * fg_rng_normal(s)));
    case 7:  return fg_as_i64((p0 <= 0.0) ? NAN : -log(1.0 - fg_rng_u01(s)) / p0);
    case 0:  return (p0 < 0.0 || p0 > 1.0 || !fg_finite(p0)) ? 0 : (long long)(fg_rng_u01(s) < p0);
    case 1: { if (p0 <= 0.0 || p1 <= 0.0) return fg_as_i64(NAN);
              double x = fg_smp_gamma(p0, 1.0, s), y = fg_smp_gamma(p1, 1.0, s); return fg_as_i64(x / (x + y)); }
    case 8:  return fg_as_i64((p0 <= 0.0 || p1 <= 0.0) ? NAN : fg_smp_gamma(p0, 1.0 / p1, s));
    case 2:  return fg_smp_binomial((unsigned long long)p0, p1, s);
    case 13: return (p0 <= 0.0 || !fg_finite(p0)) ? 0 : fg_smp_poisson(p0, s);
    case 14: { if (p0 <= 0.0 || p2 <= 0.0) return fg_as_i64(NAN);
               double z = fg_rng_normal(s), c2 = fg_smp_gamma(p0 / 2.0, 2.0, s);
               return fg_as_i64(p1 + p2 * (z / sqrt(c2 / p0))); }
    case 4:  return fg_as_i64((p1 <= 0.0) ? NAN : p0 + p1 * tan(M_PI * (fg_rng_u01(s) - 0.5)));
    case 10: { if (p1 <= 0.0) return fg_as_i64(NAN);
               double u = fg_rng_u01(s) - 0.5;
               double sg = (u < 0.0) ? -1.0 : 1.0;
               return fg_as_i64(p0 - p1 * sg * log(1.0 - 2.0 * fabs(u))); }
    case 16: return fg_as_i64((p0 <= 0.0 || p1 <= 0.0) ? NAN : p1 * pow(-log(1.0 - fg_rng_u01(s)), 1.0 / p0));
    case 5:  return fg_as_i64((p0 <= 0.0) ? NAN : fg_smp_gamma(p0 / 2.0, 2.0, s));
    case 9:  return fg_as_i64((p0 <= 0.0 || p1 <= 0.0) ? NAN : 1.0 / fg_smp_gamma(p0, 1.0 / p1, s));
    case 6: { long long lo = hoisted ? fg_as_i64(p0) : fg_f2i_sat(p0), hi = hoisted ? fg_as_i64(p1) : fg_f2i_sat(p1);
              if (hi < lo) return lo;
              unsigned long long a, b; fg_rng_block(s, a, b);
              if (lo == FG_I64_MIN && hi == FG_I64_MAX) return (long long)a;
              unsigned long long cnt = (unsigned long long)hi - (unsigned long long)lo + 1ull;
#if defined(__HIP_DEVICE_COMPILE__)
              unsigned long long off = __umul64hi(a, cnt);
#else
              unsigned long long off = (unsigned long long)(((unsigned __int128)a * cnt) >> 64);
#endif
              return (long long)((unsigned long long)lo + off); }
    default: return 0;
    }
}
