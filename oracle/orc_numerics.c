/*
 * orc_numerics.c -- CPU ORACLE (test infrastructure, not the product):
 * the 17 log-densities, log-sum-exp helpers, the counter-based RNG stream and
 * the prior samplers.  See fugue_oracle.h for the pinning status.
 *
 * Every log-pdf keeps the reference's guard order and its left-to-right
 * floating-point evaluation order (compile with -ffp-contract=off).
 */
#define _GNU_SOURCE
#include "fugue_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const double LN_2PI = 1.8378770664093456;  /* distribution.rs:206 */
static const double LN_PI  = 1.1447298858494002;  /* distribution.rs:1372 */
#define NEG_INF (-INFINITY)

static inline double lgam(double x) { int sg; return lgamma_r(x, &sg); }

/* ---- 17 log-densities: src/core/distribution.rs ---- */

/* Normal: distribution.rs:189-208 */
static double lp_normal(double x, double mu, double sigma) {
    if (sigma <= 0.0 || !isfinite(sigma) || !isfinite(mu) || !isfinite(x)) return NEG_INF;
    double z = (x - mu) / sigma;
    return -0.5 * z * z - log(sigma) - 0.5 * LN_2PI;
}
/* Uniform (half-open): distribution.rs:309-330 */
static double lp_uniform(double x, double lo, double hi) {
    if (lo >= hi || !isfinite(lo) || !isfinite(hi) || !isfinite(x)) return NEG_INF;
    if (x < lo || x >= hi) return NEG_INF;
    double width = hi - lo;
    if (width <= 0.0) return NEG_INF;
    return -log(width);
}
/* LogNormal: distribution.rs:413-434 */
static double lp_lognormal(double x, double mu, double sigma) {
    if (sigma <= 0.0 || !isfinite(sigma) || !isfinite(mu)) return NEG_INF;
    if (x <= 0.0 || !isfinite(x)) return NEG_INF;
    double lx = log(x);
    double z = (lx - mu) / sigma;
    return -0.5 * z * z - lx - log(sigma) - 0.5 * LN_2PI;
}
/* Exponential: distribution.rs:503-518 */
static double lp_exponential(double x, double rate) {
    if (rate <= 0.0 || !isfinite(rate) || !isfinite(x)) return NEG_INF;
    if (x < 0.0) return NEG_INF;
    return log(rate) - rate * x;
}
/* Bernoulli: distribution.rs:598-619 */
static double lp_bernoulli(int x, double p) {
    if (p < 0.0 || p > 1.0 || !isfinite(p)) return NEG_INF;
    if (x) return (p <= 0.0) ? NEG_INF : log(p);
    return (p >= 1.0) ? NEG_INF : log(1.0 - p);
}
/* Categorical: distribution.rs:785-791 */
static double lp_categorical(int64_t x, const double *probs, int k) {
    if (x < 0 || x >= k) return NEG_INF;
    double p = probs[x];
    return (p > 0.0) ? log(p) : NEG_INF;
}
/* Beta: distribution.rs:897-956 */
static double lp_beta(double x, double a, double b) {
    if (a <= 0.0 || b <= 0.0 || !isfinite(a) || !isfinite(b) || !isfinite(x)) return NEG_INF;
    if (!(x >= 0.0 && x <= 1.0)) return NEG_INF;
    double log_beta_fn = lgam(a) + lgam(b) - lgam(a + b);
    if (x == 0.0) return (a > 1.0) ? NEG_INF : (a < 1.0) ? INFINITY : -log_beta_fn;
    if (x == 1.0) return (b > 1.0) ? NEG_INF : (b < 1.0) ? INFINITY : -log_beta_fn;
    double ln_x = log(x);
    double ln_1mx = log(1.0 - x);
    return (a - 1.0) * ln_x + (b - 1.0) * ln_1mx - log_beta_fn;
}
/* Gamma(shape, rate): distribution.rs:1039-1068 */
static double lp_gamma(double x, double shape, double rate) {
    if (shape <= 0.0 || rate <= 0.0 || !isfinite(shape) || !isfinite(rate) || !isfinite(x)) return NEG_INF;
    if (x <= 0.0) return NEG_INF;
    double log_rate = log(rate);
    double log_x = log(x);
    double lg = lgam(shape);
    return shape * log_rate + (shape - 1.0) * log_x - rate * x - lg;
}
/* Binomial(n, p): distribution.rs:1138-1165 */
static double lp_binomial(int64_t k, double nd, double p) {
    if (!isfinite(p) || !(p >= 0.0 && p <= 1.0)) return NEG_INF;
    if (k < 0) return NEG_INF;                 /* u64 domain */
    uint64_t n = (uint64_t)nd;
    if ((uint64_t)k > n) return NEG_INF;
    if (p == 0.0) return (k == 0) ? 0.0 : NEG_INF;
    if (p == 1.0) return ((uint64_t)k == n) ? 0.0 : NEG_INF;
    double lbc = lgam((double)n + 1.0) - lgam((double)k + 1.0) - lgam((double)(n - (uint64_t)k) + 1.0);
    return lbc + ((double)k) * log(p) + ((double)(n - (uint64_t)k)) * log(1.0 - p);
}
/* Poisson: distribution.rs:1237-1257 */
static double lp_poisson(int64_t k, double lambda) {
    if (lambda <= 0.0 || !isfinite(lambda)) return NEG_INF;
    if (k < 0) return NEG_INF;
    if (lambda > 700.0 && k == 0) return -lambda;
    double kf = (double)k;
    double log_lambda = log(lambda);
    double log_fact = lgam(kf + 1.0);
    return kf * log_lambda - lambda - log_fact;
}
/* StudentT(df, loc, scale): distribution.rs:1362-1381 */
static double lp_studentt(double x, double df, double loc, double scale) {
    if (df <= 0.0 || scale <= 0.0 || !isfinite(df) || !isfinite(scale) || !isfinite(loc) || !isfinite(x)) return NEG_INF;
    double z = (x - loc) / scale;
    return lgam((df + 1.0) / 2.0) - lgam(df / 2.0) - 0.5 * (log(df) + LN_PI) - log(scale)
           - 0.5 * (df + 1.0) * log1p(z * z / df);
}
/* Cauchy: distribution.rs:1451-1459 */
static double lp_cauchy(double x, double loc, double scale) {
    if (scale <= 0.0 || !isfinite(scale) || !isfinite(loc) || !isfinite(x)) return NEG_INF;
    double z = (x - loc) / scale;
    return -LN_PI - log(scale) - log1p(z * z);
}
/* Laplace: distribution.rs:1535-1541 */
static double lp_laplace(double x, double loc, double scale) {
    if (scale <= 0.0 || !isfinite(scale) || !isfinite(loc) || !isfinite(x)) return NEG_INF;
    return -log(2.0 * scale) - fabs(x - loc) / scale;
}
/* Weibull(shape, scale): distribution.rs:1618-1644 */
static double lp_weibull(double x, double shape, double scale) {
    if (shape <= 0.0 || scale <= 0.0 || !isfinite(shape) || !isfinite(scale) || !isfinite(x)) return NEG_INF;
    if (x < 0.0) return NEG_INF;
    if (x == 0.0) return (shape > 1.0) ? NEG_INF : (shape < 1.0) ? INFINITY : -log(scale);
    return log(shape) - shape * log(scale) + (shape - 1.0) * log(x) - pow(x / scale, shape);
}
/* ChiSquared(k): distribution.rs:1699-1709 */
static double lp_chisq(double x, double k) {
    if (k <= 0.0 || !isfinite(k) || !isfinite(x)) return NEG_INF;
    if (x <= 0.0) return NEG_INF;
    double hk = k / 2.0;
    return -hk * M_LN2 - lgam(hk) + (hk - 1.0) * log(x) - x / 2.0;
}
/* InverseGamma(shape, rate): distribution.rs:1789-1806 */
static double lp_invgamma(double x, double shape, double rate) {
    if (shape <= 0.0 || rate <= 0.0 || !isfinite(shape) || !isfinite(rate) || !isfinite(x)) return NEG_INF;
    if (x <= 0.0) return NEG_INF;
    return shape * log(rate) - lgam(shape) - (shape + 1.0) * log(x) - rate / x;
}
/* DiscreteUniform(lo, hi) inclusive: distribution.rs:1917-1932.
 * lo/hi arrive as doubles (the DSL casts `a[0] as i64`, dsl.rs:852-854); the
 * KAT harness passes exact i64 bounds through orc_logpdf_du below. */
static double lp_discrete_uniform_i(int64_t x, int64_t lo, int64_t hi) {
    if (hi < lo) return NEG_INF;
    if (x < lo || x > hi) return NEG_INF;
    if (lo == INT64_MIN && hi == INT64_MAX) return -(64.0 * M_LN2);
    unsigned __int128 cnt = (unsigned __int128)((__int128)hi - (__int128)lo) + 1;
    return -log((double)cnt);
}
static int64_t f2i_sat(double v) { /* Rust `as i64`: saturating, NaN -> 0 */
    if (isnan(v)) return 0;
    if (v >= 9223372036854775807.0) return INT64_MAX;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)v;
}

double orc_logpdf(int dist, int is_int, double xf, int64_t xi,
                  const double *p, int np) {
    (void)is_int;
    switch (dist) {
    case ORC_NORMAL:      return lp_normal(xf, p[0], p[1]);
    case ORC_UNIFORM:     return lp_uniform(xf, p[0], p[1]);
    case ORC_LOGNORMAL:   return lp_lognormal(xf, p[0], p[1]);
    case ORC_EXPONENTIAL: return lp_exponential(xf, p[0]);
    case ORC_BERNOULLI:   return lp_bernoulli(xi != 0, p[0]);
    case ORC_CATEGORICAL: return lp_categorical(xi, p, np);
    case ORC_BETA:        return lp_beta(xf, p[0], p[1]);
    case ORC_GAMMA:       return lp_gamma(xf, p[0], p[1]);
    case ORC_BINOMIAL:    return lp_binomial(xi, p[0], p[1]);
    case ORC_POISSON:     return lp_poisson(xi, p[0]);
    case ORC_STUDENTT:    return lp_studentt(xf, p[0], p[1], p[2]);
    case ORC_CAUCHY:      return lp_cauchy(xf, p[0], p[1]);
    case ORC_LAPLACE:     return lp_laplace(xf, p[0], p[1]);
    case ORC_WEIBULL:     return lp_weibull(xf, p[0], p[1]);
    case ORC_CHISQUARED:  return lp_chisq(xf, p[0]);
    case ORC_INVERSEGAMMA:return lp_invgamma(xf, p[0], p[1]);
    case ORC_DISCRETEUNIFORM:
        return lp_discrete_uniform_i(xi, f2i_sat(p[0]), f2i_sat(p[1]));
    default: return NAN;
    }
}
/* exact-i64-bounds entry for the full-range DiscreteUniform KATs
 * (distribution.rs:2525-2592) */
double orc_logpdf_du(int64_t x, int64_t lo, int64_t hi) {
    return lp_discrete_uniform_i(x, lo, hi);
}

/* ---- src/core/numerical.rs ---- */
/* log_sum_exp: numerical.rs:15-38 */
double orc_log_sum_exp(const double *x, size_t n) {
    if (n == 0) return NEG_INF;
    double mx = NEG_INF;
    for (size_t i = 0; i < n; i++) mx = fmax(mx, x[i]);   /* f64::max ignores NaN */
    if (isinf(mx) && mx < 0.0) return NEG_INF;
    double s = 0.0;
    for (size_t i = 0; i < n; i++) s += exp(x[i] - mx);
    if (s == 0.0) return NEG_INF;
    return mx + log(s);
}
/* normalize_log_probs: numerical.rs:87-90 */
void orc_normalize_log_probs(const double *x, size_t n, double *out) {
    double ls = orc_log_sum_exp(x, n);
    for (size_t i = 0; i < n; i++) out[i] = exp(x[i] - ls);
}
/* log1p_exp: numerical.rs:101-113 */
double orc_log1p_exp(double x) {
    if (x > 33.3) return x;
    if (x > -37.0) return log1p(exp(x));
    return exp(x);
}
/* safe_ln: numerical.rs:125-131 */
double orc_safe_ln(double x) {
    if (x <= 0.0 || !isfinite(x)) return NEG_INF;
    return log(x);
}

/* ---- Philox4x32-10 counter-based stream ---- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void orc_stream_init(orc_stream *s, uint64_t seed, uint32_t chain, uint32_t iter, uint32_t purpose) {
    s->key0 = (uint32_t)seed; s->key1 = (uint32_t)(seed >> 32);
    s->c0 = chain; s->c1 = 0; s->c2 = iter; s->c3 = purpose;
}
void orc_stream_block(orc_stream *s, uint64_t *a, uint64_t *b) {
    uint32_t ctr[4] = { s->c0, s->c1, s->c2, s->c3 }, key[2] = { s->key0, s->key1 }, o[4];
    orc_philox4x32_10(ctr, key, o);
    s->c1 += 1;
    *a = (uint64_t)o[0] | ((uint64_t)o[1] << 32);
    *b = (uint64_t)o[2] | ((uint64_t)o[3] << 32);
}
/* rand 0.8 `Standard` f64: 53 random bits scaled into [0,1) */
static inline double u01_of(uint64_t x) { return (double)(x >> 11) * 0x1.0p-53; }
double orc_stream_u01(orc_stream *s) { uint64_t a, b; orc_stream_block(s, &a, &b); return u01_of(a); }
/* Box-Muller with u1 in (0,1]; one Philox block per call */
void orc_stream_normal_pair(orc_stream *s, double *z0, double *z1) {
    uint64_t a, b; orc_stream_block(s, &a, &b);
    double u1 = ((double)(a >> 11) + 1.0) * 0x1.0p-53;
    double u2 = u01_of(b);
    double r = sqrt(-2.0 * log(u1));
    double th = 2.0 * M_PI * u2;
    *z0 = r * cos(th); *z1 = r * sin(th);
}
double orc_stream_normal(orc_stream *s) { double a, b; orc_stream_normal_pair(s, &a, &b); return a; }
/* gaussian_z: src/inference/mh.rs:128-132 (u1 clamped at 1e-10, cosine branch) */
double orc_stream_gaussian_z(orc_stream *s) {
    uint64_t a, b; orc_stream_block(s, &a, &b);
    double u1 = fmax(u01_of(a), 1e-10);
    double u2 = u01_of(b);
    return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}

/* ---- samplers (distribution.rs `sample` bodies; rand_distr algorithms are
 * unpinned, so any exact sampler is acceptable -- SURVEY 8c) ---- */
static double smp_gamma(double shape, double scale, orc_stream *s) {
    /* Marsaglia & Tsang (2000) */
    double boost = 1.0, k = shape;
    if (k < 1.0) { double u = orc_stream_u01(s); boost = pow(1.0 - u, 1.0 / k); k += 1.0; }
    double d = k - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (int it = 0; it < 1000; it++) {
        double x = orc_stream_normal(s);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        double u = 1.0 - orc_stream_u01(s);   /* (0,1] */
        double x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2) return d * v * scale * boost;
        if (log(u) < 0.5 * x2 + d * (1.0 - v + log(v))) return d * v * scale * boost;
    }
    return d * scale * boost;
}
static int64_t smp_poisson(double lambda, orc_stream *s) {
    if (lambda < 30.0) {           /* Knuth multiplication */
        double L = exp(-lambda), p = 1.0; int64_t k = 0;
        do { k++; p *= orc_stream_u01(s); } while (p > L && k < 100000);
        return k - 1;
    }
    /* PTRS, Hoermann (1993) */
    double slam = sqrt(lambda), loglam = log(lambda);
    double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
    double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
    for (int it = 0; it < 100000; it++) {
        uint64_t ra, rb; orc_stream_block(s, &ra, &rb);
        double U = u01_of(ra) - 0.5, V = 1.0 - u01_of(rb);
        double us = 0.5 - fabs(U);
        double kf = floor((2.0 * a / us + b) * U + lambda + 0.43);
        if (us >= 0.07 && V <= vr) return (int64_t)kf;
        if (kf < 0.0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -lambda + kf * loglam - lgam(kf + 1.0))
            return (int64_t)kf;
    }
    return (int64_t)lambda;
}
static int64_t smp_binomial(uint64_t n, double p, orc_stream *s) {
    if (p <= 0.0 || n == 0) return 0;
    if (p >= 1.0) return (int64_t)n;
    int flip = p > 0.5; double q = flip ? 1.0 - p : p; double nd = (double)n;
    int64_t k;
    if (nd * q < 10.0) {           /* BINV inversion */
        double sq = q / (1.0 - q), a = (nd + 1.0) * sq, r = pow(1.0 - q, nd);
        double u = orc_stream_u01(s); k = 0;
        while (u > r && k < (int64_t)n) { u -= r; k++; r *= (a / (double)k - sq); }
    } else {                       /* BTRS, Hoermann (1993), exact lgamma acceptance */
        double spq = sqrt(nd * q * (1.0 - q));
        double b = 1.15 + 2.53 * spq, a = -0.0873 + 0.0248 * b + 0.01 * q;
        double c = nd * q + 0.5, vr = 0.92 - 4.2 / b, alpha = (2.83 + 5.1 / b) * spq;
        double m = floor((nd + 1.0) * q), lpq = log(q / (1.0 - q));
        double hm = lgam(m + 1.0) + lgam(nd - m + 1.0);
        k = (int64_t)m;
        for (int it = 0; it < 100000; it++) {
            uint64_t ra, rb; orc_stream_block(s, &ra, &rb);
            double U = u01_of(ra) - 0.5, V = 1.0 - u01_of(rb);
            double us = 0.5 - fabs(U);
            double kf = floor((2.0 * a / us + b) * U + c);
            if (kf < 0.0 || kf > nd) continue;
            if (us >= 0.07 && V <= vr) { k = (int64_t)kf; break; }
            double lv = log(V * alpha / (a / (us * us) + b));
            if (lv <= hm - lgam(kf + 1.0) - lgam(nd - kf + 1.0) + (kf - m) * lpq) { k = (int64_t)kf; break; }
        }
    }
    return flip ? (int64_t)n - k : k;
}

orc_cell orc_sample_dist(int dist, const double *p, int np, orc_stream *s) {
    orc_cell r; r.i = 0;
    switch (dist) {
    case ORC_NORMAL:      /* distribution.rs:183-188 */
        r.f = (p[1] <= 0.0) ? NAN : p[0] + p[1] * orc_stream_normal(s); break;
    case ORC_UNIFORM:     /* :302-308 */
        r.f = (p[0] >= p[1] || !isfinite(p[0]) || !isfinite(p[1])) ? NAN
              : p[0] + (p[1] - p[0]) * orc_stream_u01(s); break;
    case ORC_LOGNORMAL:   /* :407-412 */
        r.f = (p[1] <= 0.0) ? NAN : exp(p[0] + p[1] * orc_stream_normal(s)); break;
    case ORC_EXPONENTIAL: /* :497-502 */
        r.f = (p[0] <= 0.0) ? NAN : -log(1.0 - orc_stream_u01(s)) / p[0]; break;
    case ORC_BERNOULLI:   /* :591-597 */
        r.i = (p[0] < 0.0 || p[0] > 1.0 || !isfinite(p[0])) ? 0 : (orc_stream_u01(s) < p[0]); break;
    case ORC_CATEGORICAL: { /* :771-784: first i with cumulative[i] >= u, clamped */
        double u = orc_stream_u01(s), cum = 0.0; int idx = np;
        for (int i = 0; i < np; i++) { cum += p[i]; if (!(cum < u)) { idx = i; break; } }
        r.i = idx < np - 1 ? idx : np - 1; if (np <= 0) r.i = 0; break; }
    case ORC_BETA: {      /* :891-896 */
        if (p[0] <= 0.0 || p[1] <= 0.0) { r.f = NAN; break; }
        double x = smp_gamma(p[0], 1.0, s), y = smp_gamma(p[1], 1.0, s);
        r.f = x / (x + y); break; }
    case ORC_GAMMA:       /* :1031-1038 (scale = 1/rate) */
        r.f = (p[0] <= 0.0 || p[1] <= 0.0) ? NAN : smp_gamma(p[0], 1.0 / p[1], s); break;
    case ORC_BINOMIAL:    /* :1135-1137 */
        r.i = smp_binomial((uint64_t)p[0], p[1], s); break;
    case ORC_POISSON:     /* :1231-1236 */
        r.i = (p[0] <= 0.0 || !isfinite(p[0])) ? 0 : smp_poisson(p[0], s); break;
    case ORC_STUDENTT: {  /* :1353-1361 */
        if (p[0] <= 0.0 || p[2] <= 0.0) { r.f = NAN; break; }
        double z = orc_stream_normal(s), c2 = smp_gamma(p[0] / 2.0, 2.0, s);
        r.f = p[1] + p[2] * (z / sqrt(c2 / p[0])); break; }
    case ORC_CAUCHY:      /* :1445-1450 */
        r.f = (p[1] <= 0.0) ? NAN : p[0] + p[1] * tan(M_PI * (orc_stream_u01(s) - 0.5)); break;
    case ORC_LAPLACE: {   /* :1524-1534 inverse CDF, as the reference */
        if (p[1] <= 0.0) { r.f = NAN; break; }
        double u = orc_stream_u01(s) - 0.5;
        double sg = (u > 0.0) ? 1.0 : (u < 0.0 ? -1.0 : 1.0);  /* f64::signum(+0.0)=1 */
        r.f = p[0] - p[1] * sg * log(1.0 - 2.0 * fabs(u)); break; }
    case ORC_WEIBULL:     /* :1611-1617 */
        r.f = (p[0] <= 0.0 || p[1] <= 0.0) ? NAN
              : p[1] * pow(-log(1.0 - orc_stream_u01(s)), 1.0 / p[0]); break;
    case ORC_CHISQUARED:  /* :1693-1698 */
        r.f = (p[0] <= 0.0) ? NAN : smp_gamma(p[0] / 2.0, 2.0, s); break;
    case ORC_INVERSEGAMMA:/* :1778-1788 */
        r.f = (p[0] <= 0.0 || p[1] <= 0.0) ? NAN : 1.0 / smp_gamma(p[0], 1.0 / p[1], s); break;
    case ORC_DISCRETEUNIFORM: { /* :1899-1916 */
        int64_t lo = f2i_sat(p[0]), hi = f2i_sat(p[1]);
        if (hi < lo) { r.i = lo; break; }
        uint64_t a, b; orc_stream_block(s, &a, &b);
        if (lo == INT64_MIN && hi == INT64_MAX) { r.i = (int64_t)a; break; }
        uint64_t cnt = (uint64_t)((__int128)hi - (__int128)lo) + 1;
        uint64_t off = (uint64_t)(((unsigned __int128)a * cnt) >> 64);
        r.i = (int64_t)((__int128)lo + (__int128)off); break; }
    default: r.f = NAN;
    }
    return r;
}
