// fg_cold.h -- the transcendental, once-per-transition pieces of the samplers as out-of-line device functions.
//
// Philox + Box-Muller (log, sqrt, sincos), the accept probability (exp) and dual averaging (sqrt, pow, exp) run once per
// transition; the leapfrog / scoring loops run thousands of f64 instructions between them.  Inlined into one kernel,
// ocml's argument reductions and table code take part in the hot loop's register allocation (round 1: 297 VGPR + 292 SGPR
// spills at the 128-VGPR budget of k_hmc_stream_steps).  Kept behind a call they get their own allocation; a call costs
// a few dozen cycles per transition.  Each translation unit that includes this header gets its own (static) copies.
#pragma once
#include "fg_math.h"

struct FgD2 { double a, b; };

// p0 pair j of a chain's (iteration) stream: fg_rng_normal_pair at block j (hmc.rs:436-441)
static __device__ __noinline__ FgD2 fg_cold_normal_pair(uint32_t k0, uint32_t k1, uint32_t chain, uint32_t block, uint32_t iter, uint32_t purpose) {
    FgStream s; s.k0 = k0; s.k1 = k1; s.c0 = chain; s.c1 = block; s.c2 = iter; s.c3 = purpose;
    FgD2 r;
    fg_rng_normal_pair(s, r.a, r.b);
    return r;
}
// two uniforms of one Philox block (a: the accept uniform of hmc.rs:461 / mh.rs:733)
static __device__ __noinline__ FgD2 fg_cold_u01_pair(uint32_t k0, uint32_t k1, uint32_t chain, uint32_t block, uint32_t iter, uint32_t purpose) {
    FgStream s; s.k0 = k0; s.k1 = k1; s.c0 = chain; s.c1 = block; s.c2 = iter; s.c3 = purpose;
    unsigned long long a, b;
    fg_rng_block(s, a, b);
    FgD2 r; r.a = fg_u01_of(a); r.b = fg_u01_of(b);
    return r;
}
// min(1, exp(h0 - h1)): hmc.rs:460
static __device__ __noinline__ double fg_cold_accept_prob(double h0, double h_new) { return fmin(exp(h0 - h_new), 1.0); }
static __device__ __noinline__ double fg_cold_exp(double x) { return exp(x); }
static __device__ __noinline__ double fg_cold_log(double x) { return log(x); }

// DualAveraging::update (hmc.rs:168-178) for the m-th update (m already incremented): returns {exp(log_eps), hbar, leb}
struct FgD3 { double a, b, c; };
static __device__ __noinline__ FgD3 fg_cold_da_update(double hbar, double leb, double m, double mu, double target, double alpha) {
    const double a = alpha < 0.0 ? 0.0 : (alpha > 1.0 ? 1.0 : alpha);
    const double frac = 1.0 / (m + 10.0);
    hbar = (1.0 - frac) * hbar + frac * (target - a);
    const double log_eps = mu - (sqrt(m) / 0.05) * hbar;
    const double w = pow(m, -0.75);
    leb = w * log_eps + (1.0 - w) * leb;
    FgD3 r; r.a = exp(log_eps); r.b = hbar; r.c = leb;
    return r;
}
// DiminishingAdaptation::update past the 10th proposal (mcmc_utils.rs:88-150): returns {scale, log_scale}
// step_tab[n] = 1 / n^0.7 for n < step_n, computed once per session on the host (the counts are small integers: one L2-resident
// load instead of ocml's ~150-instruction pow on the control wave's path); larger counts take pow
static __device__ __noinline__ FgD2 fg_cold_mh_adapt(double log_scale, uint32_t acc, uint32_t tot, const double *step_tab, uint32_t step_n) {
    const double rate = (double)acc / (double)tot;
    double step;
    if (tot < step_n) step = step_tab[tot];
    else step = 1.0 / pow((double)tot, 0.7);
    double ls = log_scale + step * (rate - 0.44);
    const double ns = exp(ls);
    const double sc = (fg_finite(ns) && ns > 0.0) ? fmin(fmax(ns, 0.001), 100.0) : 1.0;
    ls = (sc == 1.0) ? 0.0 : fg_fast_log(sc);            // sc in [1e-3, 100]
    FgD2 r; r.a = sc; r.b = ls;
    return r;
}
// ln of the 53-bit uniform of a Philox word (the accept test of the multi-wave MH kernels compares it with log_alpha); NaN for u = 0
static __device__ __noinline__ double fg_cold_lnu(unsigned long long r) { const double u = fg_u01_of(r); return u > 0.0 ? fg_fast_log(u) : NAN; }
// gaussian_z (mh.rs:128-132) from one Philox block
static __device__ __noinline__ double fg_cold_gaussian_z(unsigned long long a, unsigned long long b) { return fg_gaussian_z_of(a, b); }
