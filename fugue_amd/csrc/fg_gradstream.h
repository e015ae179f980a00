// fg_gradstream.h -- fused finite-difference gradient for programs whose force terms are all
// fast Normals (FG_OP_NORMAL_FAST): ONE software-pipelined pass over a pre-built stream of
// 64-byte records computes, for every f64 coordinate i, the reference's central difference
//     g_i = (lp(q + h e_i) - lp(q - h e_i)) / (2h)                       (hmc.rs:315-327)
// over exactly the statements that read q_i, evaluating both signs of a record together
// (the same additions/multiplications, in the same order, as two sparse evaluations), and applies
// the leapfrog half-kick(s) to p_i as soon as g_i is known (hmc.rs:389,400).
//
// Pipeline (per record k):   wait -> [ fetch record k+2 | LDS-read operands of k+1 ] -> math(k)
// so the scalar-load and LDS latencies of the next records hide behind the ~30 f64 VALU
// instructions of the current one.  s_waitcnt is placed by hand: SMEM returns out of order, so
// the compiler would otherwise wait for the just-issued prefetch before the first LDS use.
#pragma once
#include "fg_interp.h"

__device__ __forceinline__ fg_u32x16 fg_fetch_grec(const FgGradRec *g, int k) {
    return *(const FG_AS4 fg_u32x16 *)(uintptr_t)(g + k);
}

__device__ __forceinline__ double fg_uniform(double v) {
    const long long b = __double_as_longlong(v);
    return fg_dbl(__builtin_amdgcn_readfirstlane((uint32_t)b), __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)));
}

// ---- linear predictors (FG_G_LIN): mu = c0 + s_0 c_0 + s_1 c_1 + ... in term order, two roundings per term like the
// interpreter's LOAD / FG_OP_DOT sequence it replaces.  Terms {u32 slot, u32 0, f64 c} are read four at a time with
// scalar loads; with 4 waves on the SIMD the exposed load latency of one wave is covered by the others.
// `DUAL`: the same products are added to two running sums (the suffix of a finite-difference pair, below).  The next group of
// four terms is requested before the current one is used (the pool is padded by one group, so the look-ahead past a record's
// last term stays inside it): a record of d terms is d / 4 round trips of ~one LDS latency instead of d / 4 scalar-memory
// latencies followed by d / 4 LDS latencies.
template <bool DUAL>
__device__ __forceinline__ void fg_lin_run(const FG_AS4 char *tb, uint32_t t0, uint32_t t1, const double *slots, int tw, double &mu, double &mu2) {
    uint32_t t = t0;
    if (t + 4 <= t1) {
        fg_u32x16 q = *(const FG_AS4 fg_u32x16 *)(tb + 16 * t);
        for (;;) {
            const fg_u32x16 qn = *(const FG_AS4 fg_u32x16 *)(tb + 16 * (t + 4));
            const double v0 = slots[q[0] * tw], v1 = slots[q[4] * tw], v2 = slots[q[8] * tw], v3 = slots[q[12] * tw];
            const double p0 = v0 * fg_dbl(q[2], q[3]), p1 = v1 * fg_dbl(q[6], q[7]), p2 = v2 * fg_dbl(q[10], q[11]), p3 = v3 * fg_dbl(q[14], q[15]);
            mu = mu + p0; mu = mu + p1; mu = mu + p2; mu = mu + p3;
            if (DUAL) { mu2 = mu2 + p0; mu2 = mu2 + p1; mu2 = mu2 + p2; mu2 = mu2 + p3; }
            t += 4;
            if (t + 4 > t1) break;
            q = qn;
        }
    }
    for (; t < t1; ++t) {
        const fg_u32x4 q = *(const FG_AS4 fg_u32x4 *)(tb + 16 * t);
        const double pr = slots[q[0] * tw] * fg_dbl(q[2], q[3]);
        mu = mu + pr;
        if (DUAL) mu2 = mu2 + pr;
    }
}
// short runs (the two- and three-term predictors of hierarchical models) keep the plain loop: the look-ahead version measured
// 20 % slower on them
__device__ __forceinline__ double fg_lin_prefix(const FG_AS4 char *tb, uint32_t t0, uint32_t t1, const double *slots, int tw, double mu) {
    if (t1 - t0 >= 8u && t1 > t0) {
        double unused = 0.0;
        fg_lin_run<false>(tb, t0, t1, slots, tw, mu, unused);
        return mu;
    }
    uint32_t t = t0;
    for (; t + 4 <= t1; t += 4) {
        const fg_u32x16 q = *(const FG_AS4 fg_u32x16 *)(tb + 16 * t);
        const double v0 = slots[q[0] * tw], v1 = slots[q[4] * tw], v2 = slots[q[8] * tw], v3 = slots[q[12] * tw];
        mu = mu + v0 * fg_dbl(q[2], q[3]);
        mu = mu + v1 * fg_dbl(q[6], q[7]);
        mu = mu + v2 * fg_dbl(q[10], q[11]);
        mu = mu + v3 * fg_dbl(q[14], q[15]);
    }
    for (; t < t1; ++t) {
        const fg_u32x4 q = *(const FG_AS4 fg_u32x4 *)(tb + 16 * t);
        mu = mu + slots[q[0] * tw] * fg_dbl(q[2], q[3]);
    }
    return mu;
}
// mean of a FG_G_LIN record (score: all terms)
__device__ __forceinline__ double fg_lin_mu(const fg_u32x16 &r, const double *pool, const double *slots, int tw) {
    const FG_AS4 char *tb = (const FG_AS4 char *)(uintptr_t)(pool + r[14]);
    return fg_lin_prefix(tb, 0u, r[15], slots, tw, fg_dbl(r[6], r[7]));
}
// means at q_i + h and q_i - h: the terms before the first one that reads q_i are common to both, from there on the
// two sums are carried side by side (a term that does not read q_i contributes the same product to both).  FG_G_LIN1: q_i is
// read by that one term only (a regression's x_ij beta_j) -- the rest of the suffix needs no test per term and runs four
// terms at a time like the prefix.
__device__ __forceinline__ void fg_lin_mu_dual(const fg_u32x16 &r, const double *pool, const double *slots, int tw, double h,
                                               double &mup, double &mum) {
    const FG_AS4 char *tb = (const FG_AS4 char *)(uintptr_t)(pool + r[14]);
    const uint32_t n = r[15], pos = r[2] >> 16, ci = r[3];
    const double mu = fg_lin_prefix(tb, 0u, pos < n ? pos : n, slots, tw, fg_dbl(r[6], r[7]));
    double mp = mu, mm = mu;
    if ((r[2] & FG_G_LIN1) && pos + 6u <= n) {
        const fg_u32x4 q = *(const FG_AS4 fg_u32x4 *)(tb + 16 * pos);
        const double v = slots[q[0] * tw], c = fg_dbl(q[2], q[3]);
        mp = mp + (v + h) * c; mm = mm + (v - h) * c;             // the perturbed slot holds orig +- h (hmc.rs:317-319)
        fg_lin_run<true>(tb, pos + 1u, n, slots, tw, mp, mm);
    } else {
        for (uint32_t t = pos; t < n; ++t) {
            const fg_u32x4 q = *(const FG_AS4 fg_u32x4 *)(tb + 16 * t);
            const double v = slots[q[0] * tw], c = fg_dbl(q[2], q[3]);
            if (q[0] == ci) { mp = mp + (v + h) * c; mm = mm + (v - h) * c; }
            else { const double pr = v * c; mp = mp + pr; mm = mm + pr; }
        }
    }
    mup = mp; mum = mm;
}

// ---- general distribution records (FG_G_GEN): any of the 17 log-densities with leaf operands, evaluated by the same
// fg_logpdf the interpreter calls (same guards, same evaluation order), at q_i + h and q_i - h (n_signs = 2) or once.
__device__ __forceinline__ void fg_gen_lp(const fg_u32x16 &r, double xs, const double *slots, int tw, double h, int n_signs, double *lp_out) {
    const uint32_t fl = r[2], kind = (fl >> 16) & 0xffu, ci = r[3];
    if (fl & FG_G_GEN_INVALID) { lp_out[0] = FG_NEG_INF; lp_out[1] = FG_NEG_INF; return; }
    const bool xint = (fl & FG_G_GEN_XINT) != 0u, hoisted = (fl & FG_G_GEN_HOISTED) != 0u, sh = (fl & FG_G_GEN_SH) != 0u, xh = (fl & FG_G_GEN_XH) != 0u;
    const double xv = (fl & FG_G_X_CONST) ? fg_dbl(r[4], r[5]) : xs;
    const long long xi = xint ? fg_as_i64(xv) : 0;
    const double xf = xint ? 0.0 : xv;
    const bool dual = n_signs == 2;
    const bool px = dual && !xint && !(fl & FG_G_X_CONST) && r[0] == ci;
    double p[3]; bool pp[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (fl & (FG_G_GEN_P0SLOT << q)) { const uint32_t idx = r[6 + 2 * q]; p[q] = slots[idx * tw]; pp[q] = dual && idx == ci; }
        else { p[q] = fg_dbl(r[6 + 2 * q], r[7 + 2 * q]); pp[q] = false; }
    }
    const double hh[2] = { fg_dbl(r[12], r[13]), fg_dbl(r[14], r[15]) };
#pragma nounroll
    for (int s = 0; s < n_signs; ++s) {                      // the perturbed operand holds orig +- h (hmc.rs:317-319)
        const double hs = dual ? (s == 0 ? h : -h) : 0.0;
#ifdef FG_GEN_INLINE_LOGPDF
        const double hh5[5] = { hh[0], hh[1], 0.0, xh ? hh[1] : 0.0, 0.0 };
        lp_out[s] = fg_logpdf(kind, hoisted, false, px ? xf + hs : xf, xi, pp[0] ? p[0] + hs : p[0], pp[1] ? p[1] + hs : p[1], pp[2] ? p[2] + hs : p[2], hh5, sh, xh);
#else
        lp_out[s] = fg_logpdf_cold(kind, hoisted, false, px ? xf + hs : xf, xi, pp[0] ? p[0] + hs : p[0], pp[1] ? p[1] + hs : p[1],
                                   pp[2] ? p[2] + hs : p[2], hh[0], hh[1], 0.0, xh ? hh[1] : 0.0, 0.0, sh, xh);   // out of line: the 17 densities stay out of the stream loops' registers
#endif
    }
}

// ---- mu = options[z] (FG_G_NSEL): the index site z is a discrete slot, so every lane selects its own option; the K
// options (f64 slots or constants) are walked with scalar loads and each lane keeps the one its z names -- the value the
// interpreter's FG_OP_GATHER returns (NaN when z is outside 0..K-1).  `ci` = perturbed coordinate or ~0u.
__device__ __forceinline__ void fg_nsel_mu(const fg_u32x16 &r, double zs, const double *pool, const double *slots, int tw, uint32_t ci, double h,
                                           double &mup, double &mum) {
    const uint32_t off = r[6], K = r[7];
    const long long zi = fg_as_i64(zs);
    const FG_AS4 char *tb = (const FG_AS4 char *)(uintptr_t)(pool + off);
    double mp = NAN, mm = NAN;
    for (uint32_t k = 0; k < K; ++k) {
        const fg_u32x4 q = *(const FG_AS4 fg_u32x4 *)(tb + 16 * k);
        const double v = q[1] ? fg_dbl(q[2], q[3]) : slots[q[0] * tw];
        const bool dep = !q[1] && q[0] == ci;                // the perturbed slot holds orig +- h (hmc.rs:317-319)
        const double vp = dep ? v + h : v, vm = dep ? v - h : v;
        if (zi == (long long)k) { mp = vp; mm = vm; }
    }
    mup = mp; mum = mm;
}

// The same option select for a SCORING run (no perturbation): every lane fetches the pool entry its own z names (one
// 16-byte gather, the table is L1-resident) and then the slot that entry names -- one round trip instead of K scalar
// fetches, LDS reads and selects.  The value is the one fg_nsel_mu / FG_OP_GATHER return (NaN when z is outside 0..K-1).
__device__ __forceinline__ double fg_nsel_mu_lane(const fg_u32x16 &r, double zs, const double *pool, const double *slots, int tw) {
    const uint32_t off = r[6], K = r[7];
    const long long zi = fg_as_i64(zs);
    const bool ok = zi >= 0 && zi < (long long)K;
    const double *ent = pool + off + 2 * (ok ? (int)zi : 0);
    const long long hdr = fg_as_i64(ent[0]);                 // {u32 slot, u32 is_const}
    const double cst = ent[1];
    const uint32_t slot = (uint32_t)hdr, is_const = (uint32_t)(hdr >> 32);
    const double v = is_const ? cst : slots[slot * tw];
    return ok ? v : NAN;
}

// Two in-order sums of LDS rows, side by side: sa = ((0 + A[0]) + A[1]) + ... over na rows, sb likewise over nb rows -- the
// additions of a sequential scoring run (interpreters.rs:76-163).  The wave that adds is alone on the path of its tile and
// issues one instruction per four cycles, so the loop is written for instruction count: rows are read at COMPILE-TIME offsets
// from a moving base (the compiler pairs them into ds_read2st64_b64 -- the row stride is 64 x 8 B -- and spends no address
// arithmetic per row), CH rows of each chain are in flight (fg_inorder_sum1: and the next CH already requested while the
// current ones are added).  A tail shorter than CH reads CH rows all the same (what lies behind a chain's last row is still the tile's LDS)
// and adds +0.0 in place of the strangers: a running sum that started from +0.0 is never -0.0, so `+ 0.0` changes no bit.
template <int CH>
__device__ __forceinline__ void fg_inorder_tail(const double *p, int rem, int tw, double &acc) {
    double x[CH];
    for (; rem >= CH; rem -= CH, p += (long long)CH * tw) {
#pragma unroll
        for (int q = 0; q < CH; ++q) x[q] = p[q * tw];
#pragma unroll
        for (int q = 0; q < CH; ++q) acc += x[q];
    }
    if (rem > 0) {
#pragma unroll
        for (int q = 0; q < CH; ++q) x[q] = p[q * tw];
#pragma unroll
        for (int q = 0; q < CH; ++q) acc += (q < rem) ? x[q] : 0.0;
    }
}
template <int CH = 8>
__device__ __forceinline__ void fg_inorder_sums2(const double *A, int na, const double *B, int nb, int tw, double &sa, double &sb) {
    const int n = na < nb ? na : nb;
    double a = 0.0, b = 0.0;
    const double *pa = A, *pb = B;
    int k = 0;
    for (; k + CH <= n; k += CH, pa += (long long)CH * tw, pb += (long long)CH * tw) {    // both chains, CH rows of each in flight
        double x[CH], u[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) { x[q] = pa[q * tw]; u[q] = pb[q * tw]; }
#pragma unroll
        for (int q = 0; q < CH; ++q) { a += x[q]; b += u[q]; }
    }
    fg_inorder_tail<CH>(pa, na - k, tw, a);
    fg_inorder_tail<CH>(pb, nb - k, tw, b);
    sa = a; sb = b;
}

// rows [0, rem) of p added, in order, to TWO running sums (the two whole log-joints of a dense finite difference share every
// term that did not move): the same chunking, the same +0.0 padding of the last chunk
template <int CH>
__device__ __forceinline__ void fg_inorder_run2(const double *p, int rem, int tw, double &a, double &b) {
    double x[CH];
    for (; rem >= CH; rem -= CH, p += (long long)CH * tw) {
#pragma unroll
        for (int q = 0; q < CH; ++q) x[q] = p[q * tw];
#pragma unroll
        for (int q = 0; q < CH; ++q) { a += x[q]; b += x[q]; }
    }
    if (rem > 0) {
#pragma unroll
        for (int q = 0; q < CH; ++q) x[q] = p[q * tw];
#pragma unroll
        for (int q = 0; q < CH; ++q) { const double v = (q < rem) ? x[q] : 0.0; a += v; b += v; }
    }
}
// ... to ONE running sum (the rows ahead of the first moved term: the two log-joints are still the same number there)
template <int CH>
__device__ __forceinline__ void fg_inorder_run1(const double *p, int rem, int tw, double &a) {
    double x[CH];
    for (; rem >= CH; rem -= CH, p += (long long)CH * tw) {
#pragma unroll
        for (int q = 0; q < CH; ++q) x[q] = p[q * tw];
#pragma unroll
        for (int q = 0; q < CH; ++q) a += x[q];
    }
    if (rem > 0) {
#pragma unroll
        for (int q = 0; q < CH; ++q) x[q] = p[q * tw];
#pragma unroll
        for (int q = 0; q < CH; ++q) a += (q < rem) ? x[q] : 0.0;
    }
}
// one chain of the above (a kernel whose waves take one sum each)
template <int CH = 8>
__device__ __forceinline__ double fg_inorder_sum1(const double *A, int na, int tw) {
    double a = 0.0;
    const double *pa = A;
    int k = 0;
    double x[CH], y[CH];
#define FG_SUM_LOAD(X) { _Pragma("unroll") for (int q = 0; q < CH; ++q) X[q] = pa[q * tw]; pa += (long long)CH * tw; k += CH; }
#define FG_SUM_ADD(X) { _Pragma("unroll") for (int q = 0; q < CH; ++q) a += X[q]; }
    if (na >= CH) {
        FG_SUM_LOAD(x)
        for (;;) {
            if (k + CH > na) { FG_SUM_ADD(x) break; }
            FG_SUM_LOAD(y)
            FG_SUM_ADD(x)
            if (k + CH > na) { FG_SUM_ADD(y) break; }
            FG_SUM_LOAD(x)
            FG_SUM_ADD(y)
        }
    }
#undef FG_SUM_LOAD
#undef FG_SUM_ADD
    fg_inorder_tail<CH>(pa, na - k, tw, a);
    return a;
}

// The same sums for a wave whose instruction count IS its tile's path (the decider of fg_mh_mw2_body.h): a tail of rem < 8 rows is
// rem loads and rem additions behind two scalar jumps -- the rows go to the END of the register chunk and both unrolled sequences are
// entered (8 - rem) places in -- instead of 8 loads, 8 additions and 16 selects of "+ 0.0" (a lone wave pays ~8 cycles per
// instruction whatever it is).  Split in two calls so that the caller can issue other LDS reads between the chunks and the tails.
struct FgSums2 { double a, b; const double *pa, *pb; int ra, rb; };
__device__ __forceinline__ FgSums2 fg_inorder_sums2_chunks(const double *A, int na, const double *B, int nb, int tw) {
    FgSums2 s; s.a = 0.0; s.b = 0.0; s.pa = A; s.pb = B;
    const int n = na < nb ? na : nb;
    int k = 0;
    for (; k + 8 <= n; k += 8, s.pa += 8LL * tw, s.pb += 8LL * tw) {         // both chains, 8 rows of each in flight
        double x[8], u[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { x[q] = s.pa[q * tw]; u[q] = s.pb[q * tw]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) { s.a += x[q]; s.b += u[q]; }
    }
    s.ra = na - k; s.rb = nb - k;
    for (; s.ra >= 8; s.ra -= 8, s.pa += 8LL * tw) { double x[8]; _Pragma("unroll") for (int q = 0; q < 8; ++q) x[q] = s.pa[q * tw]; _Pragma("unroll") for (int q = 0; q < 8; ++q) s.a += x[q]; }
    for (; s.rb >= 8; s.rb -= 8, s.pb += 8LL * tw) { double x[8]; _Pragma("unroll") for (int q = 0; q < 8; ++q) x[q] = s.pb[q * tw]; _Pragma("unroll") for (int q = 0; q < 8; ++q) s.b += x[q]; }
    return s;
}
#define FG_TAIL_LOAD(X, P, R)                                                                                  \
    switch (R) { case 7: X[1] = P[1 * tw]; [[fallthrough]]; case 6: X[2] = P[2 * tw]; [[fallthrough]]; case 5: X[3] = P[3 * tw]; [[fallthrough]]; \
                 case 4: X[4] = P[4 * tw]; [[fallthrough]]; case 3: X[5] = P[5 * tw]; [[fallthrough]]; case 2: X[6] = P[6 * tw]; [[fallthrough]]; \
                 case 1: X[7] = P[7 * tw]; [[fallthrough]]; default: break; }
#define FG_TAIL_ADD(X, ACC, R)                                                                                 \
    switch (R) { case 7: ACC += X[1]; [[fallthrough]]; case 6: ACC += X[2]; [[fallthrough]]; case 5: ACC += X[3]; [[fallthrough]]; \
                 case 4: ACC += X[4]; [[fallthrough]]; case 3: ACC += X[5]; [[fallthrough]]; case 2: ACC += X[6]; [[fallthrough]]; \
                 case 1: ACC += X[7]; [[fallthrough]]; default: break; }
__device__ __forceinline__ void fg_inorder_sums2_tails(FgSums2 &s, int tw) {
    double x[8] = {0, 0, 0, 0, 0, 0, 0, 0}, u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double *qa = s.pa - (long long)(8 - s.ra) * tw, *qb = s.pb - (long long)(8 - s.rb) * tw;   // row k of a tail -> register 8 - rem + k
    FG_TAIL_LOAD(x, qa, s.ra)
    FG_TAIL_LOAD(u, qb, s.rb)
    FG_TAIL_ADD(x, s.a, s.ra)
    FG_TAIL_ADD(u, s.b, s.rb)
}
__device__ __forceinline__ double fg_inorder_sum1_exact(const double *A, int na, int tw) {
    double a = fg_inorder_sum1(A, na & ~7, tw);
    const int rem = na & 7;
    double x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double *qa = A + (long long)(na & ~7) * tw - (long long)(8 - rem) * tw;
    FG_TAIL_LOAD(x, qa, rem)
    FG_TAIL_ADD(x, a, rem)
    return a;
}
#undef FG_TAIL_LOAD
#undef FG_TAIL_ADD

struct FgGradAcc { double sp, sm, prip, prim; bool bad; };
struct FgGradK { double h, hk, two_h, rcp_2h; bool two_kicks; };   // wave-uniform constants of one gradient

// one record: r = {xi, mi, flags, coord, ximm, mimm, sigma, 1/sigma, ln sigma, maskx, maskm}; xs, ms = LDS values
// of its operands (the always-zero slot for constants); pv = p[coord] (pre-read).  ONE straight-line form for every
// operand combination: operand value = slot + imm exactly as FG_OP_NORMAL_FAST forms it, perturbation = h & mask
// (scalar), so the only branches are the rare ones (non-power-of-two sigma, first observe record, last record).
// Within a coordinate the prior records come first, then the observe records (FG_G_SWITCH on the first of them):
// log_prior and log_likelihood are summed separately and added at the end, exactly like total_log_weight
// (trace.rs:198-200).  A non-finite x or mu gives z = NaN or +-inf -> lp NaN or -inf -> a non-finite g -> divergent,
// the same verdict as the reference's -inf log-density (hmc.rs:323-325); no guard is needed here.
template <int RK, bool AN>
__device__ __forceinline__ void fg_grec_math(const fg_u32x16 &r, double xs, double ms, double pv, FgGradAcc &A, const FgGradK &K,
                                             const double *pool, const double *slots, double *pl, int tw, double *gout, long long gstride, bool live) {
    const uint32_t fl = r[2];
    if (AN) {                                              // FG_GRAD_ANALYTIC: d/dq_i of -(x - mu)^2 / (2 sigma^2), one evaluation
        const double x = (fl & FG_G_X_CONST) ? fg_dbl(r[4], r[5]) : xs;
        double m = (fl & FG_G_M_CONST) ? fg_dbl(r[6], r[7]) : ms, ci = (fl & FG_G_PERT_M) ? 1.0 : 0.0;
        if (RK >= 1 && (fl & FG_G_LIN)) {                  // d mu / d q_i = sum of the coefficients of the terms that read q_i
            const FG_AS4 char *tb = (const FG_AS4 char *)(uintptr_t)(pool + r[14]);
            m = fg_dbl(r[6], r[7]); ci = 0.0;
            for (uint32_t t = 0; t < r[15]; ++t) {
                const fg_u32x4 q = *(const FG_AS4 fg_u32x4 *)(tb + 16 * t);
                const double c = fg_dbl(q[2], q[3]);
                m = m + slots[q[0] * tw] * c;
                if (q[0] == r[3]) ci += c;
            }
        }
        const double inv = fg_dbl(r[10], r[11]), sg = fg_dbl(r[8], r[9]);
        const double dl = x - m;
        const double w = (fl & FG_G_POW2) ? (dl * inv) * inv : (dl / sg) / sg;       // (x - mu) / sigma^2
        A.sp += ((fl & FG_G_PERT_X) ? ci - 1.0 : ci) * w;                             // d lp / d mu = +w, d lp / d x = -w
        if (fl & FG_G_END) {
            const double g = A.sp;
            A.bad = A.bad || !fg_finite(g);
            double p = pv + K.hk * g;
            if (K.two_kicks) p += K.hk * g;
            pl[r[3] * tw] = p;
            if (gout && live) gout[(long long)r[3] * gstride] = g;
            A.sp = 0.0;
        }
        return;
    }
    // x - mu at q_i + h and q_i - h: the perturbed operand holds orig +- h (hmc.rs:317-319), the other its value
    // (a constant operand is the record's immediate, a scalar).  Four forms, four additions each, selected by
    // scalar branches: adding the zero perturbation / the zero slot of a uniform form would give the same bits
    // for two more f64 instructions per record, and f64 issue is what bounds this loop.
    double lpp, lpm;
    if (RK >= 2 && __builtin_expect((fl & FG_G_GEN) != 0u, 0)) {   // general distribution record: fg_logpdf at both signs
        double lp2[2];
        fg_gen_lp(r, xs, slots, tw, K.h, 2, lp2);
        lpp = lp2[0]; lpm = lp2[1];
    } else {
    double dlp, dlm;
    if (RK >= 2 && __builtin_expect((fl & FG_G_NSEL) != 0u, 0)) {   // mu = options[z]
        double mup, mum;
        fg_nsel_mu(r, ms, pool, slots, tw, r[3], K.h, mup, mum);
        const double x = (fl & FG_G_X_CONST) ? fg_dbl(r[4], r[5]) : xs;
        const double hx = (fl & FG_G_PERT_X) ? K.h : 0.0;
        dlp = (x + hx) - mup; dlm = (x - hx) - mum;
    } else if (RK >= 1 && __builtin_expect((fl & FG_G_LIN) != 0u, 0)) {   // linear predictor: both means, then x - mu
        double mup, mum;
        fg_lin_mu_dual(r, pool, slots, tw, K.h, mup, mum);
        const double x = (fl & FG_G_X_CONST) ? fg_dbl(r[4], r[5]) : xs;
        const double hx = (fl & FG_G_PERT_X) ? K.h : 0.0;
        dlp = (x + hx) - mup; dlm = (x - hx) - mum;
    } else if (fl & FG_G_PERT_X) {
        const double xp = xs + K.h, xm = xs - K.h;
        if (fl & FG_G_M_CONST) { const double m = fg_dbl(r[6], r[7]); dlp = xp - m; dlm = xm - m; }
        else if (__builtin_expect((fl & FG_G_PERT_M) != 0u, 0)) { dlp = xp - (ms + K.h); dlm = xm - (ms - K.h); }   // x and mu are the same site
        else { dlp = xp - ms; dlm = xm - ms; }
    } else {
        const double mp = ms + K.h, mm = ms - K.h;
        if (fl & FG_G_X_CONST) { const double x = fg_dbl(r[4], r[5]); dlp = x - mp; dlm = x - mm; }
        else { dlp = xs - mp; dlm = xs - mm; }
    }
    const double inv = fg_dbl(r[10], r[11]);
    double zp = dlp * inv, zm = dlm * inv;                 // exact quotient when sigma = 2^k
    if (__builtin_expect(!(fl & FG_G_POW2), 0)) {          // (x - mu) / sigma, distribution.rs:205
        const double sg = fg_dbl(r[8], r[9]);
        if (fl & FG_G_DIV) { zp = dlp / sg; zm = dlm / sg; }
        else { zp = fg_div_const(dlp, sg, inv); zm = fg_div_const(dlm, sg, inv); }
    }
    const double lns = fg_dbl(r[12], r[13]);
    lpp = -0.5 * zp * zp - lns - 0.5 * FG_LN_2PI;       // distribution.rs:207
    lpm = -0.5 * zm * zm - lns - 0.5 * FG_LN_2PI;
    }
    if (__builtin_expect((fl & FG_G_SWITCH) != 0u, 0)) {  // a real (scalar) branch, not eight v_cndmask
        A.prip = A.sp; A.prim = A.sm; A.sp = 0.0; A.sm = 0.0;
        asm volatile("" ::: "memory");
    }
    A.sp += lpp; A.sm += lpm;
    if (__builtin_expect((fl & FG_G_END) != 0u, 0)) {
        // no observe record: the running sums are the prior sums and prip = prim = 0 (= log_likelihood)
        const double tp = A.prip + A.sp, tm = A.prim + A.sm;                 // total_log_weight (log_factors = +0.0 adds nothing)
        const double n = tp - tm;
        double g = fg_div_const(n, K.two_h, K.rcp_2h);                       // (lp - lm) / (2h), hmc.rs:322
        const uint32_t ne = (uint32_t)(__double_as_longlong(n) >> 32) & 0x7fffffffu;
        if (__builtin_expect(__any(!(n == 0.0 || (ne - 0x0c800000u) < 0x6f000000u)), 0)) g = n / K.two_h;   // |n| outside [2^-823, 2^953]
        A.bad = A.bad || !fg_finite(g);
        double p = pv + K.hk * g;
        if (K.two_kicks) p += K.hk * g;
        pl[r[3] * tw] = p;
        if (gout && live) gout[(long long)r[3] * gstride] = g;
        A.sp = A.sm = A.prip = A.prim = 0.0;
    }
}

// timing-only experiment switches (tools/exp_breakdown.sh; results are wrong by construction)
#ifdef FG_EXP_G_NOWAIT
#define FG_G_WAIT
#else
#define FG_G_WAIT __builtin_amdgcn_s_waitcnt(0xc07f);
#endif
#ifdef FG_EXP_G_NOLDS
#define FG_G_LDS(RB, XB, MB, PB) XB = x0; MB = x0; PB = x0;
#else
#define FG_G_LDS(RB, XB, MB, PB) XB = slots[RB[0] * tw]; MB = slots[RB[1] * tw]; PB = pl[RB[3] * tw];
#endif
#ifdef FG_EXP_G_NOMATH
#define FG_G_MATH(RA, XA, MA, PA) A.sp += XA + MA + PA + fg_dbl(RA[4], RA[5]);
#else
#define FG_G_MATH(RA, XA, MA, PA) fg_grec_math<RK, AN>(RA, XA, MA, PA, A, K, pool, slots, pl, tw, gout, gstride, live);
#endif
#ifdef FG_EXP_G_NOFETCH
#define FG_G_FETCH(RD) RD = fg_fetch_grec(g, (k + 3) & 1);
#else
#define FG_G_FETCH(RD) RD = fg_fetch_grec(g, k + 3);
#endif

#define FG_GSTAGE(RA, XA, MA, PA, RB, XB, MB, PB, RD)                                              \
    FG_G_WAIT                                      /* lgkmcnt(0): RB, RC (issued a stage ago), XA, MA, PA have landed */ \
    FG_G_FETCH(RD)                                                                                  \
    FG_G_LDS(RB, XB, MB, PB)                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    FG_G_MATH(RA, XA, MA, PA)                                                                       \
    if (++k >= n) break;

// One gradient: every coordinate's g_i and the half-kick(s) on p_i.  Returns "some force
// component was non-finite" for this lane.  Records are fetched THREE ahead (four rotating
// 16-SGPR buffers), operands one ahead.
// `g`, `n`: the whole stream (P.gstream, P.n_gstream) or one wave's run of whole coordinates of it (multi-wave HMC);
// reading up to 3 records past `n` is safe either way (the next wave's records or the pad records).
// RK = record kinds the instantiation understands: 0 fast Normals only, 1 + linear predictors (FG_G_LIN), 2 + general
// distribution records (FG_G_GEN).  The 128-VGPR multi-wave kernel is sensitive to every register the hot loop holds,
// so a program only pays for the kinds it contains.
template <int RK, bool AN = false>
__device__ __forceinline__ bool fg_grad_stream(const FgGradRec *g, const int n, const double *pool, double *slots, double *pl, int tw, double h, double hk,
                                               bool two_kicks, double *gout, long long gstride, bool live) {
    FgGradK K;
    K.h = fg_uniform(h); K.hk = hk; K.two_kicks = two_kicks;
    K.two_h = fg_uniform(2.0 * h); K.rcp_2h = fg_uniform(1.0 / (2.0 * h));     // wave-uniform: keep them in SGPRs, not in (spillable) VGPRs
    FgGradAcc A = {0.0, 0.0, 0.0, 0.0, false};   // running sums, stashed prior sums
    fg_u32x16 r0 = fg_fetch_grec(g, 0), r1 = fg_fetch_grec(g, 1), r2 = fg_fetch_grec(g, 2), r3;
    double x0 = slots[r0[0] * tw], m0 = slots[r0[1] * tw], p0 = pl[r0[3] * tw];
    double x1, m1, p1, x2, m2, p2, x3, m3, p3;
    int k = 0;
    for (;;) {
        FG_GSTAGE(r0, x0, m0, p0, r1, x1, m1, p1, r3)
        FG_GSTAGE(r1, x1, m1, p1, r2, x2, m2, p2, r0)
        FG_GSTAGE(r2, x2, m2, p2, r3, x3, m3, p3, r1)
        FG_GSTAGE(r3, x3, m3, p3, r0, x0, m0, p0, r2)
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    return A.bad;
}

// The endpoint score of an all-fast-Normal program: one record per statement in program order, the same values and
// the same additions as FG_OP_NORMAL_FAST in the interpreter (operand = slot or immediate; z != z -> -inf guard;
// log_prior and log_likelihood accumulated separately).  Records are fetched two ahead, operands one ahead.
template <int RK>
// lane_pool: where the per-lane table lookups (Categorical ln p, option lists) read the constant pool -- a kernel may stage a
// small pool into LDS; linear-predictor terms always come from `pool` by scalar loads.
__device__ __forceinline__ double fg_score_one(const fg_u32x16 &r, double xs, double ms, const double *pool, const double *slots, int tw, FgAcc3 &A,
                                               const double *lane_pool = nullptr) {
    const double *lp_tab = lane_pool ? lane_pool : pool;
    const uint32_t fl = r[2];
    double lp;
    if (RK == 2 && __builtin_expect((fl & FG_G_GEN) != 0u, 0)) {      // RK = 3: linear predictors, option selects and Categorical tables, but no general records
        double lp2[2];
        fg_gen_lp(r, xs, slots, tw, 0.0, 1, lp2);
        lp = lp2[0];
    } else if (RK >= 2 && __builtin_expect((fl & FG_G_CATC) != 0u, 0)) {   // Categorical site, constant table: ln p[z] precomputed
        const long long zi = fg_as_i64(xs);
        const uint32_t base = r[6], K = r[7];
        lp = (zi < 0 || zi >= (long long)K) ? FG_NEG_INF : lp_tab[base + K + (zi < 0 || zi >= (long long)K ? 0 : (int)zi)];
    } else {
        // operand = slot + immediate, as FG_OP_NORMAL_FAST forms it (fg_interp.h): a constant operand reads the always-zero
        // slot, a site operand has the immediate 0.0 -- one f64 add per operand instead of a flag test and a select
        const double x = xs + fg_dbl(r[4], r[5]);
        double m = (RK == 0) ? ms + fg_dbl(r[6], r[7]) : ((fl & FG_G_M_CONST) ? fg_dbl(r[6], r[7]) : ms);
        if (RK >= 2 && __builtin_expect((fl & FG_G_NSEL) != 0u, 0)) m = fg_nsel_mu_lane(r, ms, lp_tab, slots, tw);
        if (RK >= 1 && __builtin_expect((fl & FG_G_LIN) != 0u, 0)) m = fg_lin_mu(r, pool, slots, tw);
        const double dl = x - m, inv = fg_dbl(r[10], r[11]);
        double z = dl * inv;
        if (__builtin_expect(!(fl & FG_G_POW2), 0)) {
            const double sg = fg_dbl(r[8], r[9]);
            z = (fl & FG_G_DIV) ? dl / sg : fg_div_const(dl, sg, inv);
        }
        lp = -0.5 * z * z - fg_dbl(r[12], r[13]) - 0.5 * FG_LN_2PI;
        lp = (z != z) ? FG_NEG_INF : lp;
    }
    if (fl & FG_S_OBS) A.lik += lp; else A.prior += lp;
    return lp;
}
template <int RK>
__device__ __forceinline__ void fg_score_stream(const FgGradRec *g, const int n, const double *pool, const double *slots, int tw, FgAcc3 &A) {
    fg_u32x16 r0 = fg_fetch_grec(g, 0), r1 = fg_fetch_grec(g, 1), r2;
    double x0 = slots[r0[0] * tw], m0 = slots[r0[1] * tw], x1, m1;
    for (int k = 0; k < n; ++k) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        r2 = fg_fetch_grec(g, k + 2);
        x1 = slots[r1[0] * tw]; m1 = slots[r1[1] * tw];
        __builtin_amdgcn_sched_barrier(0);
        fg_score_one<RK>(r0, x0, m0, pool, slots, tw, A);
        r0 = r1; r1 = r2; x0 = x1; m0 = m1;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
}

// grad_log_joint VERBATIM (hmc.rs:304-329) for an all-fast-Normal program: for every coordinate i of [k0, k1) the
// WHOLE program is scored at q + h e_i and at q - h e_i (two full passes of the score stream, fused: a statement
// that does not read q_i has the same value in both passes and is evaluated once, which changes no bit), the
// three accumulators are summed like total_log_weight, g_i = (lp - lm) / (2h), and p_i is kicked.  Nothing is
// written to the shared q tile -- the perturbation is applied to the operand on the fly (coordinate i lives in
// slot i) -- so the waves of a tile can work on different coordinates at the same time.
__device__ __forceinline__ bool fg_grad_dense_stream(const FgGradRec *g, const int n, int k0, int k1, const double *slots, double *pl, int tw,
                                                     double h, double hk, bool two_kicks, double *gout, long long gstride, bool live) {
    const double hu = fg_uniform(h), two_h = fg_uniform(2.0 * h), rcp_2h = fg_uniform(1.0 / (2.0 * h));
    bool bad = false;
    for (int i = k0; i < k1; ++i) {
        FgAcc3 Ap = {0.0, 0.0, 0.0}, Am = {0.0, 0.0, 0.0};
        fg_u32x16 r0 = fg_fetch_grec(g, 0), r1 = fg_fetch_grec(g, 1), r2;
        double x0 = slots[r0[0] * tw], m0 = slots[r0[1] * tw], x1, m1;
        for (int k = 0; k < n; ++k) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            r2 = fg_fetch_grec(g, k + 2);
            x1 = slots[r1[0] * tw]; m1 = slots[r1[1] * tw];
            __builtin_amdgcn_sched_barrier(0);
            {
                const uint32_t fl = r0[2];
                const double x = (fl & FG_G_X_CONST) ? fg_dbl(r0[4], r0[5]) : x0, m = (fl & FG_G_M_CONST) ? fg_dbl(r0[6], r0[7]) : m0;
                const double inv = fg_dbl(r0[10], r0[11]), lns = fg_dbl(r0[12], r0[13]);
                const bool dep_x = r0[0] == (uint32_t)i, dep_m = r0[1] == (uint32_t)i;       // scalar: does this statement read q_i?
                double lpp, lpm;
                if (dep_x || dep_m) {
                    const double hx = dep_x ? hu : 0.0, hm = dep_m ? hu : 0.0;
                    const double dlp = (x + hx) - (m + hm), dlm = (x - hx) - (m - hm);
                    double zp = dlp * inv, zm = dlm * inv;
                    if (__builtin_expect(!(fl & FG_G_POW2), 0)) {
                        const double sg = fg_dbl(r0[8], r0[9]);
                        if (fl & FG_G_DIV) { zp = dlp / sg; zm = dlm / sg; } else { zp = fg_div_const(dlp, sg, inv); zm = fg_div_const(dlm, sg, inv); }
                    }
                    lpp = -0.5 * zp * zp - lns - 0.5 * FG_LN_2PI; lpp = (zp != zp) ? FG_NEG_INF : lpp;
                    lpm = -0.5 * zm * zm - lns - 0.5 * FG_LN_2PI; lpm = (zm != zm) ? FG_NEG_INF : lpm;
                } else {
                    const double dl = x - m;
                    double z = dl * inv;
                    if (__builtin_expect(!(fl & FG_G_POW2), 0)) { const double sg = fg_dbl(r0[8], r0[9]); z = (fl & FG_G_DIV) ? dl / sg : fg_div_const(dl, sg, inv); }
                    lpp = -0.5 * z * z - lns - 0.5 * FG_LN_2PI; lpp = (z != z) ? FG_NEG_INF : lpp;
                    lpm = lpp;
                }
                if (fl & FG_S_OBS) { Ap.lik += lpp; Am.lik += lpm; } else { Ap.prior += lpp; Am.prior += lpm; }
            }
            r0 = r1; r1 = r2; x0 = x1; m0 = m1;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const double nn = fg_total(Ap) - fg_total(Am);
        double gi = fg_div_const(nn, two_h, rcp_2h);                                   // hmc.rs:322
        const uint32_t ne = (uint32_t)(__double_as_longlong(nn) >> 32) & 0x7fffffffu;
        if (__builtin_expect(__any(!(nn == 0.0 || (ne - 0x0c800000u) < 0x6f000000u)), 0)) gi = nn / two_h;
        bad = bad || !fg_finite(gi);
        double p = pl[i * tw] + hk * gi;
        if (two_kicks) p += hk * gi;
        pl[i * tw] = p;
        if (gout && live) gout[(long long)i * gstride] = gi;
    }
    return bad;
}
