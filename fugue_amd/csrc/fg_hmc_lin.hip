// fg_hmc_lin.hip -- HmcSession::step x n (hmc.rs:819-919) for DENSE REGRESSIONS: every coordinate's force terms are its own
// prior record(s) plus the same N observe statements  y_i ~ Normal(c0_i + sum_t q[s_t] c_it, sigma_i), each reading all D
// coordinates once in one common term order (BASELINE configs[2]: examples/linear_regression.rs:396-424 at 32 coefficients
// x 1 024 observations).
//
// The reference's finite difference (grad_log_joint, hmc.rs:304-329) makes every (coordinate j, observation i, sign) a fresh
// in-order sum over the predictor's D terms: O(D^2 N) additions per gradient.  The gradient stream (fg_gradstream.h) walks
// them coordinate by coordinate, one record per (j, i): per record D products, the prefix sum up to the coordinate's term and
// two suffix sums -- ~150 instructions per (j, i) pair at D = 32.  But the products P_it = q_t c_it and the prefix sums
// S_it = c0_i + P_i0 + ... + P_i,t-1 are the SAME numbers for every coordinate: only the two suffix chains
//     mu+-_ij = ((S_ij + (q_j +- h) c_ij) + P_i,j+1) + ... + P_i,D-1
// belong to the pair.  So the traversal here is OBSERVATION-MAJOR: a wave owns M = D / W term positions, keeps all of q in
// registers, and per observation forms the D products and the prefix ONCE, carrying the 2 M suffix chains of its own
// coordinates side by side -- the same additions and multiplications in the same order per chain, hence bit-identical forces
// (tests/test_gpu_parity.py::test_hmc_lin_kernel_is_bit_identical) at ~65 instead of ~150 instructions per pair.
//
// Which positions a wave owns decides how long its suffix chains are (position p costs 2 (D - p) additions).  Positions are
// dealt in blocks of 2 W: wave w takes 2 W b + w and 2 W b + (2 W - 1 - w) of every block b, so every wave carries the same
// number of additions.  The term loop is fully unrolled and the set of active chains changes at compile-time positions: one
// straight-line instance per (D, W, wave) -- the kernel switches on the wave index once per gradient, not per term (round 2's
// attempt tested four owned positions per term with scalar branches and gained 5 %).  Coefficients arrive by scalar loads,
// half a row ahead; the hot loop touches no LDS (SMEM and LDS share a counter and SMEM returns out of order).
//
// Everything around the gradient is k_hmc_stream_steps' (fg_engine.hip): momentum pairs by whichever wave gets them,
// barriers around the drift, wave 0's in-order kinetic sums, endpoint score (score stream, program order), accept and dual
// averaging; commit / roll back by the owning wave.
#include "fg_engine_internal.h"
#include "fg_gradstream.h"
#include "fg_cold.h"

#define FG_LIN_WMAX 16

// term position of own slot A of wave WV (ascending in A)
template <int W, int WV, int A> struct FgLinPos { static constexpr int v = 2 * W * (A / 2) + ((A & 1) ? 2 * W - 1 - WV : WV); };
__host__ __device__ constexpr int fg_lin_pos(int W, int wv, int a) { return 2 * W * (a / 2) + ((a & 1) ? 2 * W - 1 - wv : wv); }

// table rows are 16-byte aligned (the coefficients start 16 bytes into a row): scalar loads need no more
typedef fg_u32x16 fg_u32x16_r __attribute__((aligned(16)));
typedef fg_u32x8 fg_u32x8_r __attribute__((aligned(16)));
typedef __attribute__((address_space(3))) double fg_lds_double;

// HB coefficients of one half row in SGPRs
template <int HB> struct FgLinHalf;
template <> struct FgLinHalf<16> {
    fg_u32x16 a, b;
    __device__ __forceinline__ void load(const FG_AS4 char *p) { a = *(const FG_AS4 fg_u32x16_r *)p; b = *(const FG_AS4 fg_u32x16_r *)(p + 64); }
    __device__ __forceinline__ double get(int t) const { return t < 8 ? fg_dbl(a[2 * t], a[2 * t + 1]) : fg_dbl(b[2 * (t - 8)], b[2 * (t - 8) + 1]); }
};
template <> struct FgLinHalf<8> {
    fg_u32x16 a;
    __device__ __forceinline__ void load(const FG_AS4 char *p) { a = *(const FG_AS4 fg_u32x16_r *)p; }
    __device__ __forceinline__ double get(int t) const { return fg_dbl(a[2 * t], a[2 * t + 1]); }
};
template <> struct FgLinHalf<4> {
    fg_u32x8 a;
    __device__ __forceinline__ void load(const FG_AS4 char *p) { a = *(const FG_AS4 fg_u32x8_r *)p; }
    __device__ __forceinline__ double get(int t) const { return fg_dbl(a[2 * t], a[2 * t + 1]); }
};

__device__ __forceinline__ const FG_AS4 char *fg_uniform_ptr(const void *p) {
    const unsigned long long b = (unsigned long long)(uintptr_t)p;
    // readfirstlane returns int: through uint32_t, or a low word with its top bit set sign-extends over the high word
    const unsigned long long u = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b) |
                                 ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32);
    return (const FG_AS4 char *)(uintptr_t)u;
}

// log-density of an own-coordinate prior record at q + h and q - h: fg_grec_math's arithmetic for a fast Normal whose other
// operand is a constant (fg_gradstream.h)
__device__ __forceinline__ void fg_lin_prior_pair(const fg_u32x16 &r, double qv, double h, double &lpp, double &lpm) {
    const uint32_t fl = r[2];
    double dlp, dlm;
    if (fl & FG_G_PERT_X) { const double xp = qv + h, xm = qv - h, m = fg_dbl(r[6], r[7]); dlp = xp - m; dlm = xm - m; }
    else { const double mp = qv + h, mm = qv - h, x = fg_dbl(r[4], r[5]); dlp = x - mp; dlm = x - mm; }
    const double inv = fg_dbl(r[10], r[11]);
    double zp = dlp * inv, zm = dlm * inv;
    if (!(fl & FG_G_POW2)) {
        const double sg = fg_dbl(r[8], r[9]);
        if (fl & FG_G_DIV) { zp = dlp / sg; zm = dlm / sg; }
        else { zp = fg_div_const(dlp, sg, inv); zm = fg_div_const(dlm, sg, inv); }
    }
    const double lns = fg_dbl(r[12], r[13]);
    lpp = -0.5 * zp * zp - lns - 0.5 * FG_LN_2PI;
    lpm = -0.5 * zm * zm - lns - 0.5 * FG_LN_2PI;
}

// One gradient for the M own coordinates of wave WV: g_k = (lp(q + h e_k) - lp(q - h e_k)) / (2h) over the coordinate's prior
// record(s) and the N observe statements, then the half-kick(s) on p_k.  Returns "some own force component was non-finite".
// FUSED: -0.5 z z - ln sigma as ONE fma with an exact product (fg_hmc_sep.hip's FG_SEP_LPF: the same bits except for z z in [2^1024, 2^1025), where the
// fused form is -inf and the reference's (-0.5 z) z still finite) -- a non-finite force component in the fused instance sends the wave through the
// unfused one again before any momentum is kicked (q is not written during a gradient, a wave's coordinates are its own): 16 of 453 instructions per
// observation of the eight-coordinate layout.
template <int D, int W, int WV, bool P2, bool FUSED = false>
__device__ __noinline__ bool fg_lin_grad(const double *tab_v, int n_obs_v, const int *meta_v, const FgGradRec *gs_v, const fg_lds_double *slots,
                                         fg_lds_double *pl, double h_v, double hk, int two_kicks_v, int d_v) {
    // a row's coefficients arrive in NC chunks of CS (two SGPR buffers, the next chunk requested while the current one is used);
    // QL (D = 64): q stays in LDS and is read a chunk at a time -- 64 coordinates in registers would be the whole budget
    constexpr int M = D / W, CS = D >= 32 ? 16 : D / 2, NC = D / CS, ROWB = FG_LIN_ROW_DOUBLES(D) * 8, tw = FG_WAVE;
    constexpr bool QL = D > 32;
    constexpr int PLAST = fg_lin_pos(W, WV, M - 1);
    static_assert(NC % 2 == 0, "the next row's first chunk lands in buffer 0");
    const FG_AS4 char *row = fg_uniform_ptr(tab_v);
    const FG_AS4 int *meta = (const FG_AS4 int *)fg_uniform_ptr(meta_v);
    const int N = __builtin_amdgcn_readfirstlane(n_obs_v), d_real = __builtin_amdgcn_readfirstlane(d_v);
    const bool two_kicks = __builtin_amdgcn_readfirstlane(two_kicks_v) != 0;
    const double h = fg_uniform(h_v), two_h = fg_uniform(2.0 * h_v), rcp_2h = fg_uniform(1.0 / (2.0 * h_v));
    double q[QL ? CS : D];
    if (!QL) {
#pragma unroll
        for (int t = 0; t < D; ++t) q[t] = slots[meta[t] * tw];
    }
    double qown[M];                                               // the own coordinates' values (QL: q itself is not resident)
#pragma unroll
    for (int a = 0; a < M; ++a) qown[a] = slots[meta[fg_lin_pos(W, WV, a)] * tw];
    double sp[M], sm[M];
#pragma unroll
    for (int a = 0; a < M; ++a) { sp[a] = 0.0; sm[a] = 0.0; }
    FgLinHalf<CS> cf[2];
    fg_u32x4 ha = *(const FG_AS4 fg_u32x4 *)row;                  // c0
    cf[0].load(row + 16);
    for (int i = 0; i < N; ++i) {
        double S = 0.0;
        double mp[M], mm[M];
        fg_u32x8 hb; fg_u32x4 hf;
#define FG_LIN_TERM(t, CF)                                                                          \
        {                                                                                           \
            const double c_ = (CF);                                                                 \
            const double qt_ = q[QL ? (t) % CS : (t)];                                              \
            const double P_ = qt_ * c_;                                                             \
            _Pragma("unroll") for (int a = 0; a < M; ++a) {                                         \
                if (fg_lin_pos(W, WV, a) < (t)) { mp[a] = mp[a] + P_; mm[a] = mm[a] + P_; }         \
                if (fg_lin_pos(W, WV, a) == (t)) {                                                  \
                    const double qp_ = qt_ + h, qm_ = qt_ - h;            /* the perturbed coordinate holds orig +- h (hmc.rs:317-319) */ \
                    mp[a] = S + qp_ * c_; mm[a] = S + qm_ * c_;                                     \
                }                                                                                   \
            }                                                                                       \
            if ((t) < PLAST) S = S + P_;                                                            \
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            __builtin_amdgcn_s_waitcnt(0xc07f);                   // this chunk (requested a chunk ago)
            if (c == 0) S = fg_dbl(ha[0], ha[1]);
            if (c + 1 < NC) cf[(c + 1) & 1].load(row + 16 + 8 * CS * (c + 1));
            else { ha = *(const FG_AS4 fg_u32x4 *)(row + ROWB); cf[0].load(row + ROWB + 16); }     // the next row's first chunk (one zero row follows the table)
            if (c == 0) {
                hb = *(const FG_AS4 fg_u32x8_r *)(row + 16 + 8 * D);                               // y, 1 / sigma, ln sigma, sigma
                hf = *(const FG_AS4 fg_u32x4 *)(row + 16 + 8 * D + 32);                            // flags
            }
            if (QL) {
                const fg_u32x16 mc = *(const FG_AS4 fg_u32x16 *)((const FG_AS4 char *)meta + 64 * c);   // the coordinates at this chunk's term positions
#pragma unroll
                for (int j = 0; j < CS; ++j) q[j] = slots[mc[j] * tw];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < CS; ++j) FG_LIN_TERM(c * CS + j, cf[c & 1].get(j))
        }
#undef FG_LIN_TERM
        const double y = fg_dbl(hb[0], hb[1]), inv = fg_dbl(hb[2], hb[3]), lns = fg_dbl(hb[4], hb[5]);
        const uint32_t fl = hf[0];
#pragma unroll
        for (int a = 0; a < M; ++a) {
            const double dlp = y - mp[a], dlm = y - mm[a];
            double zp = dlp * inv, zm = dlm * inv;                // exact quotient when sigma = 2^k
            if (!P2 && !(fl & FG_G_POW2)) {                      // (x - mu) / sigma, distribution.rs:205
                const double sg = fg_dbl(hb[6], hb[7]);
                if (fl & FG_G_DIV) { zp = dlp / sg; zm = dlm / sg; }
                else { zp = fg_div_const(dlp, sg, inv); zm = fg_div_const(dlm, sg, inv); }
            }
            const double lpp = FUSED ? __builtin_fma(-0.5, zp * zp, -lns) - 0.5 * FG_LN_2PI : -0.5 * zp * zp - lns - 0.5 * FG_LN_2PI;     // distribution.rs:207
            const double lpm = FUSED ? __builtin_fma(-0.5, zm * zm, -lns) - 0.5 * FG_LN_2PI : -0.5 * zm * zm - lns - 0.5 * FG_LN_2PI;
            sp[a] += lpp; sm[a] += lpm;
        }
        row += ROWB;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // log_prior at q +- h e_k (the coordinate's own record(s), before the observe records in the stream), total_log_weight,
    // the central difference and the kick(s): fg_grec_math's FG_G_END
    const FG_AS4 char *gs = fg_uniform_ptr(gs_v);
    bool bad = false;
    double gk[M];
#pragma unroll
    for (int a = 0; a < M; ++a) {
        const int pos = fg_lin_pos(W, WV, a);
        gk[a] = 0.0;
        if (pos >= d_real) continue;                                         // a padded term position (d < D): no coordinate
        const int k = meta[pos], r0 = meta[D + 2 * k], nr = meta[D + 2 * k + 1];
        double prip = 0.0, prim = 0.0;
        for (int j = 0; j < nr; ++j) {
            const fg_u32x16 r = *(const FG_AS4 fg_u32x16 *)(gs + 64 * (long long)(r0 + j));
            double lpp, lpm;
            fg_lin_prior_pair(r, qown[a], h, lpp, lpm);
            prip += lpp; prim += lpm;
        }
        const double tp = prip + sp[a], tm = prim + sm[a];                   // total_log_weight (log_factors = +0.0 adds nothing)
        const double n = tp - tm;
        double g = fg_div_const(n, two_h, rcp_2h);                           // (lp - lm) / (2h), hmc.rs:322
        const uint32_t ne = (uint32_t)(__double_as_longlong(n) >> 32) & 0x7fffffffu;
        if (__builtin_expect(__any(!(n == 0.0 || (ne - 0x0c800000u) < 0x6f000000u)), 0)) g = n / two_h;   // |n| outside [2^-823, 2^953]
        bad = bad || !fg_finite(g);
        gk[a] = g;
    }
    if (FUSED) { if (__builtin_expect(__any(bad), 0)) return fg_lin_grad<D, W, WV, P2, false>(tab_v, n_obs_v, meta_v, gs_v, slots, pl, h_v, hk, two_kicks_v, d_v); }
#pragma unroll
    for (int a = 0; a < M; ++a) {
        const int pos = fg_lin_pos(W, WV, a);
        if (pos >= d_real) continue;
        const int k = meta[pos];
        double p = pl[k * tw] + hk * gk[a];                                  // hmc.rs:389 / :400
        if (two_kicks) p += hk * gk[a];
        pl[k * tw] = p;
    }
    return bad;
}

// The same gradient on a HALF TILE: 32 chains per workgroup, lanes l and l + 32 carry the same chain -- the lower half the
// sums at q + h e_k, the upper half those at q - h e_k (one suffix chain, one density and one accumulator per own coordinate
// and lane instead of two: a - b and a + (-b) are the same operation, so every number is the one the full tile forms).  The
// two totals meet in one cross-lane exchange per coordinate and gradient.  Products and prefix sums are formed by both halves,
// so a wave issues ~0.64 of the full tile's instructions for half the chains: worth it exactly when 64-chain tiles would leave
// CUs idle (8 192 chains = 128 tiles on 256 CUs: BASELINE's 8-GPU sharding of C3).
template <int D, int W, int WV, bool P2, bool FUSED = false>
__device__ __noinline__ bool fg_lin_grad_half(const double *tab_v, int n_obs_v, const int *meta_v, const FgGradRec *gs_v, const fg_lds_double *slots,
                                              fg_lds_double *pl, double h_v, double hk, int two_kicks_v, int d_v) {
    constexpr int M = D / W, HB = D / 2, ROWB = FG_LIN_ROW_DOUBLES(D) * 8, tw = FG_WAVE / 2;
    constexpr int PLAST = fg_lin_pos(W, WV, M - 1);
    const FG_AS4 char *row = fg_uniform_ptr(tab_v);
    const FG_AS4 int *meta = (const FG_AS4 int *)fg_uniform_ptr(meta_v);
    const int N = __builtin_amdgcn_readfirstlane(n_obs_v), d_real = __builtin_amdgcn_readfirstlane(d_v);
    const bool two_kicks = __builtin_amdgcn_readfirstlane(two_kicks_v) != 0;
    const double h = fg_uniform(h_v), two_h = fg_uniform(2.0 * h_v), rcp_2h = fg_uniform(1.0 / (2.0 * h_v));
    const bool upper = (threadIdx.x & 32u) != 0u;
    const double hs = upper ? -h : h;                             // this lane's perturbation
    double q[D];
#pragma unroll
    for (int t = 0; t < D; ++t) q[t] = slots[meta[t] * tw];
    double sa[M];
#pragma unroll
    for (int a = 0; a < M; ++a) sa[a] = 0.0;
    FgLinHalf<HB> ca, cb;
    fg_u32x4 ha = *(const FG_AS4 fg_u32x4 *)row;                  // c0
    ca.load(row + 16);
    for (int i = 0; i < N; ++i) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        cb.load(row + 16 + 8 * HB);
        const fg_u32x8 hb = *(const FG_AS4 fg_u32x8_r *)(row + 16 + 16 * HB);
        const fg_u32x4 hf = *(const FG_AS4 fg_u32x4 *)(row + 16 + 16 * HB + 32);
        __builtin_amdgcn_sched_barrier(0);
        double S = fg_dbl(ha[0], ha[1]);
        double mu[M];
#define FG_LIN_TERM(t, CF)                                                                          \
        {                                                                                           \
            const double c_ = (CF);                                                                 \
            const double P_ = q[t] * c_;                                                            \
            _Pragma("unroll") for (int a = 0; a < M; ++a) {                                         \
                if (fg_lin_pos(W, WV, a) < (t)) mu[a] = mu[a] + P_;                                 \
                if (fg_lin_pos(W, WV, a) == (t)) { const double qs_ = q[t] + hs; mu[a] = S + qs_ * c_; }   \
            }                                                                                       \
            if ((t) < PLAST) S = S + P_;                                                            \
        }
#pragma unroll
        for (int t = 0; t < HB; ++t) FG_LIN_TERM(t, ca.get(t))
        __builtin_amdgcn_s_waitcnt(0xc07f);
        ha = *(const FG_AS4 fg_u32x4 *)(row + ROWB);
        ca.load(row + ROWB + 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = HB; t < D; ++t) FG_LIN_TERM(t, cb.get(t - HB))
#undef FG_LIN_TERM
        const double y = fg_dbl(hb[0], hb[1]), inv = fg_dbl(hb[2], hb[3]), lns = fg_dbl(hb[4], hb[5]);
        const uint32_t fl = hf[0];
#pragma unroll
        for (int a = 0; a < M; ++a) {
            const double dl = y - mu[a];
            double z = dl * inv;
            if (!P2 && !(fl & FG_G_POW2)) {
                const double sg = fg_dbl(hb[6], hb[7]);
                z = (fl & FG_G_DIV) ? dl / sg : fg_div_const(dl, sg, inv);
            }
            sa[a] += FUSED ? __builtin_fma(-0.5, z * z, -lns) - 0.5 * FG_LN_2PI : -0.5 * z * z - lns - 0.5 * FG_LN_2PI;
        }
        row += ROWB;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const FG_AS4 char *gs = fg_uniform_ptr(gs_v);
    bool bad = false;
    double gk[M];
#pragma unroll
    for (int a = 0; a < M; ++a) {
        const int pos = fg_lin_pos(W, WV, a);
        gk[a] = 0.0;
        if (pos >= d_real) continue;                                         // a padded term position (d < D): no coordinate
        const int k = meta[pos], r0 = meta[D + 2 * k], nr = meta[D + 2 * k + 1];
        double pri = 0.0;
        for (int j = 0; j < nr; ++j) {
            const fg_u32x16 r = *(const FG_AS4 fg_u32x16 *)(gs + 64 * (long long)(r0 + j));
            double lpp, lpm;
            fg_lin_prior_pair(r, q[pos], h, lpp, lpm);
            pri += upper ? lpm : lpp;
        }
        const double t_own = pri + sa[a];                                    // total_log_weight at q + hs e_k
        const double t_oth = __shfl_xor(t_own, 32, 64);
        const double n = upper ? t_oth - t_own : t_own - t_oth;              // lp(q + h e_k) - lp(q - h e_k), the same subtraction in both halves
        double g = fg_div_const(n, two_h, rcp_2h);                           // (lp - lm) / (2h), hmc.rs:322
        const uint32_t ne = (uint32_t)(__double_as_longlong(n) >> 32) & 0x7fffffffu;
        if (__builtin_expect(__any(!(n == 0.0 || (ne - 0x0c800000u) < 0x6f000000u)), 0)) g = n / two_h;   // |n| outside [2^-823, 2^953]
        bad = bad || !fg_finite(g);
        gk[a] = g;
    }
    if (FUSED) { if (__builtin_expect(__any(bad), 0)) return fg_lin_grad_half<D, W, WV, P2, false>(tab_v, n_obs_v, meta_v, gs_v, slots, pl, h_v, hk, two_kicks_v, d_v); }
#pragma unroll
    for (int a = 0; a < M; ++a) {
        const int pos = fg_lin_pos(W, WV, a);
        if (pos >= d_real) continue;
        const int k = meta[pos];
        double p = pl[k * tw] + hk * gk[a];                                  // hmc.rs:389 / :400 (both halves: the same value to the same cell)
        if (two_kicks) p += hk * gk[a];
        pl[k * tw] = p;
    }
    return bad;
}

template <int D, int W, bool P2, bool HALF, int WV>
struct FgLinDispatch {
    static __device__ __forceinline__ bool run(int wv, const double *tab, int n_obs, const int *meta, const FgGradRec *gs, const fg_lds_double *slots,
                                               fg_lds_double *pl, double h, double hk, int two_kicks, int d) {
        if (wv == WV) {
            constexpr bool FUSED = D / W == 8;                    // (the default layouts of D >= 32; the others keep the unfused form only)
            if constexpr (HALF) return fg_lin_grad_half<D, W, WV, P2, FUSED>(tab, n_obs, meta, gs, slots, pl, h, hk, two_kicks, d);
            else return fg_lin_grad<D, W, WV, P2, FUSED>(tab, n_obs, meta, gs, slots, pl, h, hk, two_kicks, d);
        }
        return FgLinDispatch<D, W, P2, HALF, WV + 1>::run(wv, tab, n_obs, meta, gs, slots, pl, h, hk, two_kicks, d);
    }
};
template <int D, int W, bool P2, bool HALF>
struct FgLinDispatch<D, W, P2, HALF, W> {
    static __device__ __forceinline__ bool run(int, const double *, int, const int *, const FgGradRec *, const fg_lds_double *, fg_lds_double *, double, double, int, int) { return false; }
};

// HALF: 32 chains per workgroup (fg_lin_grad_half); everything outside the gradient runs in both lane halves on the same chain
// (same values to the same LDS cells), only the lower half stores to global memory; the halves share the momentum pairs.
template <int D, int W, bool P2, bool HALF>
__global__ __launch_bounds__(FG_WAVE * W) __attribute__((amdgpu_waves_per_eu(D / W >= 8 ? 2 : 4, D / W >= 8 ? 2 : 4)))
void k_hmc_lin_steps(FgProgramDev P, FgChainCtx X, FgHmcDev H, int iter0, int n_steps, int n_warmup, int welford_on, double *draws, int first_sample_t,
                     double *pos_all /*[n][d][C] or null*/, double *info /*[n][4][C] or null*/) {
    extern __shared__ double lds[];
    constexpr int tw = HALF ? FG_WAVE / 2 : FG_WAVE, M = D / W;
    const int lane = threadIdx.x & (tw - 1);                       // chain of the tile
    const int half = HALF ? (int)((threadIdx.x >> 5) & 1u) : 0;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long chain = (long long)blockIdx.x * tw + lane;
    const bool live = chain < X.C && half == 0;
    const long long c = chain < X.C ? chain : X.C - 1;
    double *slots = lds + lane;
    double *pl = lds + (long long)P.n_slots * tw + lane;
    const int d = P.d;                                              // the coordinates (D = d padded to the next built size: 8 / 16 / 32 term positions)
    double *xch = lds + (long long)(P.n_slots + d) * tw + lane;      // rows: 0 step size, 1 accepted, 2.. per-wave divergence flags
    const int L = H.L;
    const double *mi = H.use_mass ? H.m_inv + c : nullptr;
    const double *ms = H.use_mass ? H.mass_sqrt + c : nullptr;
    const uint32_t sk0 = (uint32_t)X.seed, sk1 = (uint32_t)(X.seed >> 32), gchain = X.chain0 + (uint32_t)c;
    int own[M];                                                   // the own coordinates (wave-uniform)
#pragma unroll
    for (int a = 0; a < M; ++a) own[a] = fg_lin_pos(W, wv, a) < d ? P.lin_meta[fg_lin_pos(W, wv, a)] : -1;      // (-1: a padded position)
    // wave 0 owns the per-chain sampler state
    double lj = 0.0, eps = 0.0, frozen = 0.0, da_mu = 0.0, da_leb = 0.0, da_hbar = 0.0, asum = 0.0;
    unsigned long long da_m = 0, ndiv = 0;
    if (wv == 0) {
        fg_load_values(P, X, c, slots, tw);
        lj = H.lj[c]; eps = H.eps[c]; frozen = H.frozen[c];
        da_mu = H.da_mu[c]; da_leb = H.da_leb[c]; da_hbar = H.da_hbar[c]; da_m = H.da_m[c];
    }
    for (int t = 0; t < n_steps; ++t) {
        const int iter = iter0 + t;
        const bool warming = iter < n_warmup;
        double h0 = 0.0, u = 0.0;
        // p0 ~ N(0, M) (hmc.rs:436-441): Box-Muller pair j of the chain's (iteration) Philox stream is block j
        const int n_pairs = (d + 1) >> 1;
        for (int j = HALF ? 2 * wv + half : wv; j < n_pairs; j += HALF ? 2 * W : W) {
            const FgD2 zz = fg_cold_normal_pair(sk0, sk1, gchain, (uint32_t)j, (uint32_t)iter, FG_RNG_HMC);
            const int i = 2 * j;
            pl[i * tw] = zz.a * (ms ? ms[(long long)i * X.C] : 1.0);
            if (i + 1 < d) pl[(i + 1) * tw] = zz.b * (ms ? ms[(long long)(i + 1) * X.C] : 1.0);
        }
        if (wv == 0) {
            double e;
            if (warming) e = eps;
            else {                                             // frozen_or_current: hmc.rs:789-798
                if (frozen == frozen) e = frozen;
                else if (n_warmup > 0) e = fg_cold_exp(da_leb);
                else e = eps;
                frozen = e;
            }
            u = fg_cold_u01_pair(sk0, sk1, gchain, (uint32_t)n_pairs, (uint32_t)iter, FG_RNG_HMC).a;
            xch[0] = e;
        }
        __syncthreads();
        if (wv == 0) h0 = -lj + fg_kinetic(P, pl, tw, mi, X.C);  // hmc.rs:442-443 (all of p0, before any kick)
        __syncthreads();
        const double e = xch[0], hk = 0.5 * e;
        bool bad = false;
        for (int gs = 0; gs <= L; ++gs) {                       // leapfrog, hmc.rs:353-407
            bad = FgLinDispatch<D, W, P2, HALF, 0>::run(wv, P.lin_tab, P.lin_n, P.lin_meta, P.gstream, (const fg_lds_double *)slots, (fg_lds_double *)pl, H.h, hk,
                                                  (gs > 0 && gs < L) ? 1 : 0, d) || bad;
            __syncthreads();                                     // every p kicked, every read of q done
            if (gs < L) {
#pragma unroll
                for (int a = 0; a < M; ++a) {
                    const int k = own[a];
                    if (k < 0) continue;
                    if (mi) slots[k * tw] += e * mi[(long long)k * X.C] * pl[k * tw];
                    else slots[k * tw] += e * pl[k * tw];        // eps * 1.0 * p == eps * p
                }
                __syncthreads();
            }
        }
        xch[(2 + wv) * tw] = bad ? 1.0 : 0.0;
        __syncthreads();
        if (wv == 0) {
            bool div = false;
            for (int w = 0; w < W; ++w) div = div || xch[(2 + w) * tw] != 0.0;
            FgAcc3 A = {0.0, 0.0, 0.0};
            fg_score_stream<1>(P.sstream, P.n_sstream, P.pool, slots, tw, A);                                       // score_full, hmc.rs:283-299
            const double lj_new = fg_total(A);
            div = div || !fg_finite(lj_new);
            double ap = 0.0; bool acc = false;
            if (!div) {
                const double h_new = -lj_new + fg_kinetic(P, pl, tw, mi, X.C);
                ap = fg_cold_accept_prob(h0, h_new);             // hmc.rs:460
                acc = u < ap;                                    // hmc.rs:461
            }
            if (acc) lj = lj_new;
            xch[tw] = acc ? 1.0 : 0.0;
            asum += ap; ndiv += div ? 1ull : 0ull;
            if (live && info) {                                  // HmcStepInfo: hmc.rs:587-602
                double *r = info + (long long)t * 4 * X.C + c;
                r[0] = acc ? 1.0 : 0.0; r[X.C] = div ? 1.0 : 0.0; r[2 * X.C] = ap; r[3 * X.C] = e;
            }
            if (warming) {                                       // DualAveraging::update: hmc.rs:168-178
                da_m += 1ull;
                const FgD3 r = fg_cold_da_update(da_hbar, da_leb, (double)da_m, da_mu, H.target, ap);
                eps = r.a; da_hbar = r.b; da_leb = r.c;
            }
        }
        __syncthreads();
        const bool acc = xch[tw] != 0.0;
        unsigned long long wn = 0;
        if (warming && welford_on) wn = H.w_n[c] + 1ull;          // every wave reads the old count before wave 0 bumps it below
#pragma unroll
        for (int a = 0; a < M; ++a) {                             // commit or roll back this wave's f64 sites
            const int i = own[a];
            if (i < 0) continue;
            const long long g = (long long)P.f64_site[i] * X.C + c;
            if (acc) { if (live) X.values[g] = fg_as_i64(slots[i * tw]); }
            else slots[i * tw] = fg_as_double(X.values[g]);
            const double x = slots[i * tw];
            if (live && pos_all) pos_all[((long long)t * d + i) * X.C + c] = x;
            if (warming) {
                if (welford_on) {                                 // Welford::push: hmc.rs:202-211
                    const long long gi = (long long)i * X.C + c;
                    const double n = (double)wn;
                    double mean = H.w_mean[gi];
                    const double delta = x - mean;
                    mean += delta / n;
                    const double delta2 = x - mean;
                    if (live) { H.w_mean[gi] = mean; H.w_m2[gi] += delta * delta2; }
                }
            } else if (draws && live) draws[((long long)(t - first_sample_t) * d + i) * X.C + c] = x;   // hmc.rs:577-582
        }
        if (warming && welford_on) {
            __syncthreads();                                      // all waves hold the old count
            if (wv == 0 && live) H.w_n[c] = wn;
        }
    }
    if (wv == 0 && live) {
        H.lj[c] = lj; H.eps[c] = eps; H.frozen[c] = frozen;
        H.da_mu[c] = da_mu; H.da_leb[c] = da_leb; H.da_hbar[c] = da_hbar; H.da_m[c] = da_m;
        H.alpha_sum[c] += asum; H.n_div[c] += ndiv;
    }
}

// Launch for `n` transitions from iteration `iter0`; FG_E_UNSUPPORTED when the program / configuration is not a dense
// regression in the sparse finite-difference mode (the caller then takes the gradient-stream kernel).
int fg_hmc_lin_launch(fg_engine *e, int iter0, int n, int welford_on, double *draws, int first_sample_t, double *pos_all, double *info) {
    if (e->gt) return FG_E_UNSUPPORTED;                       // tiles in global memory: the one-wave-per-tile kernels (fg_engine.hip)
    if (!e->P.lin_tab || e->cfg.grad_mode != FG_GRAD_FD_SPARSE || e->lin_disabled || e->tw != FG_WAVE) return FG_E_UNSUPPORTED;
    if (e->d < 2 || e->d > 64) return FG_E_UNSUPPORTED;      // (the table exists for d <= 64: fg_program.cpp)
    const int D = e->d <= 8 ? 8 : (e->d <= 16 ? 16 : (e->d <= 32 ? 32 : 64));     // term positions of the build that takes it (padded with +0.0 terms)
    const long long n_cu = std::max(1, e->n_simd / 4);
    const long long tiles64 = (e->C + FG_WAVE - 1) / FG_WAVE;
    // half tiles (32 chains per workgroup) when 64-chain tiles would leave half of the CUs without one
    bool half = 2 * tiles64 <= n_cu;
    if (const char *hv = std::getenv("FG_HMC_LIN_HALF")) half = std::atoi(hv) != 0;
    if (D == 64) half = false;                               // (64 positions: 16 waves of four, q read from LDS; full tiles only)
    const int tw = half ? FG_WAVE / 2 : FG_WAVE;
    const unsigned tiles = (unsigned)((e->C + tw - 1) / tw);
    const size_t lds = (size_t)(e->n_slots + e->d + 2 + FG_LIN_WMAX) * tw * sizeof(double);
    if (lds > 160 * 1024) return FG_E_UNSUPPORTED;
    // waves per tile.  Every wave of a tile forms all D products and the shared prefix sums again, so fewer waves with more coordinates
    // each execute fewer instructions per observation: W (2 D + 16 M) + D (D + 1) for M = D / W coordinates per wave.
    //   M = 8 (D / 8 waves; 256 VGPRs, two waves per SIMD): the default for D >= 32 and for full tiles of D = 16 -- C3 at 65 536 chains 1.55e7 -> 1.69e7
    //         leapfrog-steps/s (1.77e7 with the one-fma density of fg_lin_grad's FUSED instances), half tiles at 8 192 chains 9.2e6 -> 1.17e7 (one wave per
    //         SIMD: a wave's 8 or 16 suffix sums are independent, they fill the issue slots), 64 coefficients 3.6e6 -> 4.2e6, 16 coefficients 4.8e7 -> 5.1e7
    //         (but half tiles of D = 16, two waves per tile: 2.2e7 -> 1.8e7) (profiles/round4_lin_eight_per_wave.txt);
    //   M = 4 (D / 4 waves; 128 VGPRs): round 3's layout, the default for D = 8 and for half tiles of D = 16; M = 2 (D / 2 waves) measured slower at every
    //         chain count; M = 16 (two waves of 256 VGPRs + 184 spilled: one wave per SIMD) loses a third.
    // FG_HMC_WAVES = D / 8, D / 4, D / 2 picks one for the bit-identity tests.
    int W = (D >= 32 || (D == 16 && !half)) ? D / 8 : D / 4;
    if (e->mw_override == D / 4 || (D < 64 && e->mw_override == D / 2) || (D >= 16 && e->mw_override == D / 8)) W = e->mw_override;
    const bool p2 = e->P.lin_p2 != 0;
#define FG_LIN_KERNELS_D(X, DD, HH) X(DD, DD / 8, false, HH) X(DD, DD / 8, true, HH) X(DD, DD / 4, false, HH) X(DD, DD / 4, true, HH) X(DD, DD / 2, false, HH) X(DD, DD / 2, true, HH)
#define FG_LIN_KERNELS(X) FG_LIN_KERNELS_D(X, 32, false) FG_LIN_KERNELS_D(X, 32, true) FG_LIN_KERNELS_D(X, 16, false) FG_LIN_KERNELS_D(X, 16, true)      \
                          X(8, 2, false, false) X(8, 2, true, false) X(8, 4, false, false) X(8, 4, true, false) X(8, 2, false, true) X(8, 2, true, true)    \
                          X(8, 4, false, true) X(8, 4, true, true) X(64, 8, false, false) X(64, 8, true, false) X(64, 16, false, false) X(64, 16, true, false)
    const void *fn = nullptr;
    int variant = -1, idx = 0;
#define FG_LIN_FN(DD, WW, PP, HH) if (D == DD && W == WW && p2 == PP && half == HH) { fn = (const void *)k_hmc_lin_steps<DD, WW, PP, HH>; variant = idx; } ++idx;
    FG_LIN_KERNELS(FG_LIN_FN)
#undef FG_LIN_FN
    if (!fn) return FG_E_UNSUPPORTED;
    static bool attr_set_dev[64][64];
    bool &attr_set = attr_set_dev[e->device & 63][variant];
    if (!attr_set && lds > 64 * 1024) {
        const hipError_t he = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (he != hipSuccess) { fg_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(he)); return FG_E_HIP; }
        attr_set = true;
    }
#define FG_LIN_GO(DD, WW, PP, HH) if (D == DD && W == WW && p2 == PP && half == HH) hipLaunchKernelGGL((k_hmc_lin_steps<DD, WW, PP, HH>), dim3(tiles), dim3(FG_WAVE * WW), lds, e->stream, e->P, e->X, e->H, \
                                                                                                        iter0, n, e->n_warmup, welford_on, draws, first_sample_t, pos_all, info);
    FG_LIN_KERNELS(FG_LIN_GO)
#undef FG_LIN_GO
#undef FG_LIN_KERNELS
#undef FG_LIN_KERNELS_D
    HIPCHK(hipGetLastError());
    e->last_hmc_kernel = std::string(half ? "k_hmc_lin_steps (half tiles) W=" : "k_hmc_lin_steps W=") + std::to_string(W);
    return FG_OK;
}
