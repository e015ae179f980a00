// fg_hmc_sep.hip -- HmcSession::step x n (hmc.rs:819-919) for INDEPENDENT-SITES programs: every force term is a Normal with
// constant sigma that reads ONE coordinate and constants (the north-star model: x#i ~ N(0,1); y#i ~ N(x#i, 0.5)).
//
// Inside a trajectory such a coordinate never sees another one: its L leapfrog steps (hmc.rs:353-407) are a closed
// recurrence on (q_i, p_i).  So a wave runs the WHOLE trajectory of a coordinate with q_i, p_i in registers and the
// coordinate's <= FG_SEP_MAXREC records in SGPRs (loaded once per trajectory, not once per gradient): no LDS access, no
// record fetch, no address arithmetic inside the L-step loop -- just the f64 instructions of the reference's arithmetic
// (the same additions and multiplications in the same order as fg_grad_stream / two sparse evaluations of log_prob, hence
// bit-identical draws: tests/test_gpu_parity.py::test_hmc_sep_kernel_is_bit_identical) and a handful of scalar branches.
// A 64-chain tile is shared by W = 1..16 waves, wave w owning an even-aligned run of coordinates (Box-Muller pairs are not
// split); with few tiles per CU (small chain counts) W grows so that every SIMD still holds 4 waves.
//
// Per transition (two workgroup barriers): every wave, for each own coordinate: draw p0, emit its kinetic term, run the
// trajectory in registers, emit the endpoint's kinetic term and the endpoint-score TERMS of the coordinate's statements
// (LDS rows) | barrier | wave 0 adds the terms IN ORDER -- kinetic terms in coordinate order (hmc.rs:442-443), log_prior and
// log_likelihood terms in program order (score_full, hmc.rs:283-299): the same sums in the same order as one wave doing all
// the arithmetic -- then accept and dual averaging (cold code, out of line) | barrier | commit / roll back the own coordinates.
#include "fg_engine_internal.h"
#include "fg_gradstream.h"
#include "fg_cold.h"

#define FG_SEP_WMAX 16

// FG_HMC_PROF (experiment builds, tools/prof_hmc_phases.py): cycles tile 0's waves spend in each part of a transition
#ifdef FG_HMC_PROF
__device__ unsigned long long fg_hmc_prof[FG_SEP_WMAX][8];
#define FG_PROF_T(i) { const unsigned long long now_ = __builtin_readcyclecounter(); prof_[i] += now_ - tprev_; tprev_ = now_; }
#else
#define FG_PROF_T(i)
#endif
struct FgSegSep { int c[FG_SEP_WMAX + 1]; int sum4;       // sum4: the four in-order sums of a transition's end on four waves (tiles that are alone on their CU)
                  int own[FG_SEP_WMAX][4], n_own[FG_SEP_WMAX]; };   // MODE 3 (dense, coordinates in registers): each wave's <= 4 coordinates (whole Box-Muller pairs), ascending
#ifndef FG_SEP_STAGGER
#define FG_SEP_STAGGER 2          /* x 4 096 cycles: the late start of a CU's second tile (k_hmc_sep_steps) */
#endif

// a coordinate's records: by scalar loads from a wave-uniform pointer (SGPRs), or -- half tiles, where the two lane halves of a
// wave run different coordinates -- by per-lane loads (VGPRs)
__device__ __forceinline__ fg_u32x8 fg_sep_ld8(const FG_AS4 char *p) { return *(const FG_AS4 fg_u32x8 *)p; }
__device__ __forceinline__ fg_u32x4 fg_sep_ld4(const FG_AS4 char *p) { return *(const FG_AS4 fg_u32x4 *)p; }
__device__ __forceinline__ fg_u32x8 fg_sep_ld8(const char *p) { return *(const fg_u32x8 *)p; }
__device__ __forceinline__ fg_u32x4 fg_sep_ld4(const char *p) { return *(const fg_u32x4 *)p; }
#define FG_SEP_LOAD(k) \
    const fg_u32x8 a##k = fg_sep_ld8(rb + 64 * k); \
    const fg_u32x4 b##k = fg_sep_ld4(rb + 64 * k + 32);

// (x - mu) / sigma of record k for the operand difference dl = q - c (x - mu is +dl or -dl: the quotient's sign flips with
// it exactly -- multiplication, the FMA sequence of fg_div_const and IEEE division are odd functions under round-to-nearest
// -- and z only enters the density as z * z)
#define FG_SEP_Z(k, dl) (P2 ? (dl) * fg_dbl(a##k[4], a##k[5])                                                                   \
                            : ((a##k[0] & FG_G_POW2) ? (dl) * fg_dbl(a##k[4], a##k[5])                                           \
                                                     : ((a##k[0] & FG_G_DIV) ? (dl) / fg_dbl(b##k[0], b##k[1])                    \
                                                                             : fg_div_const((dl), fg_dbl(b##k[0], b##k[1]), fg_dbl(a##k[4], a##k[5])))))
// Normal log-density of record k at z: distribution.rs:207
#define FG_SEP_LP(k, z) (-0.5 * (z) * (z) - fg_dbl(a##k[6], a##k[7]) - 0.5 * FG_LN_2PI)
// record k at q + h and q - h
#define FG_SEP_DUAL(k, outp, outm)                                                                     \
    double outp, outm;                                                                                 \
    { const double c = fg_dbl(a##k[2], a##k[3]); const double zp = FG_SEP_Z(k, qp - c), zm = FG_SEP_Z(k, qm - c); \
      outp = FG_SEP_LP(k, zp); outm = FG_SEP_LP(k, zm); }
// record k at q (endpoint score term; NaN z -> -inf like FG_OP_NORMAL_FAST)
#define FG_SEP_TERM_TO(k, dst)                                                                         \
    { const double z = FG_SEP_Z(k, q - fg_dbl(a##k[2], a##k[3])); const double lp = FG_SEP_LP(k, z); \
      (dst)[a##k[1] * tw] = (z != z) ? FG_NEG_INF : lp; }
#define FG_SEP_TERM(k) FG_SEP_TERM_TO(k, terms)
// record 0 when the coordinate's own sample statement is Normal(0, 1) (U0): q - 0, * 1 and - ln 1 change no bit of any double
// (x - 0.0 = x, x * 1.0 = x for every x incl. -0.0, inf, NaN), so they are not issued: 6 of the 42 instructions of a coordinate-step
#define FG_SEP_LP_U(z) (-0.5 * (z) * (z) - 0.5 * FG_LN_2PI)
// The same values in one instruction less (hot loop only).  -0.5 * z is exact, so RN(RN(-0.5 z) z) = -0.5 RN(z z) and the
// reference's RN(RN(-0.5 z z) - ln sigma) is ONE correctly rounded -0.5 m - ln sigma of m = RN(z z): an fma whose product is
// exact.  Two corners differ before the last subtraction and not after it: m subnormal (halving rounds; the difference is
// below 2^-1074 next to a 0.9189 that swallows it) -- and z z in [2^1024, 2^1025), where the reference's (-0.5 z) z is still
// finite and this is -inf: the force is then non-finite, so is the endpoint momentum, and the coordinate takes the checked
// instance (the unfused arithmetic) again.  The endpoint's score term is always the unfused form.
#define FG_SEP_LPF(k, z) (__builtin_fma(-0.5, (z) * (z), -fg_dbl(a##k[6], a##k[7])) - 0.5 * FG_LN_2PI)
#define FG_SEP_LPF_U(z) __builtin_fma(-0.5, (z) * (z), -0.5 * FG_LN_2PI)
#define FG_SEP_DUALF(k, outp, outm)                                                                    \
    double outp, outm;                                                                                 \
    { const double c = fg_dbl(a##k[2], a##k[3]); const double zp = FG_SEP_Z(k, qp - c), zm = FG_SEP_Z(k, qm - c); \
      outp = FG_SEP_LPF(k, zp); outm = FG_SEP_LPF(k, zm); }
// record k at q + h and q - h with the guard of a scoring run (the dense mode adds these into whole log-joints)
#define FG_SEPD_OWN(k, outp, outm) double outp, outm;                                                  \
    { const double c_ = fg_dbl(a##k[2], a##k[3]); const double zp_ = FG_SEP_Z(k, qp - c_), zm_ = FG_SEP_Z(k, qm - c_); \
      const double lp_ = FG_SEP_LP(k, zp_), lm_ = FG_SEP_LP(k, zm_);                                    \
      outp = (zp_ != zp_) ? FG_NEG_INF : lp_; outm = (zm_ != zm_) ? FG_NEG_INF : lm_; }

// The whole trajectory of one coordinate: (q, p) -> (q', p') after L leapfrog steps (hmc.rs:353-407) with step size e, then
// the endpoint-score terms of its statements.  NOBS observe records follow the coordinate's own sample record; P2: every
// sigma is a power of two (no quotient branch at all).  The sums are fg_grec_math's: log_prior = the sample record,
// log_likelihood = the observe records in order, total = prior + likelihood (trace.rs:198-200), g = (lp - lm) / (2h).
// emi = e * m_inv_i (hmc.rs:391-393: eps * m_inv[i] * p[i], left to right) or e.  Returns "a force component was non-finite".
// CHECK = false leaves the per-step "force component non-finite" test (hmc.rs:395-399) to the caller: a non-finite g makes
// hk * g non-finite, p only ever changes by adding kicks, and inf / NaN never add back to a finite number -- so a finite
// endpoint p proves every g was finite; a non-finite one sends the coordinate through the CHECK = true instance again.
// The uniform conditions on gs are real scalar branches (the empty asm keeps the compiler from turning them into selects:
// two v_cndmask per f64 and a compare in a loop that is bound by VALU issue).
// AN = FG_GRAD_ANALYTIC: g_i = sum over the coordinate's records of d lp / d q_i = -(q - c) / sigma^2, one evaluation per record,
// the additions of fg_grec_math's analytic branch in the same order ((x - mu) is +-(q - c) and its coefficient -+1: the same bits).
template <int NOBS, bool P2, bool CHECK, bool AN = false, bool U0 = false, typename RB = const FG_AS4 char *>
__device__ __forceinline__ bool fg_sep_trajectory(RB rb, double &q_io, double &p_io, double emi, double hk, int L, double h, double two_h,
                                                  double rcp_2h, double *terms, int tw, int nobs_rt) {
#define FG_SEP_HAS(k) (NOBS >= 0 ? NOBS >= (k) : nobs_rt >= (k))
    FG_SEP_LOAD(0) FG_SEP_LOAD(1) FG_SEP_LOAD(2) FG_SEP_LOAD(3)
    double q = q_io, p = p_io;
    bool bad = false;
    for (int gs = 0; gs <= L; ++gs) {
        double g;
        if (AN) {
#define FG_SEP_ANTERM(k) { const double dl = q - fg_dbl(a##k[2], a##k[3]);                                                   \
        const double w = (P2 || (a##k[0] & FG_G_POW2)) ? (dl * fg_dbl(a##k[4], a##k[5])) * fg_dbl(a##k[4], a##k[5])         \
                                                       : (dl / fg_dbl(b##k[0], b##k[1])) / fg_dbl(b##k[0], b##k[1]); g = g - w; }
            g = 0.0;
            FG_SEP_ANTERM(0)
            if (FG_SEP_HAS(1)) FG_SEP_ANTERM(1)
            if (FG_SEP_HAS(2)) FG_SEP_ANTERM(2)
            if (FG_SEP_HAS(3)) FG_SEP_ANTERM(3)
#undef FG_SEP_ANTERM
        } else {
        const double qp = q + h, qm = q - h;                     // the perturbed coordinate holds orig +- h (hmc.rs:317-319)
        double tp, tm;
        if (!CHECK) {                                            // fused forms: a non-finite force re-runs the coordinate with CHECK
            if (U0) { tp = FG_SEP_LPF_U(qp); tm = FG_SEP_LPF_U(qm); }
            else { FG_SEP_DUALF(0, tp_, tm_) tp = tp_; tm = tm_; }
            if (FG_SEP_HAS(1)) {
                FG_SEP_DUALF(1, lp1, lm1)
                double sp = lp1, sm = lm1;
                if (FG_SEP_HAS(2)) { FG_SEP_DUALF(2, lp2, lm2) sp += lp2; sm += lm2; }
                if (FG_SEP_HAS(3)) { FG_SEP_DUALF(3, lp3, lm3) sp += lp3; sm += lm3; }
                tp = tp + sp; tm = tm + sm;                      // log_prior + log_likelihood
            }
        } else {
        if (U0) { tp = FG_SEP_LP_U(qp); tm = FG_SEP_LP_U(qm); }
        else { FG_SEP_DUAL(0, tp_, tm_) tp = tp_; tm = tm_; }
        if (FG_SEP_HAS(1)) {
            FG_SEP_DUAL(1, lp1, lm1)
            double sp = lp1, sm = lm1;
            if (FG_SEP_HAS(2)) { FG_SEP_DUAL(2, lp2, lm2) sp += lp2; sm += lm2; }
            if (FG_SEP_HAS(3)) { FG_SEP_DUAL(3, lp3, lm3) sp += lp3; sm += lm3; }
            tp = tp + sp; tm = tm + sm;                          // log_prior + log_likelihood
        }
        }
        const double n = tp - tm;
        g = fg_div_const(n, two_h, rcp_2h);                      // (lp - lm) / (2h), hmc.rs:322
        const double an = __builtin_fabs(n);                     // |n| outside [2^-823, 2^953] (0 and NaN too): a true division is always right
        if (__builtin_expect(__any(!(an >= 0x1p-823 && an <= 0x1p953)), 0)) g = n / two_h;
        }
        if (CHECK) bad = bad || !fg_finite(g);
        const double kick = hk * g;
        p = p + kick;                                            // hmc.rs:389 / :400
        if (gs > 0 && gs < L) { asm volatile(""); p = p + kick; }   // trailing kick of this step + leading kick of the next
        if (gs < L) { asm volatile(""); q = q + emi * p; }       // hmc.rs:391-393
    }
    if (U0) { const double lp = FG_SEP_LP_U(q); terms[a0[1] * tw] = (q != q) ? FG_NEG_INF : lp; }
    else FG_SEP_TERM(0)
    if (FG_SEP_HAS(1)) FG_SEP_TERM(1)
    if (FG_SEP_HAS(2)) FG_SEP_TERM(2)
    if (FG_SEP_HAS(3)) FG_SEP_TERM(3)
    q_io = q; p_io = p;
    return bad;
#undef FG_SEP_HAS
}

// the checked re-run of a coordinate whose endpoint momentum came out non-finite: any record mix, out of line
template <bool AN, typename RB>
__device__ __noinline__ FgD3 fg_sep_trajectory_checked(RB rb, double q, double p, double emi, double hk, int L, double h,
                                                       double two_h, double rcp_2h, double *terms, int tw, int nobs) {
    const bool bad = fg_sep_trajectory<-1, false, true, AN, false, RB>(rb, q, p, emi, hk, L, h, two_h, rcp_2h, terms, tw, nobs);
    FgD3 r; r.a = q; r.b = p; r.c = bad ? 1.0 : 0.0;
    return r;
}

// ---- MODE 3: the dense mode with a wave's coordinates in registers ------------------------------------------------------------
// The two whole log-joints of coordinate i differ from the plain in-order sum of the rows only behind i's own row, and the rows a wave
// adds for its <= 4 coordinates are the same rows: every row is read ONCE per wave and added to the running prefix (the sum all
// coordinates still share) and to the two sums of every coordinate whose own row lies behind -- the additions of two full scoring
// runs per coordinate (hmc.rs:304-329), each sum in program order, one LDS read per up to nine additions instead of one per two.
// K: coordinates already behind their own row; PSON: the shared prefix still runs.
template <int K, bool PSON>
__device__ __forceinline__ void fg_dense_add_rows(const double *p, int cnt, int tw, double &PS, double (&ap)[4], double (&am)[4]) {
#define FG_DENSE_BLOCK(R) { double y[R];                                                        \
        _Pragma("unroll") for (int q = 0; q < R; ++q) y[q] = p[q * tw];                         \
        _Pragma("unroll") for (int q = 0; q < R; ++q) { if (PSON) PS += y[q];                   \
            _Pragma("unroll") for (int jj = 0; jj < K; ++jj) { ap[jj] += y[q]; am[jj] += y[q]; } } \
        p += (long long)R * tw; }
    for (; cnt >= 8; cnt -= 8) FG_DENSE_BLOCK(8)
    if (cnt & 4) FG_DENSE_BLOCK(4)
    if (cnt & 2) FG_DENSE_BLOCK(2)
    if (cnt & 1) FG_DENSE_BLOCK(1)
#undef FG_DENSE_BLOCK
}
// one section (rows [lo, hi): log_prior or log_likelihood) for NC coordinates with own rows r[0] < r[1] < ... and own terms tp / tm
// at q + h / q - h: ap[j] / am[j] = the section's in-order sum with row r[j] replaced
template <int NC>
__device__ __forceinline__ void fg_dense_section(const double *T, int tw, int lo, int hi, const int (&r)[4], const double (&tp)[4], const double (&tm)[4],
                                                 double (&ap)[4], double (&am)[4]) {
    double PS = 0.0;
    int pos = lo;
#define FG_DENSE_STEP(J) if (NC > J) {                                                          \
        fg_dense_add_rows<J, true>(T + (long long)pos * tw, r[J] - pos, tw, PS, ap, am);       \
        const double x = T[(long long)r[J] * tw];                                               \
        ap[J] = PS + tp[J]; am[J] = PS + tm[J];                                                 \
        _Pragma("unroll") for (int jj = 0; jj < J; ++jj) { ap[jj] += x; am[jj] += x; }          \
        if (NC > J + 1) PS += x;                                                                \
        pos = r[J] + 1; }
    FG_DENSE_STEP(0) FG_DENSE_STEP(1) FG_DENSE_STEP(2) FG_DENSE_STEP(3)
#undef FG_DENSE_STEP
    fg_dense_add_rows<NC, false>(T + (long long)pos * tw, hi - pos, tw, PS, ap, am);
}

// The whole trajectory of a wave's NC coordinates (L + 1 gradients, hmc.rs:353-407) with q, p and the statements' constants in
// registers.  Per gradient: the own statements' terms at q into this gradient's term rows (two row sets take turns: ONE barrier per
// gradient), barrier, the sums above, kick and drift.  OBS: every coordinate has one observation.  All sigmas are powers of two
// (host check): (x - mu) / sigma is the exact product with 1 / sigma.  Returns "a force component was non-finite".
template <int NC, bool OBS, bool MASS>
__device__ __forceinline__ bool fg_dense_trajectory(const FgProgramDev &P, const FgChainCtx &X, const FgHmcDev &H, const int (&own)[4], double *T0, double *T1,
                                                    double *kin0, double *kin1, int tw, int n_pri, int n_s, int L, double e, double h, double two_h,
                                                    double rcp_2h, uint32_t sk0, uint32_t sk1, uint32_t gchain, uint32_t iter, long long c, bool live) {
    int ci[4] = {0, 0, 0, 0}, rP[4] = {0, 0, 0, 0}, rO[4] = {0, 0, 0, 0};
    double q[4], p[4], k0v[4], mii[4], cP[4], sP[4], lP[4], cO[4], sO[4], lO[4];
    const double hk = 0.5 * e;
    // p0 ~ N(0, M), hmc.rs:436-441: the list holds one or two whole Box-Muller pairs (Philox blocks), in row order
    const int blk_a = own[0] >> 1;
    int blk_b = blk_a;
#pragma unroll
    for (int j = 1; j < NC; ++j) if ((own[j] >> 1) != blk_a) blk_b = own[j] >> 1;
    const FgD2 za = fg_cold_normal_pair(sk0, sk1, gchain, (uint32_t)blk_a, iter, FG_RNG_HMC);
    FgD2 zb2 = za;
    if (NC > 2) zb2 = fg_cold_normal_pair(sk0, sk1, gchain, (uint32_t)blk_b, iter, FG_RNG_HMC);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        ci[j] = own[j];
        const FgSepCoord cd = P.sep_coord[ci[j]];
        const FG_AS4 char *rb = (const FG_AS4 char *)(uintptr_t)(P.sep + cd.off);
        { FG_SEP_LOAD(0) rP[j] = (int)a0[1]; cP[j] = fg_dbl(a0[2], a0[3]); sP[j] = fg_dbl(a0[4], a0[5]); lP[j] = fg_dbl(a0[6], a0[7]); (void)b0; }
        if (OBS) { FG_SEP_LOAD(1) rO[j] = (int)a1[1]; cO[j] = fg_dbl(a1[2], a1[3]); sO[j] = fg_dbl(a1[4], a1[5]); lO[j] = fg_dbl(a1[6], a1[7]); (void)b1; }
        const bool in_a = (ci[j] >> 1) == blk_a, second = (ci[j] & 1) != 0;
        const double z = in_a ? (second ? za.b : za.a) : (second ? zb2.b : zb2.a);
        mii[j] = MASS ? H.m_inv[(long long)ci[j] * X.C + c] : 1.0;
        p[j] = MASS ? z * H.mass_sqrt[(long long)ci[j] * X.C + c] : z;
        k0v[j] = MASS ? p[j] * p[j] * mii[j] : p[j] * p[j];          // hmc.rs:442-443
        q[j] = fg_as_double(X.values[(long long)P.f64_site[ci[j]] * X.C + c]);
    }
    bool bad = false;
    for (int gs = 0; gs <= L; ++gs) {
        double *T = ((L - gs) & 1) ? T1 : T0;                        // the last gradient's rows (T0) are the endpoint's score terms
        double tpP[4], tmP[4], tpO[4], tmO[4];
#define FG_DENSE_LP(x_, c_, s_, l_, out) { const double z_ = ((x_) - (c_)) * (s_); const double lp_ = -0.5 * z_ * z_ - (l_) - 0.5 * FG_LN_2PI; \
                                           out = lp_; if (__builtin_expect(__any(z_ != z_), 0)) out = (z_ != z_) ? FG_NEG_INF : lp_; }
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const double qp = q[j] + h, qm = q[j] - h;               // the perturbed coordinate holds orig +- h (hmc.rs:317-319)
            double t0; FG_DENSE_LP(q[j], cP[j], sP[j], lP[j], t0) T[(long long)rP[j] * tw] = t0;
            FG_DENSE_LP(qp, cP[j], sP[j], lP[j], tpP[j]) FG_DENSE_LP(qm, cP[j], sP[j], lP[j], tmP[j])
            if (OBS) {
                double t1; FG_DENSE_LP(q[j], cO[j], sO[j], lO[j], t1) T[(long long)rO[j] * tw] = t1;
                FG_DENSE_LP(qp, cO[j], sO[j], lO[j], tpO[j]) FG_DENSE_LP(qm, cO[j], sO[j], lO[j], tmO[j])
            }
        }
#undef FG_DENSE_LP
        __syncthreads();                                             // every statement's term at q is in T (the other row set is free again)
        double Pp[4], Pm[4], Lp[4], Lm[4];
        fg_dense_section<NC>(T, tw, 0, n_pri, rP, tpP, tmP, Pp, Pm);
        if (OBS) fg_dense_section<NC>(T, tw, n_pri, n_s, rO, tpO, tmO, Lp, Lm);
        else {
            double LS = 0.0;
            fg_dense_add_rows<0, true>(T + (long long)n_pri * tw, n_s - n_pri, tw, LS, Lp, Lm);
#pragma unroll
            for (int j = 0; j < NC; ++j) { Lp[j] = LS; Lm[j] = LS; }
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const double n = (Pp[j] + Lp[j] + 0.0) - (Pm[j] + Lm[j] + 0.0);   // total_log_weight at q + h e_i minus at q - h e_i (log_factors = 0)
            double g = fg_div_const(n, two_h, rcp_2h);                  // hmc.rs:322
            const uint32_t ne = (uint32_t)(__double_as_longlong(n) >> 32) & 0x7fffffffu;
            if (__builtin_expect(__any(!((ne - 0x0c800000u) < 0x6f000000u)), 0)) g = n / two_h;
            bad = bad || !fg_finite(g);
            double pj = p[j] + hk * g;                                   // hmc.rs:389 / :400
            if (gs > 0 && gs < L) pj = pj + hk * g;                      // trailing kick of this step + leading kick of the next
            p[j] = pj;
            if (gs < L) q[j] = q[j] + (MASS ? e * mii[j] : e) * pj;     // hmc.rs:391-393
        }
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {                                       // the proposal row; the kinetic terms of p0 and p (the idle row set)
        if (live) H.p0_scratch[(long long)ci[j] * X.C + c] = q[j];
        kin0[(long long)ci[j] * tw] = k0v[j];
        kin1[(long long)ci[j] * tw] = MASS ? p[j] * p[j] * mii[j] : p[j] * p[j];
    }
    return bad;
}

// DENSE = grad_log_joint verbatim (FG_GRAD_FD_DENSE, hmc.rs:304-329): g_i is the difference of two WHOLE log-joints.  Of their
// S + O terms only the coordinate's own change with the sign of the perturbation, but every term takes part in the two in-order
// sums.  So per gradient every wave first leaves the terms of its statements at the current q in LDS rows (double-buffered:
// one barrier per gradient), then forms, for each own coordinate, log_prior and log_likelihood at q + h e_i and q - h e_i by
// adding ALL rows in program order with the coordinate's own terms substituted -- the additions of two full scoring runs,
// without re-evaluating the S + O - (own) densities that did not move.  q and p live in LDS rows between gradients.
// MODE: 0 = dependency-aware finite difference (FG_GRAD_FD_SPARSE), 1 = DENSE, 2 = analytic (FG_GRAD_ANALYTIC)
// HALF (MODE 0, programs whose coordinates all have the same record shape with power-of-two sigmas): a tile is 32 chains, the
// lower lane half of a wave runs coordinate i of a Box-Muller pair and the upper half coordinate i + 1 of the SAME chains -- half
// the work per wave and twice the tiles when 64-chain tiles would leave CUs without one (8 192 chains on 256 CUs).  HALF = 2:
// quarter tiles of 16 chains, four coordinates per wave in its four lane groups -- when even half tiles leave a CU with ONE tile, whose
// waves all wait while wave 0 adds and decides (tools/prof_hmc_phases.py at 8 192 chains: 8 200 of a transition's 22 400 ticks); two
// quarter tiles per CU take turns.  Per
// (chain, coordinate) the arithmetic is unchanged; wave 0's in-order sums run in both halves on the same rows.
template <bool MASS, int MODE, int HALF = 0 /* 1: half tiles, 2: quarter tiles */>
__global__ __launch_bounds__(FG_WAVE * FG_SEP_WMAX, 4) void k_hmc_sep_steps(FgProgramDev P, FgChainCtx X, FgHmcDev H, FgSegSep seg, int iter0, int n_steps,
                                                                             int n_warmup, int welford_on, double *draws, int first_sample_t,
                                                                             double *pos_all /*[n][d][C] or null*/, double *info /*[n][4][C] or null*/) {
    extern __shared__ double lds[];
    constexpr int tw = FG_WAVE >> HALF;
    constexpr bool DFAST = MODE == 3, DENSE = MODE == 1 || DFAST, AN = MODE == 2;   // (3: DENSE with the coordinates in registers, fg_dense_trajectory)
    static_assert(!HALF || MODE == 0, "half tiles: sparse finite difference only");
    const int lane = threadIdx.x & (tw - 1);                       // chain of the tile
    const int half = HALF == 1 ? (int)((threadIdx.x >> 5) & 1u) : (HALF == 2 ? (int)((threadIdx.x >> 4) & 3u) : 0);   // which of the wave's 2 / 4 coordinates this lane runs
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long chain = (long long)blockIdx.x * tw + lane;
    const bool live = chain < X.C;                                 // stores of the lane's own coordinate
    const bool live0 = live && half == 0;                          // stores of per-chain state (one lane half)
    const long long c = live ? chain : X.C - 1;
    const int d = P.d, L = H.L, n_s = P.n_sstream, n_pri = P.n_prior_terms;
    // LDS tile [rows][64]: (site values, only when a statement reads no coordinate) | kinetic terms of p0 | kinetic terms of
    // the endpoint p | score terms | exchange.  The coordinates themselves are NOT in LDS: a trajectory starts from the chain's
    // committed value row in HBM (L2-resident: d * C * 8 B split over 8 XCDs) and parks its endpoint in the proposal rows
    // (H.p0_scratch) until the accept decision -- 32 rows less per tile, which is what lets TWO tiles share a CU's 160 KB at
    // d = 32, so one tile's in-order sums / accept step overlap the other tile's trajectories.
    const int srows = P.n_sep_free > 0 ? P.n_slots : 0;
    double *slots = lds + lane;
    // DENSE: (site values) | score terms | exchange | positions | momenta -- q and p live in LDS rows between gradients, and at the
    // end of a trajectory the same rows take the two kinetic-energy terms of each coordinate (p0's is formed again from its random
    // number then: one Box-Muller pair per two coordinates and transition against d rows per tile).  4 d + 2 n_s + 3 rows with
    // separate kinetic rows and double-buffered terms left ONE tile on a CU at d = 32; 2 d + n_s + 3 = 131 rows leave two, and one
    // tile's barriers overlap the other's arithmetic (tools/exp_dense_residency.py: d = 19 / 20 either side of that boundary ran
    // 4.3e9 / 2.2e9 leapfrog-steps/s).
    double *terms = lds + (long long)(srows + (DENSE ? 0 : 2 * d)) * tw + lane;
    double *xch = terms + (long long)n_s * tw;                      // rows: 0 step size, 1 accepted, 2 divergence bits, (sparse) 3 .. 6 three of the sums and the accept uniform
    double *qrow = xch + 3 * tw, *prow = qrow + (long long)d * tw;  // DENSE only (MODE 3: the second set of term rows, which takes the kinetic terms at a trajectory's end)
    double *kin0 = DENSE ? qrow : lds + (long long)srows * tw + lane;
    double *kin1 = DENSE ? prow : kin0 + (long long)d * tw;
    const int k0 = seg.c[wv], k1 = seg.c[wv + 1];
    const bool prio_turns = seg.c[FG_SEP_WMAX] == -1;             // host flags (the last boundary is otherwise d)
    const int stagger = seg.c[FG_SEP_WMAX] <= -2 ? -seg.c[FG_SEP_WMAX] - 1 : 0;   // 1: the second half of the grid starts late, 2: the odd tiles
    const double *mi = MASS ? H.m_inv + c : nullptr;
    const double *ms = MASS ? H.mass_sqrt + c : nullptr;
    const double h = fg_uniform(H.h), two_h = fg_uniform(2.0 * H.h), rcp_2h = fg_uniform(1.0 / (2.0 * H.h));
    const uint32_t sk0 = (uint32_t)X.seed, sk1 = (uint32_t)(X.seed >> 32), gchain = X.chain0 + (uint32_t)c;
    // wave 0 owns the per-chain sampler state
    double lj = 0.0, eps = 0.0, frozen = 0.0, da_mu = 0.0, da_leb = 0.0, da_hbar = 0.0, asum = 0.0, e_cur = 0.0;
    unsigned long long da_m = 0, ndiv = 0;
    if (wv == 0) {
        if (srows) fg_load_values(P, X, c, slots, tw);
        lj = H.lj[c]; eps = H.eps[c]; frozen = H.frozen[c];
        da_mu = H.da_mu[c]; da_leb = H.da_leb[c]; da_hbar = H.da_hbar[c]; da_m = H.da_m[c];
        for (int k = 0; k < P.n_sep_free; ++k) {                    // statements that read no coordinate: their terms never change
            const FgSepFree f = P.sep_free[k];
            const fg_u32x16 r = fg_fetch_grec(P.sstream, (int)f.sidx);
            FgAcc3 dummy = {0.0, 0.0, 0.0};
            terms[f.trow * tw] = fg_score_one<0>(r, slots[r[0] * tw], slots[r[1] * tw], P.pool, slots, tw, dummy);
        }
        if (iter0 < n_warmup) e_cur = eps;
        else {                                                 // frozen_or_current: hmc.rs:789-798
            if (frozen == frozen) e_cur = frozen;
            else if (n_warmup > 0) e_cur = fg_cold_exp(da_leb);
            else e_cur = eps;
            frozen = e_cur;
        }
        xch[0] = e_cur;
        xch[2 * tw] = 0.0;
    }
    __syncthreads();
#ifdef FG_HMC_PROF
    unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = __builtin_readcyclecounter();
#endif
    // Two tiles that start together on a CU stay in step -- both in their trajectories (the SIMDs full), then both waiting for their
    // wave 0 (the SIMDs empty).  The odd tiles start a wave-0 phase late: one tile's sums and accept step then fall into the other's
    // trajectories for the whole launch (the lag neither grows nor shrinks: each is ahead of the other for as long as it is behind).
    // Measured: +5 % for two half tiles on a CU (16 384 chains), nothing for two quarter tiles, -5 % for 64-chain tiles (which come
    // in several rounds and fall out of step by themselves): the host asks for it in the first case only.
    if (stagger && (stagger == 1 ? blockIdx.x >= (gridDim.x + 1) / 2 : (blockIdx.x & 1u) != 0u))
        for (int q = 0; q < FG_SEP_STAGGER; ++q) __builtin_amdgcn_s_sleep(64);            // 64 x 64 cycles each
    // The momentum of a transition needs nothing of the transition before it: while wave 0 adds, decides and adapts, the other waves
    // draw the Box-Muller pair their NEXT transition starts with (Philox is counter based: the same numbers).  A tile that is alone on
    // its CU otherwise leaves that phase's SIMD cycles empty (8 192 chains: a quarter of a trajectory's instructions are this pair).
    FgD2 zpre = {0.0, 0.0};
    bool have_pre = false;
    const uint32_t first_block = DENSE ? 0u : (HALF == 2 ? (uint32_t)(((k0 + half < d) ? k0 + half : k0) >> 1)
                                                         : (uint32_t)(k0 >> 1) + ((HALF == 1 && half && k0 + 2 < k1) ? 1u : 0u));
    for (int t = 0; t < n_steps; ++t) {
        const int iter = iter0 + t;
        const bool warming = iter < n_warmup;
        const double e = xch[0], hk = 0.5 * e;
        // ---- the own coordinates: p0 ~ N(0, M) (hmc.rs:436-441; pair j of the chain's (iteration) stream is Philox block j),
        // its kinetic term, the whole trajectory in registers, the endpoint's kinetic and score terms
        bool bad = false;
        double zb = 0.0;
        FgD2 zzh = {0.0, 0.0}; bool zh_next = false;                 // half tiles: the pair a lane half generated, and whether the upper half's is the next pair
        const double *termsE = terms;                                // rows of the endpoint's score terms
        if (DFAST) {
            const int own[4] = { seg.own[wv][0], seg.own[wv][1], seg.own[wv][2], seg.own[wv][3] };
#define FG_DENSE_CALL(NC_, OBS_) fg_dense_trajectory<NC_, OBS_, MASS>(P, X, H, own, terms, qrow, kin0, kin1, tw, n_pri, n_s, L, e, h, two_h, rcp_2h, sk0, sk1, gchain, (uint32_t)iter, c, live)
            const int nown = seg.n_own[wv];
            if (n_s > n_pri) { if (nown == 4) bad = FG_DENSE_CALL(4, true); else if (nown == 3) bad = FG_DENSE_CALL(3, true); else if (nown == 2) bad = FG_DENSE_CALL(2, true); else bad = FG_DENSE_CALL(1, true); }
            else { if (nown == 4) bad = FG_DENSE_CALL(4, false); else if (nown == 3) bad = FG_DENSE_CALL(3, false); else if (nown == 2) bad = FG_DENSE_CALL(2, false); else bad = FG_DENSE_CALL(1, false); }
#undef FG_DENSE_CALL
        } else if (DENSE) {
            for (int i = k0; i < k1; ++i) {                          // p0 ~ N(0, M), its kinetic term, the start position
                double z;
                if (!(i & 1)) { const FgD2 zz = fg_cold_normal_pair(sk0, sk1, gchain, (uint32_t)(i >> 1), (uint32_t)iter, FG_RNG_HMC); z = zz.a; zb = zz.b; }
                else z = zb;
                const double p = MASS ? z * ms[(long long)i * X.C] : z;
                prow[i * tw] = p;
                qrow[i * tw] = fg_as_double(X.values[(long long)P.f64_site[i] * X.C + c]);
            }
            for (int gs = 0; gs <= L; ++gs) {                        // leapfrog, hmc.rs:353-407
                double *T = terms;
                if (gs > 0) __syncthreads();                         // the previous gradient's sums have read T
                for (int i = k0; i < k1; ++i) {                      // the own statements at the current q
                    const FgSepCoord cd = P.sep_coord[i];
                    const FG_AS4 char *rb = (const FG_AS4 char *)(uintptr_t)(P.sep + cd.off);
                    const int nobs = (cd.n & 7) - 1;
                    constexpr bool P2 = false;
                    FG_SEP_LOAD(0) FG_SEP_LOAD(1) FG_SEP_LOAD(2) FG_SEP_LOAD(3)
                    const double q = qrow[i * tw];
                    FG_SEP_TERM_TO(0, T)
                    if (nobs >= 1) FG_SEP_TERM_TO(1, T)
                    if (nobs >= 2) FG_SEP_TERM_TO(2, T)
                    if (nobs >= 3) FG_SEP_TERM_TO(3, T)
                }
                __syncthreads();                                     // every statement's term at q is in T
                int ppos = 0, lpos = n_pri;                          // the wave's running prefixes of log_prior / log_likelihood rows at this q
                double PS = 0.0, LS = 0.0;
                for (int i = k0; i < k1; ++i) {                      // two full in-order scoring sums per own coordinate, own terms substituted
                    const FgSepCoord cd = P.sep_coord[i];
                    const FG_AS4 char *rb = (const FG_AS4 char *)(uintptr_t)(P.sep + cd.off);
                    const int nobs = (cd.n & 7) - 1;
                    constexpr bool P2 = false;
                    FG_SEP_LOAD(0) FG_SEP_LOAD(1) FG_SEP_LOAD(2) FG_SEP_LOAD(3)
                    const double q = qrow[i * tw];
                    const double qp = q + h, qm = q - h;             // the perturbed coordinate holds orig +- h (hmc.rs:317-319)
                    double Pp, Pm, Lp, Lm;
                    {                                                // log_prior: the coordinate's own sample statement is its only prior term
                        // Ahead of the own row the two sums are the same number, and the same for every coordinate: the wave keeps
                        // that prefix (rows [0, ppos) added in order from 0.0) and extends it from one own coordinate to the next
                        // instead of adding it twice per coordinate (+4 %; dealing the coordinates so that every wave has the same
                        // number of additions changed nothing: the phase is not bound by them).  Behind the own row, eight rows at
                        // compile-time offsets per chunk (ds_read2st64_b64, no address arithmetic or loop control per row; fg_inorder_run2).
                        FG_SEPD_OWN(0, tp0, tm0)
                        const int r0 = (int)a0[1];
                        if (r0 < ppos) { ppos = 0; PS = 0.0; }
                        fg_inorder_run1<8>(T + (long long)ppos * tw, r0 - ppos, tw, PS);
                        ppos = r0;
                        Pp = PS + tp0; Pm = PS + tm0;
                        fg_inorder_run2<8>(T + (long long)(r0 + 1) * tw, n_pri - r0 - 1, tw, Pp, Pm);
                    }
                    {                                                // log_likelihood: its observe statements, in program (= row) order
                        int k = n_pri;
                        if (nobs >= 1) {                             // the same prefix, of the rows ahead of the first own observation
                            FG_SEPD_OWN(1, tp1, tm1)
                            const int r1 = (int)a1[1];
                            if (r1 < lpos) { lpos = n_pri; LS = 0.0; }
                            fg_inorder_run1<8>(T + (long long)lpos * tw, r1 - lpos, tw, LS);
                            lpos = r1;
                            Lp = LS + tp1; Lm = LS + tm1; k = r1 + 1;
                        } else { Lp = 0.0; Lm = 0.0; }
                        if (nobs >= 2) { FG_SEPD_OWN(2, tp2, tm2) const int r2 = (int)a2[1]; fg_inorder_run2<8>(T + (long long)k * tw, r2 - k, tw, Lp, Lm); Lp += tp2; Lm += tm2; k = r2 + 1; }
                        if (nobs >= 3) { FG_SEPD_OWN(3, tp3, tm3) const int r3 = (int)a3[1]; fg_inorder_run2<8>(T + (long long)k * tw, r3 - k, tw, Lp, Lm); Lp += tp3; Lm += tm3; k = r3 + 1; }
                        fg_inorder_run2<8>(T + (long long)k * tw, n_s - k, tw, Lp, Lm);
                    }
                    const double n = (Pp + Lp + 0.0) - (Pm + Lm + 0.0);   // total_log_weight at q + h e_i minus at q - h e_i (log_factors = 0)
                    double g = fg_div_const(n, two_h, rcp_2h);          // hmc.rs:322
                    const uint32_t ne = (uint32_t)(__double_as_longlong(n) >> 32) & 0x7fffffffu;
                    if (__builtin_expect(__any(!((ne - 0x0c800000u) < 0x6f000000u)), 0)) g = n / two_h;
                    bad = bad || !fg_finite(g);
                    const double mii = MASS ? mi[(long long)i * X.C] : 1.0;
                    double p = prow[i * tw] + hk * g;                   // hmc.rs:389 / :400
                    if (gs > 0 && gs < L) p = p + hk * g;               // trailing kick of this step + leading kick of the next
                    prow[i * tw] = p;
                    if (gs < L) qrow[i * tw] = q + (MASS ? e * mii : e) * p;   // hmc.rs:391-393
                }
            }
            termsE = terms;                                              // the terms of the last gradient are the endpoint's
            for (int i = k0; i < k1; ++i) {                              // the proposal row; then the kinetic terms of p0 and p in the rows of q and p
                double z;
                if (!(i & 1)) { const FgD2 zz = fg_cold_normal_pair(sk0, sk1, gchain, (uint32_t)(i >> 1), (uint32_t)iter, FG_RNG_HMC); z = zz.a; zb = zz.b; }
                else z = zb;
                const double p0 = MASS ? z * ms[(long long)i * X.C] : z;
                const double mii = MASS ? mi[(long long)i * X.C] : 1.0;
                const double p = prow[i * tw];
                if (live) H.p0_scratch[(long long)i * X.C + c] = qrow[i * tw];
                kin0[i * tw] = MASS ? p0 * p0 * mii : p0 * p0;
                kin1[i * tw] = MASS ? p * p * mii : p * p;
            }
        } else
        for (int i = k0; i < k1; i += 1 << HALF) {
            // The SIMD's arbiter serves its oldest wave first: of the two waves a tile has on a SIMD the younger one (waves 4..7 of
            // 8) took 30 % longer over the same work and the tile waited for it at the barrier (tools/prof_hmc_phases.py).  The two
            // take turns at the higher priority, one coordinate each: +4 % (either wave always ahead, or turns per transition: -3 %).
            if (prio_turns) { if (((i - k0) ^ (wv >> 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
            const bool on = !HALF || i + half < d;                   // odd d: the upper half idles on the last pair
            const int ci = HALF ? (on ? i + half : i) : i;           // the lane's coordinate
            double z;
            if (HALF == 2) {                                         // quarter tiles: every lane group forms its own coordinate's pair (the pair's two groups: twice)
                const FgD2 zz = (HALF != 0 && i == k0 && have_pre) ? zpre : fg_cold_normal_pair(sk0, sk1, gchain, (uint32_t)(ci >> 1), (uint32_t)iter, FG_RNG_HMC);
                z = (ci & 1) ? zz.b : zz.a;
            } else if (HALF) {
                // Both lane halves carry the same chains, so generating a pair in both would double the Philox + Box-Muller work per
                // chain.  Instead a wave takes its pairs two at a time: the lower half generates this pair, the upper half the wave's
                // NEXT pair, and each hands the other the component it needs (one cross-half exchange per pair).  An idle upper half
                // (odd d, last pair) repeats the lower half's coordinate: the same values to the same cells.
                if (!(((i - k0) >> 1) & 1)) {
                    zh_next = i + 2 < k1;
                    zzh = (HALF != 0 && i == k0 && have_pre) ? zpre : fg_cold_normal_pair(sk0, sk1, gchain, (uint32_t)(i >> 1) + ((half && zh_next) ? 1u : 0u), (uint32_t)iter, FG_RNG_HMC);
                    const double ob = __shfl_xor(zzh.b, 32, 64);       // this pair's second component, from the lower half
                    z = half ? (on ? (zh_next ? ob : zzh.b) : zzh.a) : zzh.a;
                } else {
                    const double oa = __shfl_xor(zzh.a, 32, 64);       // the next pair's first component, from the upper half
                    z = half ? (on ? zzh.b : zzh.a) : oa;
                }
            }
            else if (!(i & 1)) { const FgD2 zz = (HALF != 0 && i == k0 && have_pre) ? zpre : fg_cold_normal_pair(sk0, sk1, gchain, (uint32_t)(i >> 1), (uint32_t)iter, FG_RNG_HMC); z = zz.a; zb = zz.b; }
            else z = zb;
            double p = MASS ? z * ms[(long long)ci * X.C] : z;
            const double mii = MASS ? mi[(long long)ci * X.C] : 1.0;
            if (on) kin0[ci * tw] = MASS ? p * p * mii : p * p;         // hmc.rs:442-443, summed in coordinate order by wave 0
            const long long gq = (long long)P.f64_site[ci] * X.C + c;
            double q = fg_as_double(X.values[gq]);
            const double emi = MASS ? e * mii : e;
            const FgSepCoord cd = P.sep_coord[i];                      // half tiles: every coordinate has this one's record shape (host check)
            const int nobs = (cd.n & 7) - 1;
            const double q0 = q, p0 = p;
            if (HALF) {
                const char *rb = (const char *)(P.sep + P.sep_coord[ci].off);
#define FG_SEP_CALLH(NO, UU) fg_sep_trajectory<NO, true, false, false, UU, const char *>(rb, q, p, emi, hk, L, h, two_h, rcp_2h, terms, tw, NO)
                if ((cd.n & 512) && nobs == 1) FG_SEP_CALLH(1, true);
                else if (nobs == 1) FG_SEP_CALLH(1, false); else if (nobs == 0) FG_SEP_CALLH(0, false); else if (nobs == 2) FG_SEP_CALLH(2, false); else FG_SEP_CALLH(3, false);
#undef FG_SEP_CALLH
                if (__builtin_expect(__any(!fg_finite(p)), 0)) {       // some force component may have been non-finite: the exact per-step test
                    const FgD3 r = fg_sep_trajectory_checked<AN, const char *>(rb, q0, p0, emi, hk, L, h, two_h, rcp_2h, terms, tw, nobs);
                    q = r.a; p = r.b; bad = bad || (on && r.c != 0.0);
                }
            } else {
            const FG_AS4 char *rb = (const FG_AS4 char *)(uintptr_t)(P.sep + cd.off);
#define FG_SEP_CALL(NO, PP) fg_sep_trajectory<NO, PP, false, AN>(rb, q, p, emi, hk, L, h, two_h, rcp_2h, terms, tw, NO)
            if (!AN && (cd.n & 512) && nobs == 1) fg_sep_trajectory<1, true, false, false, true>(rb, q, p, emi, hk, L, h, two_h, rcp_2h, terms, tw, 1);
            else if (cd.n & 256) { if (nobs == 1) FG_SEP_CALL(1, true); else if (nobs == 0) FG_SEP_CALL(0, true); else if (nobs == 2) FG_SEP_CALL(2, true); else FG_SEP_CALL(3, true); }
            else { if (nobs == 1) FG_SEP_CALL(1, false); else if (nobs == 0) FG_SEP_CALL(0, false); else if (nobs == 2) FG_SEP_CALL(2, false); else FG_SEP_CALL(3, false); }
#undef FG_SEP_CALL
            if (__builtin_expect(__any(!fg_finite(p)), 0)) {           // some force component may have been non-finite: the exact per-step test
                const FgD3 r = fg_sep_trajectory_checked<AN, const FG_AS4 char *>(rb, q0, p0, emi, hk, L, h, two_h, rcp_2h, terms, tw, nobs);
                q = r.a; p = r.b; bad = bad || r.c != 0.0;
            }
            }
            if (live && on) H.p0_scratch[(long long)ci * X.C + c] = q;       // the proposal row
            if (on) kin1[ci * tw] = MASS ? p * p * mii : p * p;
        }
        if (bad) atomicOr((unsigned long long *)(xch + 2 * tw), 1ull);
        if (prio_turns) { if (wv == 0) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }   // wave 0's sums, accept and dual averaging are the tile's path
        FG_PROF_T(0)
        __syncthreads();                                         // every coordinate's endpoint and terms
        FG_PROF_T(1)
        // four in-order sums (independent chains): H0's and the endpoint's kinetic energy in coordinate order, log_prior and
        // log_likelihood in program order (score_full, hmc.rs:283-299).  A tile with a CU to itself (small chain counts: nothing else
        // hides wave 0's serial phase) adds them on four waves, a fifth draws the accept uniform; wave 0 keeps the first sum and picks the
        // others up behind a barrier -- the same additions in the same order either way.
        const int W = (int)(blockDim.x >> 6);
        const bool sum4 = HALF != 0 && seg.sum4 != 0 && W >= 4;        // (compile-time off for 64-chain tiles: their instantiation keeps its registers)
        double s0 = 0.0;
        if (sum4) {
            if (wv == 0) s0 = fg_inorder_sum1(kin0, d, tw);
            else if (wv == 1) xch[3 * tw] = fg_inorder_sum1(kin1, d, tw);
            else if (wv == 2) xch[4 * tw] = fg_inorder_sum1(termsE, n_pri, tw);
            else if (wv == 3) xch[5 * tw] = fg_inorder_sum1(termsE + (long long)n_pri * tw, n_s - n_pri, tw);
            if (wv == (W > 4 ? 4 : 3)) xch[6 * tw] = fg_cold_u01_pair(sk0, sk1, gchain, (uint32_t)((d + 1) >> 1), (uint32_t)iter, FG_RNG_HMC).a;
            __syncthreads();
        }
        have_pre = false;
        if (HALF != 0 && seg.sum4 != 0 && wv != 0 && t + 1 < n_steps && k0 < k1 && !(k0 & 1)) {
            zpre = fg_cold_normal_pair(sk0, sk1, gchain, first_block, (uint32_t)(iter + 1), FG_RNG_HMC);
            have_pre = true;
        }
        if (wv == 0) {
            double s1, pri, lik;
            if (sum4) { s1 = xch[3 * tw]; pri = xch[4 * tw]; lik = xch[5 * tw]; }
            else {
                fg_inorder_sums2<8>(kin0, d, kin1, d, tw, s0, s1);
                fg_inorder_sums2<8>(termsE, n_pri, termsE + (long long)n_pri * tw, n_s - n_pri, tw, pri, lik);
            }
            const double h0 = -lj + 0.5 * s0;                        // hmc.rs:442-443
            const double lj_new = pri + lik + 0.0;                   // total_log_weight (log_factors = 0: no factor statement has a record)
            bool div = fg_as_i64(xch[2 * tw]) != 0;
            div = div || !fg_finite(lj_new);
            double ap = 0.0; bool acc = false;
            if (!div) {
                const double h_new = -lj_new + 0.5 * s1;
                ap = fg_cold_accept_prob(h0, h_new);             // hmc.rs:460
                const double u = sum4 ? xch[6 * tw] : fg_cold_u01_pair(sk0, sk1, gchain, (uint32_t)((d + 1) >> 1), (uint32_t)iter, FG_RNG_HMC).a;
                acc = u < ap;                                    // hmc.rs:461
            }
            if (acc) lj = lj_new;
            xch[tw] = acc ? 1.0 : 0.0;
            asum += ap; ndiv += div ? 1ull : 0ull;
            if (live0 && info) {                                 // HmcStepInfo: hmc.rs:587-602
                double *r = info + (long long)t * 4 * X.C + c;
                r[0] = acc ? 1.0 : 0.0; r[X.C] = div ? 1.0 : 0.0; r[2 * X.C] = ap; r[3 * X.C] = e_cur;
            }
            if (warming) {                                       // DualAveraging::update: hmc.rs:168-178
                da_m += 1ull;
                const FgD3 r = fg_cold_da_update(da_hbar, da_leb, (double)da_m, da_mu, H.target, ap);
                eps = r.a; da_hbar = r.b; da_leb = r.c;
            }
            // the next transition's step size
            if (iter + 1 < n_warmup) e_cur = eps;
            else {                                               // frozen_or_current: hmc.rs:789-798
                if (frozen == frozen) e_cur = frozen;
                else if (n_warmup > 0) e_cur = fg_cold_exp(da_leb);
                else e_cur = eps;
                if (t + 1 < n_steps) frozen = e_cur;
            }
            if (t + 1 < n_steps) xch[0] = e_cur;
            xch[2 * tw] = 0.0;
        }
        FG_PROF_T(2)
        if (prio_turns) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        FG_PROF_T(3)
        const bool acc = xch[tw] != 0.0;
        unsigned long long wn = 0;
        if (warming && welford_on) wn = H.w_n[c] + 1ull;          // every wave reads the old count before wave 0 bumps it below
        for (int jo = 0; jo < (DFAST ? seg.n_own[wv] : k1 - k0); jo += 1 << HALF) {   // commit or roll back the own f64 sites
            const int i0 = DFAST ? seg.own[wv][jo] : k0 + jo;
            const bool on = !HALF || i0 + half < d;
            const int i = HALF ? (on ? i0 + half : i0) : i0;
            const long long g = (long long)P.f64_site[i] * X.C + c;
            const double x = acc ? H.p0_scratch[(long long)i * X.C + c] : fg_as_double(X.values[g]);
            if (acc && live && on) X.values[g] = fg_as_i64(x);
            if (live && on && pos_all) pos_all[((long long)t * d + i) * X.C + c] = x;
            if (warming) {
                if (welford_on) {                                 // Welford::push: hmc.rs:202-211
                    const long long gi = (long long)i * X.C + c;
                    const double n = (double)wn;
                    double mean = H.w_mean[gi];
                    const double delta = x - mean;
                    mean += delta / n;
                    const double delta2 = x - mean;
                    if (live && on) { H.w_mean[gi] = mean; H.w_m2[gi] += delta * delta2; }
                }
            } else if (draws && live && on) draws[((long long)(t - first_sample_t) * d + i) * X.C + c] = x;   // hmc.rs:577-582
        }
        if (warming && welford_on) {
            __syncthreads();                                      // all waves hold the old count
            if (wv == 0 && live0) H.w_n[c] = wn;
        }
        FG_PROF_T(4)
    }
#ifdef FG_HMC_PROF
    if (blockIdx.x == 0 && lane == 0) for (int q = 0; q < 8; ++q) fg_hmc_prof[wv][q] = prof_[q];
#endif
    if (wv == 0 && live0) {
        H.lj[c] = lj; H.eps[c] = eps; H.frozen[c] = frozen;
        H.da_mu[c] = da_mu; H.da_leb[c] = da_leb; H.da_hbar[c] = da_hbar; H.da_m[c] = da_m;
        H.alpha_sum[c] += asum; H.n_div[c] += ndiv;
    }
}

// Launch for `n` transitions from iteration `iter0`; returns FG_E_UNSUPPORTED when the program / configuration is not an
// independent-sites FD-sparse run (the caller then takes the gradient-stream kernel).
int fg_hmc_sep_launch(fg_engine *e, int iter0, int n, int welford_on, double *draws, int first_sample_t, double *pos_all, double *info) {
    const bool dense = e->cfg.grad_mode == FG_GRAD_FD_DENSE, analytic = e->cfg.grad_mode == FG_GRAD_ANALYTIC;
    if (e->gt) return FG_E_UNSUPPORTED;                       // tiles in global memory: the one-wave-per-tile kernels (fg_engine.hip)
    if (!e->P.sep || (e->cfg.grad_mode != FG_GRAD_FD_SPARSE && !dense && !analytic) || e->d < 1 || e->sep_disabled) return FG_E_UNSUPPORTED;
    const long long n_cu = std::max(1, e->n_simd / 4);
    const long long tiles64 = (e->C + FG_WAVE - 1) / FG_WAVE;
    // half tiles (32 chains per workgroup, the two coordinates of a Box-Muller pair in the two lane halves): when 64-chain tiles
    // would leave half of the CUs without one, for programs whose coordinates all have one record shape with power-of-two sigmas
    int half = 0;                                            // 1: half tiles, 2: quarter tiles (16 chains, four coordinates per wave)
    if (!dense && !analytic && e->d >= 2) {
        const std::vector<FgSepCoord> &cd = e->prog->sep_coord;
        bool uniform = true;
        for (const FgSepCoord &q : cd) uniform = uniform && q.n == cd[0].n && (q.n & 256);
        // half tiles up to two of them per CU (16 384 chains: 1.57e10 with 64-chain tiles, 1.67e10, 1.75e10 with the late start below);
        // quarter tiles where even half tiles leave CUs without one (4 096 chains: 6.7e9 -> 9.2e9; at 8 192 the two are level)
        if (uniform && tiles64 <= n_cu) half = (4 * tiles64 < 2 * n_cu && e->d >= 8) ? 2 : 1;
        if (const char *hv = std::getenv("FG_HMC_SEP_HALF")) half = uniform ? std::max(0, std::min(2, std::atoi(hv))) : 0;
        if (half == 2 && e->d < 4) half = 1;
    }
    const int tw = FG_WAVE >> half;
    const unsigned tiles = (unsigned)((e->C + tw - 1) / tw);
    // dense with the coordinates in registers (fg_dense_trajectory): every coordinate one Normal prior with or without ONE observation,
    // all sigmas powers of two, no statement that reads no coordinate
    bool dfast = dense && e->P.n_sep_free == 0 && e->d >= 1;
    if (dfast) {
        const std::vector<FgSepCoord> &cd = e->prog->sep_coord;
        for (const FgSepCoord &q : cd) dfast = dfast && (q.n & 256) && (q.n & 7) == (cd[0].n & 7) && ((q.n & 7) == 1 || (q.n & 7) == 2);
        if (const char *dv = std::getenv("FG_HMC_DENSE_FAST")) dfast = dfast && std::atoi(dv) != 0;
    }
    const size_t rows = (size_t)(e->P.n_sep_free > 0 ? e->n_slots : 0) + 2 * (size_t)e->d + (size_t)e->P.n_sstream + 3 +
                        (dense ? 8 : 4 + 8);     // (dense: the kinetic terms end the tile -- the in-order sums read whole chunks of eight rows; sparse: 4 exchange rows + the chunk a sum may read past them)
    // (the register-resident dense form has two sets of n_s term rows, the second of which also takes the 2 d kinetic terms: with n_s = d
    // or 2 d that is the 2 d + n_s rows of the row-resident form)
    const size_t lds = rows * tw * sizeof(double);
    if (lds > 160 * 1024) return FG_E_UNSUPPORTED;
    // waves per tile: aim at 4 waves per SIMD (16 per CU); the LDS tile caps the tiles resident on a CU, few tiles (small
    // chain counts) leave CUs with one tile -- the waves then come from sharing the tile.  Every wave owns >= 2 coordinates
    // (a half tile: >= 1 pair, both coordinates at once).
    const int unit = half == 2 ? 4 : 2;                      // coordinates a wave takes at a time
    const int pairs = (e->d + unit - 1) / unit;
    int W = e->mw_override > 0 ? e->mw_override : 1;
    if (e->mw_override <= 0) {
        const long long resident = std::max(1LL, std::min<long long>((160 * 1024) / (long long)lds, ((long long)tiles + n_cu - 1) / n_cu));
        while (W < FG_SEP_WMAX && resident * W < 16 && (half ? pairs >= 2 * W : e->d >= 4 * W)) W *= 2;     // (a half tile with a pair per wave beats two pairs per wave sharing their random numbers: 1.31e10 against 1.25e10 at 8 192 chains)
    }
    while (W > 1 && unit * (W - 1) >= e->d + 1) W /= 2;          // no empty waves
    FgSegSep seg;
    std::memset(&seg, 0, sizeof(seg));
    if (dfast) {
        // Whole Box-Muller pairs, at most two per wave.  A coordinate whose own row is r adds 2 (rows - r) terms behind it: the pairs go
        // out by row, to the waves and back (0 .. W-1, W-1 .. 0), so every wave adds about the same number; within a wave by row.
        const int np = (e->d + 1) / 2, W_plain = W;
        W = std::min(std::max(W, (np + 1) / 2), np);
        if (W > FG_SEP_WMAX) dfast = false;
        else {
            const std::vector<FgSepCoord> &cd = e->prog->sep_coord;
            auto row_of = [&](int i, int k) { return (int)e->prog->sep[(size_t)cd[i].off + k].trow; };
            std::vector<int> pr(np);
            for (int q = 0; q < np; ++q) pr[q] = q;
            std::stable_sort(pr.begin(), pr.end(), [&](int a, int b) { return row_of(2 * a, 0) < row_of(2 * b, 0); });
            for (int q = 0; q < np; ++q) {
                const int w = q < W ? q : 2 * W - 1 - q;
                for (int i = 2 * pr[q]; i < std::min(e->d, 2 * pr[q] + 2); ++i) seg.own[w][seg.n_own[w]++] = i;
            }
            const bool obs = (cd[0].n & 7) == 2;
            for (int w = 0; w < W && dfast; ++w) {
                std::sort(seg.own[w], seg.own[w] + seg.n_own[w], [&](int a, int b) { return row_of(a, 0) < row_of(b, 0); });
                for (int j = 0; j + 1 < seg.n_own[w]; ++j)
                    if (row_of(seg.own[w][j], 0) >= row_of(seg.own[w][j + 1], 0) || (obs && row_of(seg.own[w][j], 1) >= row_of(seg.own[w][j + 1], 1))) dfast = false;
                if (seg.n_own[w] < 1) dfast = false;
            }
        }
        if (!dfast) { std::memset(&seg, 0, sizeof(seg)); W = W_plain; }       // (rows out of order within a wave, or more than 32 pairs: the row-resident form)
    }
    seg.sum4 = (!dense && half != 0) ? 1 : 0;                    // a tile alone on its CU: the transition's four end sums on four waves
    if (const char *sv = std::getenv("FG_HMC_SUM4")) seg.sum4 = std::atoi(sv) != 0 ? 1 : 0;
    for (int w = 0; w <= FG_SEP_WMAX; ++w) seg.c[w] = e->d;
    for (int w = 0; w < W; ++w) seg.c[w] = std::min(e->d, unit * (int)((long long)pairs * w / W));
    if (W == 8 && !half && !(std::getenv("FG_HMC_PRIO") && std::atoi(std::getenv("FG_HMC_PRIO")) == 0)) seg.c[FG_SEP_WMAX] = -1;   // priority turns: two waves of a tile per SIMD
    if (half == 1 && (long long)tiles > n_cu && (long long)tiles <= 2 * n_cu) seg.c[FG_SEP_WMAX] = -3;                                  // two half tiles on a CU: the odd ones start late (+5 %; 64-chain tiles lose 5 % to it)
    if (const char *sg = std::getenv("FG_HMC_STAGGER")) { const int v = std::atoi(sg); seg.c[FG_SEP_WMAX] = (v == 1 || v == 2) ? -1 - v : (seg.c[FG_SEP_WMAX] <= -2 ? e->d : seg.c[FG_SEP_WMAX]); }   // experiments
    static bool attr_set_dev[64][12];
    const int mass = e->H.use_mass ? 1 : 0, mode = dense ? 1 : (analytic ? 2 : 0), variant = dfast ? 10 + mass : (half ? 4 + 2 * half + mass : 2 * mode + mass);
    const void *fns[12] = { (const void *)k_hmc_sep_steps<false, 0>, (const void *)k_hmc_sep_steps<true, 0>, (const void *)k_hmc_sep_steps<false, 1>,
                           (const void *)k_hmc_sep_steps<true, 1>, (const void *)k_hmc_sep_steps<false, 2>, (const void *)k_hmc_sep_steps<true, 2>,
                           (const void *)k_hmc_sep_steps<false, 0, 1>, (const void *)k_hmc_sep_steps<true, 0, 1>,
                           (const void *)k_hmc_sep_steps<false, 0, 2>, (const void *)k_hmc_sep_steps<true, 0, 2>,
                           (const void *)k_hmc_sep_steps<false, 3>, (const void *)k_hmc_sep_steps<true, 3> };
    bool &attr_set = attr_set_dev[e->device & 63][variant];
    if (!attr_set) {
        const hipError_t he = hipFuncSetAttribute(fns[variant], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (he != hipSuccess) { fg_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(he)); return FG_E_HIP; }
        attr_set = true;
    }
#define FG_SEP_LAUNCH(...) hipLaunchKernelGGL((k_hmc_sep_steps<__VA_ARGS__>), dim3(tiles), dim3(FG_WAVE * W), lds, e->stream, e->P, e->X, e->H, seg, iter0, n, \
                                               e->n_warmup, welford_on, draws, first_sample_t, pos_all, info)
    switch (variant) {
        case 0: FG_SEP_LAUNCH(false, 0); break; case 1: FG_SEP_LAUNCH(true, 0); break; case 2: FG_SEP_LAUNCH(false, 1); break;
        case 3: FG_SEP_LAUNCH(true, 1); break;  case 4: FG_SEP_LAUNCH(false, 2); break; case 5: FG_SEP_LAUNCH(true, 2); break;
        case 6: FG_SEP_LAUNCH(false, 0, 1); break; case 7: FG_SEP_LAUNCH(true, 0, 1); break;
        case 8: FG_SEP_LAUNCH(false, 0, 2); break; case 9: FG_SEP_LAUNCH(true, 0, 2); break;
        case 10: FG_SEP_LAUNCH(false, 3); break; default: FG_SEP_LAUNCH(true, 3); break;
    }
#undef FG_SEP_LAUNCH
    HIPCHK(hipGetLastError());
    e->last_hmc_kernel = std::string(dfast ? "k_hmc_sep_steps (dense, coordinates in registers) W=" : dense ? "k_hmc_sep_steps (dense) W=" : (analytic ? "k_hmc_sep_steps (analytic) W=" : (half == 2 ? "k_hmc_sep_steps (quarter tiles) W=" : (half ? "k_hmc_sep_steps (half tiles) W=" : "k_hmc_sep_steps W=")))) + std::to_string(W);
    return FG_OK;
}

#ifdef FG_HMC_PROF
extern "C" int fg_debug_hmc_prof(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fg_hmc_prof), sizeof(unsigned long long) * FG_SEP_WMAX * 8) == hipSuccess ? 0 : -1;
}
#endif
