// fg_engine.hip -- gfx950 kernels + engine half of the C ABI (include/fugue_amd.h).
//
// Data layout in HBM (all per-engine, struct-of-arrays, LAST index = chain so that the
// 64 lanes of a wave touch 64 consecutive 8-byte words):
//   values   [S][C]   trace cells (f64 bits or int64)              <- current state
//   lj       [C]      log pi of the current state (HmcSession::lj_cur, hmc.rs:648)
//   eps      [C]      running step size; frozen [C] (NaN = not frozen yet)
//   da_mu, da_leb, da_hbar [C], da_m [C]   DualAveraging (hmc.rs:141-150)
//   m_inv, mass_sqrt [d][C], w_mean, w_m2 [d][C], w_n [C]   (only when adapt_mass)
//   alpha_sum [C], n_div [C]               per-chain statistics
// One block = one wave of 64 chains; the wave's working set (site values + expression
// temporaries + momentum) is a [(n_slots + d)][64] tile of doubles in LDS.
#include "fg_engine_internal.h"
#include "fg_gradstream.h"
#include "fg_cold.h"

// ---- run(PriorHandler, model) per chain: interpreters.rs:88-104 ----
template <bool GT>
__global__ __launch_bounds__(FG_WAVE, FG_MIN_WAVES) void k_prior_init(FgProgramDev P, FgChainCtx X, uint32_t iteration, uint32_t purpose,
                                                        double *acc_out /*[3][C]*/, double *lj_out /*[C]*/) {
    extern __shared__ double lds_[];
    double *lds = GT ? X.gtile + (size_t)blockIdx.x * X.gtile_rows * FG_WAVE : lds_;        // GT: the tile lives in global memory (programs beyond 160 KB of LDS)
    constexpr int tw = FG_WAVE;                        // tile width: every lane of the wave owns a chain
    const long long chain = (long long)blockIdx.x * tw + threadIdx.x;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    double *slots = lds + threadIdx.x;
    for (int j = 0; j < P.n_slots; ++j) slots[j * tw] = 0.0;
    FgStream rng = fg_stream(X.seed, X.chain0 + (uint32_t)c, iteration, purpose);
    FgAcc3 A = {0.0, 0.0, 0.0};
    fg_exec<FG_MODE_PRIOR, false>(P.ins, P.n_ins, P.pool, slots, tw, A, &rng, nullptr, 0, live);
    if (live) {
        fg_store_values(P, X, c, slots, tw);
        if (acc_out) { acc_out[c] = A.prior; acc_out[X.C + c] = A.lik; acc_out[2 * X.C + c] = A.fac; }
        if (lj_out) lj_out[c] = fg_total(A);
    }
}

// ---- run(ScoreGivenTrace, model) per chain: interpreters.rs:138-163 ----
template <bool GT>
__global__ __launch_bounds__(FG_WAVE, FG_MIN_WAVES) void k_log_joint(FgProgramDev P, FgChainCtx X, double *acc_out, double *logp_out,
                                                       double *lj_out) {
    extern __shared__ double lds_[];
    double *lds = GT ? X.gtile + (size_t)blockIdx.x * X.gtile_rows * FG_WAVE : lds_;        // GT: the tile lives in global memory (programs beyond 160 KB of LDS)
    constexpr int tw = FG_WAVE;                        // tile width: every lane of the wave owns a chain
    const long long chain = (long long)blockIdx.x * tw + threadIdx.x;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    double *slots = lds + threadIdx.x;
    fg_load_values(P, X, c, slots, tw);
    FgAcc3 A = {0.0, 0.0, 0.0};
    fg_exec<FG_MODE_SCORE, true>(P.ins_fast, P.n_ins, P.pool, slots, tw, A, nullptr, logp_out ? logp_out + c : nullptr, X.C, live);
    if (live) {
        if (acc_out) { acc_out[c] = A.prior; acc_out[X.C + c] = A.lik; acc_out[2 * X.C + c] = A.fac; }
        if (lj_out) lj_out[c] = fg_total(A);
    }
}

// ---- the same scoring run over the SCORE STREAM (one 64-byte record per statement, program order): what the HMC
// endpoint, the MH pre-run proposal path and SMC rejuvenation evaluate.  rec_lp (optional) [n_sstream][C] receives every
// record's log-density, so each record kind can be checked against the reference's known-answer values.
template <bool GT>
__global__ __launch_bounds__(FG_WAVE, FG_MIN_WAVES) void k_log_joint_stream(FgProgramDev P, FgChainCtx X, double *acc_out, double *rec_lp) {
    extern __shared__ double lds_[];
    double *lds = GT ? X.gtile + (size_t)blockIdx.x * X.gtile_rows * FG_WAVE : lds_;        // GT: the tile lives in global memory (programs beyond 160 KB of LDS)
    constexpr int tw = FG_WAVE;
    const long long chain = (long long)blockIdx.x * tw + threadIdx.x;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    double *slots = lds + threadIdx.x;
    fg_load_values(P, X, c, slots, tw);
    FgAcc3 A = {0.0, 0.0, 0.0};
    for (int k = 0; k < P.n_sstream; ++k) {
        const fg_u32x16 r = fg_fetch_grec(P.sstream, k);
        const double xs = slots[r[0] * tw], ms = slots[r[1] * tw];
        const double lp = fg_score_one<2>(r, xs, ms, P.pool, slots, tw, A);
        if (live && rec_lp) rec_lp[(long long)k * X.C + c] = lp;
    }
    if (live && acc_out) { acc_out[c] = A.prior; acc_out[X.C + c] = A.lik; acc_out[2 * X.C + c] = A.fac; }
}

// ---- HMC building blocks ---------------------------------------------------------------

// leapfrog (hmc.rs:353-407) followed by the endpoint score (score_full, hmc.rs:283-299), as
// ONE flat loop over model evaluations so that the interpreter has a single call site:
//   evaluations 0 .. (L+1)*2d-1 : grad_log_joint (hmc.rs:304-329), coordinate i of gradient
//                                 s = e / 2d, at q_i + h (even) then q_i - h (odd);
//   last evaluation             : the full program at the endpoint.
// q lives in `slots`, p in `pl`.  The gradient is consumed coordinate by coordinate: p_i gets
// the trailing half-kick of step s and the leading half-kick of step s+1 as soon as g_i is
// known -- the same operations on p_i, in the same order, as the reference's vector loops --
// so no gradient vector is stored.  DENSE re-runs the whole program per evaluation (the
// reference's 2d model runs per gradient); SPARSE re-runs only the statements that read
// coordinate i (identical central difference, the cancelling terms are never formed).
// Returns `divergent` (a non-finite force component or endpoint log-joint).
__device__ __forceinline__ bool fg_trajectory(const FgProgramDev &P, double *slots, double *pl, int tw, double eps, int L, double h,
                                              bool sparse, const double *m_inv /*[d][C] column or null*/, long long C,
                                              double &lj_end) {
    const int d = P.d;
    const double hk = 0.5 * eps;
    const int n_evals = (L + 1) * 2 * d + 1;
    bool bad = false;
    int s = 0, i = 0, slot = 0;
    FgCoord cd = {0, 0, 0, 0};
    double orig = 0.0, lp_plus = 0.0;
    lj_end = FG_NEG_INF;
    int e0 = 0;
    if (sparse && P.gstream) {
        // all force terms are fast Normals: (L+1) fused gradient passes, then only the endpoint score below
        for (int gs = 0; gs <= L; ++gs) {
#ifndef FG_EXP_NOSTREAM
            bad = fg_grad_stream<2>(P.gstream, P.n_gstream, P.pool, slots, pl, tw, h, hk, gs > 0 && gs < L, nullptr, 0, false) || bad;
#endif
            if (__all(bad)) return true;
#ifndef FG_EXP_NODRIFT
            if (gs < L) {
                if (m_inv) {                                      // q += eps * M^-1 p   (hmc.rs:391-393)
                    for (int k = 0; k < d; ++k) slots[k * tw] += eps * m_inv[(long long)k * C] * pl[k * tw];
                } else {                                          // identity mass: eps * 1.0 * p == eps * p exactly
#pragma unroll 8
                    for (int k = 0; k < d; ++k) slots[k * tw] += eps * pl[k * tw];
                }
            }
#endif
        }
        e0 = n_evals - 1;
    }
    for (int e = e0; e < n_evals; ++e) {
        const bool is_final = (e == n_evals - 1);
        const bool minus = (e & 1) != 0;
        const FgIns *prog = P.ins_fast;
        int n = P.n_ins;
        if (!is_final) {
            if (!minus) { cd = P.coord[i]; slot = cd.slot; orig = slots[slot * tw]; slots[slot * tw] = orig + h; }
            else slots[slot * tw] = orig - h;
            if (sparse) { prog = P.sub + cd.sub_off; n = cd.sub_n; }
        }
        FgAcc3 A = {0.0, 0.0, 0.0};
        fg_exec<FG_MODE_SCORE, false>(prog, n, P.pool, slots, tw, A, nullptr, nullptr, 0, false);
        const double tot = fg_total(A);
        if (is_final) { lj_end = tot; break; }
        if (!minus) { lp_plus = tot; continue; }
        slots[slot * tw] = orig;
        const double g = (lp_plus - tot) / (2.0 * h);            // hmc.rs:322
        bad = bad || !fg_finite(g);
        double p = pl[i * tw];
        p += hk * g;                                              // hmc.rs:389 / :400
        if (s > 0 && s < L) p += hk * g;                          // trailing kick of step s + leading kick of s+1
        pl[i * tw] = p;
        if (++i == d) {
            i = 0;
            if (__all(bad)) return true;                          // every lane left the support (hmc.rs:384-398)
            if (s < L) {                                          // q += eps * M^-1 p   (hmc.rs:391-393)
                for (int k = 0; k < d; ++k) {
                    const double mi = m_inv ? m_inv[(long long)k * C] : 1.0;
                    slots[k * tw] += eps * mi * pl[k * tw];
                }
            }
            ++s;
        }
    }
    return bad || !fg_finite(lj_end);
}

struct FgTransOut { bool accepted, divergent; double alpha, lj; };

// hmc_transition (hmc.rs:419-473).  On entry: slots = current q (and discrete sites),
// pl = p0.  On exit: slots hold the NEXT state (accepted endpoint or the restored current
// state) and X.values is up to date.
__device__ __forceinline__ FgTransOut fg_hmc_transition(const FgProgramDev &P, const FgChainCtx &X, const FgHmcDev &H, long long c,
                                                        bool live, double *slots, double *pl, int tw, double lj_cur, double eps,
                                                        double u) {
    const double *mi = H.use_mass ? H.m_inv + c : nullptr;
    const double h0 = -lj_cur + fg_kinetic(P, pl, tw, mi, X.C);      // hmc.rs:442-443
    double lj_new;
    const bool div = fg_trajectory(P, slots, pl, tw, eps, H.L, H.h, H.grad_mode != FG_GRAD_FD_DENSE, mi, X.C, lj_new);
    FgTransOut o;
    o.divergent = div;
    double ap = 0.0;
    bool acc = false;
    if (!div) {
        const double h_new = -lj_new + fg_kinetic(P, pl, tw, mi, X.C);
        ap = fmin(exp(h0 - h_new), 1.0);                          // hmc.rs:460
        acc = u < ap;                                             // hmc.rs:461
    }
    o.alpha = ap; o.accepted = acc; o.lj = acc ? lj_new : lj_cur;
    for (int i = 0; i < P.d; ++i) {                               // commit or roll back the f64 sites
        const long long g = (long long)P.f64_site[i] * X.C + c;
        if (acc) { if (live) X.values[g] = fg_as_i64(slots[i * tw]); }
        else slots[i * tw] = fg_as_double(X.values[g]);
    }
    return o;
}

// HmcSession::step x n_steps (hmc.rs:819-919), d > 0, without the mass-matrix reset (the
// host splits launches at that iteration).
template <bool GT>
__global__ __launch_bounds__(FG_WAVE, FG_MIN_WAVES) void k_hmc_steps(FgProgramDev P, FgChainCtx X, FgHmcDev H, int iter0, int n_steps,
                                                       int n_warmup, int welford_on, double *draws, int first_sample_t,
                                                       double *pos_all /*[n][d][C] or null*/, double *info /*[n][4][C] or null*/) {
    extern __shared__ double lds_[];
    double *lds = GT ? X.gtile + (size_t)blockIdx.x * X.gtile_rows * FG_WAVE : lds_;        // GT: the tile lives in global memory (programs beyond 160 KB of LDS)
    constexpr int tw = FG_WAVE;                        // tile width: every lane of the wave owns a chain
    const long long chain = (long long)blockIdx.x * tw + threadIdx.x;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    double *slots = lds + threadIdx.x;
    double *pl = lds + (long long)P.n_slots * tw + threadIdx.x;
    fg_load_values(P, X, c, slots, tw);
    double lj = H.lj[c], eps = H.eps[c], frozen = H.frozen[c];
    double da_mu = H.da_mu[c], da_leb = H.da_leb[c], da_hbar = H.da_hbar[c];
    unsigned long long da_m = H.da_m[c];
    double asum = 0.0; unsigned long long ndiv = 0;
    const double *ms = H.use_mass ? H.mass_sqrt + c : nullptr;

    for (int t = 0; t < n_steps; ++t) {
        const int iter = iter0 + t;
        const bool warming = iter < n_warmup;
        double e;
        if (warming) e = eps;
        else {                                             // frozen_or_current: hmc.rs:789-798
            if (frozen == frozen) e = frozen;
            else if (n_warmup > 0) e = exp(da_leb);
            else e = eps;
            frozen = e;
        }
        FgStream rng = fg_stream(X.seed, X.chain0 + (uint32_t)c, (uint32_t)iter, FG_RNG_HMC);
        fg_draw_momentum(P, rng, pl, tw, ms, X.C);
        const double u = fg_rng_u01(rng);
        const FgTransOut o = fg_hmc_transition(P, X, H, c, live, slots, pl, tw, lj, e, u);
        lj = o.lj;
        asum += o.alpha; ndiv += o.divergent ? 1ull : 0ull;
        if (live && info) {                                // HmcStepInfo: hmc.rs:587-602
            double *r = info + (long long)t * 4 * X.C + c;
            r[0] = o.accepted ? 1.0 : 0.0; r[X.C] = o.divergent ? 1.0 : 0.0; r[2 * X.C] = o.alpha; r[3 * X.C] = e;
        }
        if (live && pos_all) {
            double *row = pos_all + (long long)t * P.d * X.C + c;
            for (int i = 0; i < P.d; ++i) row[(long long)i * X.C] = slots[i * tw];
        }
        if (warming) {                                     // DualAveraging::update: hmc.rs:168-178
            da_m += 1ull;
            const double m = (double)da_m;
            const double a = o.alpha < 0.0 ? 0.0 : (o.alpha > 1.0 ? 1.0 : o.alpha);
            const double frac = 1.0 / (m + 10.0);
            da_hbar = (1.0 - frac) * da_hbar + frac * (H.target - a);
            const double log_eps = da_mu - (sqrt(m) / 0.05) * da_hbar;
            const double w = pow(m, -0.75);
            da_leb = w * log_eps + (1.0 - w) * da_leb;
            eps = exp(log_eps);
            if (welford_on) {                              // Welford::push: hmc.rs:202-211
                const unsigned long long wn = H.w_n[c] + 1ull;
                if (live) H.w_n[c] = wn;
                const double n = (double)wn;
                for (int i = 0; i < P.d; ++i) {
                    const long long g = (long long)i * X.C + c;
                    const double x = slots[i * tw];
                    double mean = H.w_mean[g];
                    const double delta = x - mean;
                    mean += delta / n;
                    const double delta2 = x - mean;
                    if (live) { H.w_mean[g] = mean; H.w_m2[g] += delta * delta2; }
                }
            }
        } else if (draws && live) {                        // hmc_chain pushes the current state: hmc.rs:577-582
            double *row = draws + (long long)(t - first_sample_t) * P.d * X.C + c;
            for (int i = 0; i < P.d; ++i) row[(long long)i * X.C] = slots[i * tw];
        }
    }
    if (live) {
        H.lj[c] = lj; H.eps[c] = eps; H.frozen[c] = frozen;
        H.da_mu[c] = da_mu; H.da_leb[c] = da_leb; H.da_hbar[c] = da_hbar; H.da_m[c] = da_m;
        H.alpha_sum[c] += asum; H.n_div[c] += ndiv;
    }
}

// ---- multi-wave HmcSession::step for programs with a fused gradient stream ---------------------------------------
// One wave issues in order: with one wave per tile every scalar instruction, branch, LDS/SMEM wait and address
// computation of the interpreter sits between the f64 VALU instructions (a lone wave needs ~6 cycles per dependent
// f64 op, the SIMD's f64 pipe takes one per ~4.3 cycles from >= 2 waves: tools/mb_clock.hip,
// profiles/round1_f64_issue_microbench.txt), and the LDS tile (36 KB for the 32-site model) caps a CU at 4 tiles,
// i.e. ONE wave per SIMD however many chains there are -- nothing overlaps anything.
// Here a tile of 64 chains is owned by a workgroup of W waves (W = 1 ... 16): every coordinate's finite-difference
// gradient, half-kick, drift and commit is independent of the other coordinates' within one leapfrog step, so wave w
// does coordinates [seg.c[w], seg.c[w+1]) -- its run of whole coordinates of the gradient stream -- on the SHARED LDS
// tile, with a workgroup barrier between "all p kicked" and "q drifted" (hmc.rs:389-400: the same operations per
// coordinate in the same order; only the interleaving BETWEEN coordinates differs, and they do not interact).  The
// momentum pairs are drawn by all waves (counter-based RNG); the sequential parts (Hamiltonians, endpoint score in
// program order, accept, dual averaging) stay on wave 0.  Results are bit-identical for every W (tests/test_gpu_parity.py::test_hmc_multiwave_is_bit_identical).
struct FgSeg { int c[FG_MW_MAX + 1]; int g[FG_MW_MAX + 1];
               int separable; };   // every record of a wave reads only that wave's own coordinates (and constants): no barrier inside the leapfrog loop

// the interpreter's scoring run behind a call: its seventeen inlined log-densities would otherwise take part in the register
// allocation of the leapfrog loop
static __device__ __noinline__ FgAcc3 fg_cold_score_exec(const FgIns *ins, int n_ins, const double *pool, double *slots, int tw) {
    FgAcc3 A = {0.0, 0.0, 0.0};
    fg_exec<FG_MODE_SCORE, false>(ins, n_ins, pool, slots, tw, A, nullptr, nullptr, 0, false);
    return A;
}
// propose_and_score (SingleSiteProposalHandler, mh.rs:298-570) behind a call as well: taken only when some lane's site needs
// the model to make its proposal (undecided kinds, prior-resample kinds, computed Categorical tables)
static __device__ __noinline__ FgAcc3 fg_cold_mh_exec(const FgIns *ins, int n_ins, const double *pool, double *slots, int tw, bool live, FgMhCtx *mh) {
    FgAcc3 A = {0.0, 0.0, 0.0};
    fg_exec<FG_MODE_MH, false>(ins, n_ins, pool, slots, tw, A, nullptr, nullptr, 0, live, mh);
    return A;
}

// SS: the program has a score stream (endpoint score = one pass over records); without it the endpoint score is the
// interpreter's scoring run, kept in its own instantiations
template <int RK, bool AN, bool SS>
__global__ __launch_bounds__(FG_WAVE * FG_MW_MAX, 4) void k_hmc_stream_steps(FgProgramDev P, FgChainCtx X, FgHmcDev H, FgSeg seg, int iter0, int n_steps,
                                                                               int n_warmup, int welford_on, double *draws, int first_sample_t,
                                                                               double *pos_all /*[n][d][C] or null*/, double *info /*[n][4][C] or null*/) {
    extern __shared__ double lds[];
    constexpr int tw = FG_WAVE;
    const int lane = threadIdx.x & (FG_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = (int)(blockDim.x >> 6);
    const long long chain = (long long)blockIdx.x * tw + lane;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    double *slots = lds + lane;
    double *pl = lds + (long long)P.n_slots * tw + lane;
    double *xch = lds + (long long)(P.n_slots + P.d) * tw + lane;      // rows: 0 step size, 1 accepted, 2.. per-wave divergence flags
    const int k0 = seg.c[wv], k1 = seg.c[wv + 1];
    const FgGradRec *gs0 = P.gstream + seg.g[wv];
    const int gn = seg.g[wv + 1] - seg.g[wv];
    const int d = P.d, L = H.L;
    const bool dense = H.grad_mode == FG_GRAD_FD_DENSE;          // whole-program finite difference over the score stream
    const bool sep = seg.separable != 0 && !dense;
    const double *mi = H.use_mass ? H.m_inv + c : nullptr;
    const double *ms = H.use_mass ? H.mass_sqrt + c : nullptr;
    const uint32_t sk0 = (uint32_t)X.seed, sk1 = (uint32_t)(X.seed >> 32), gchain = X.chain0 + (uint32_t)c;
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
        // p0 ~ N(0, M) (hmc.rs:436-441): Box-Muller pair j of the chain's (iteration) Philox stream is block j, so the
        // pairs are drawn by whichever wave gets them -- the values do not depend on who draws
        const int n_pairs = (d + 1) >> 1;
#ifndef FG_EXP_NOMOM
        for (int j = wv; j < n_pairs; j += W) {
            const FgD2 zz = fg_cold_normal_pair(sk0, sk1, gchain, (uint32_t)j, (uint32_t)iter, FG_RNG_HMC);
            const double z0 = zz.a, z1 = zz.b;
            const int i = 2 * j;
            pl[i * tw] = z0 * (ms ? ms[(long long)i * X.C] : 1.0);
            if (i + 1 < d) pl[(i + 1) * tw] = z1 * (ms ? ms[(long long)(i + 1) * X.C] : 1.0);
        }
#endif
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
#ifndef FG_EXP_NOSTREAM
            if (dense) bad = fg_grad_dense_stream(P.sstream, P.n_sstream, k0, k1, slots, pl, tw, H.h, hk, gs > 0 && gs < L, nullptr, 0, false) || bad;
            else if (gn > 0) bad = fg_grad_stream<RK, AN>(gs0, gn, P.pool, slots, pl, tw, H.h, hk, gs > 0 && gs < L, nullptr, 0, false) || bad;
#endif
            if (!sep) __syncthreads();                           // every p kicked, every read of q done
            if (gs < L) {
                if (mi) { for (int k = k0; k < k1; ++k) slots[k * tw] += e * mi[(long long)k * X.C] * pl[k * tw]; }
                else    { for (int k = k0; k < k1; ++k) slots[k * tw] += e * pl[k * tw]; }      // eps * 1.0 * p == eps * p
                if (!sep) __syncthreads();
            }
        }
        xch[(2 + wv) * tw] = bad ? 1.0 : 0.0;
        __syncthreads();
        if (wv == 0) {
            // the endpoint score, accept and dual averaging are the tile's path: ahead of the other tiles' gradient waves (refmodel8 +6 %)
            // -- except with linear predictors, whose endpoint score is long enough to starve them (linreg -18 %, C3 -7 %)
            if (RK != 1) __builtin_amdgcn_s_setprio(2);
            bool div = false;
            for (int w = 0; w < W; ++w) div = div || xch[(2 + w) * tw] != 0.0;
            FgAcc3 A = {0.0, 0.0, 0.0};
#ifndef FG_EXP_NOSCORE
            if (SS) fg_score_stream<RK>(P.sstream, P.n_sstream, P.pool, slots, tw, A);                                       // score_full, hmc.rs:283-299
            else A = fg_cold_score_exec(P.ins_fast, P.n_ins, P.pool, slots, tw);
#endif
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
#ifndef FG_EXP_NODA
            if (warming) {                                       // DualAveraging::update: hmc.rs:168-178
                da_m += 1ull;
                const FgD3 r = fg_cold_da_update(da_hbar, da_leb, (double)da_m, da_mu, H.target, ap);
                eps = r.a; da_hbar = r.b; da_leb = r.c;
            }
#endif
            if (RK != 1) __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        const bool acc = xch[tw] != 0.0;
        unsigned long long wn = 0;
        if (warming && welford_on) wn = H.w_n[c] + 1ull;          // every wave reads the old count before wave 0 bumps it below
        for (int i = k0; i < k1; ++i) {                           // commit or roll back this wave's f64 sites
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

// hmc_transition with injected momentum / uniform (test hook)
template <bool GT>
__global__ __launch_bounds__(FG_WAVE, FG_MIN_WAVES) void k_hmc_transition_injected(FgProgramDev P, FgChainCtx X, FgHmcDev H, double eps,
                                                                     const double *p0, const double *u_in, int *acc_out,
                                                                     double *alpha_out, int *div_out) {
    extern __shared__ double lds_[];
    double *lds = GT ? X.gtile + (size_t)blockIdx.x * X.gtile_rows * FG_WAVE : lds_;        // GT: the tile lives in global memory (programs beyond 160 KB of LDS)
    constexpr int tw = FG_WAVE;                        // tile width: every lane of the wave owns a chain
    const long long chain = (long long)blockIdx.x * tw + threadIdx.x;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    double *slots = lds + threadIdx.x;
    double *pl = lds + (long long)P.n_slots * tw + threadIdx.x;
    fg_load_values(P, X, c, slots, tw);
    for (int i = 0; i < P.d; ++i) pl[i * tw] = p0[(long long)i * X.C + c];
    const FgTransOut o = fg_hmc_transition(P, X, H, c, live, slots, pl, tw, H.lj[c], eps, u_in[c]);
    if (live) {
        H.lj[c] = o.lj;
        if (acc_out) acc_out[c] = o.accepted;
        if (alpha_out) alpha_out[c] = o.alpha;
        if (div_out) div_out[c] = o.divergent;
    }
}

// grad_log_joint (hmc.rs:304-329) at the current values (test hook)
template <bool GT>
__global__ __launch_bounds__(FG_WAVE, FG_MIN_WAVES) void k_hmc_grad(FgProgramDev P, FgChainCtx X, double h, int sparse /* 2: analytic */, double *grad, int *ok) {
    extern __shared__ double lds_[];
    double *lds = GT ? X.gtile + (size_t)blockIdx.x * X.gtile_rows * FG_WAVE : lds_;        // GT: the tile lives in global memory (programs beyond 160 KB of LDS)
    constexpr int tw = FG_WAVE;                        // tile width: every lane of the wave owns a chain
    const long long chain = (long long)blockIdx.x * tw + threadIdx.x;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    double *slots = lds + threadIdx.x;
    fg_load_values(P, X, c, slots, tw);
    bool good = true;
    double orig = 0.0, lp_plus = 0.0;
    int slot = 0;
    if (sparse && P.gstream) {
        double *pl = lds + (long long)P.n_slots * tw + threadIdx.x;
        for (int i = 0; i < P.d; ++i) pl[i * tw] = 0.0;
        if (sparse == 2) good = !fg_grad_stream<1, true>(P.gstream, P.n_gstream, P.pool, slots, pl, tw, h, 0.0, false, grad + c, X.C, live);
        else good = !fg_grad_stream<2>(P.gstream, P.n_gstream, P.pool, slots, pl, tw, h, 0.0, false, grad + c, X.C, live);
        if (live && ok) ok[c] = good;
        return;
    }
    for (int e = 0; e < 2 * P.d; ++e) {
        const int i = e >> 1;
        const bool minus = (e & 1) != 0;
        if (!minus) { slot = i; orig = slots[slot * tw]; slots[slot * tw] = orig + h; }
        else slots[slot * tw] = orig - h;
        const FgIns *prog = P.ins_fast; int n = P.n_ins;
        if (sparse) { const int o0 = P.sub_off[i]; prog = P.sub + o0; n = P.sub_off[i + 1] - o0; }
        FgAcc3 A = {0.0, 0.0, 0.0};
        fg_exec<FG_MODE_SCORE, false>(prog, n, P.pool, slots, tw, A, nullptr, nullptr, 0, false);
        if (!minus) { lp_plus = fg_total(A); continue; }
        slots[slot * tw] = orig;
        const double g = (lp_plus - fg_total(A)) / (2.0 * h);
        good = good && fg_finite(g);
        if (live) grad[(long long)i * X.C + c] = g;
    }
    if (live && ok) ok[c] = good;
}

// find_reasonable_epsilon (hmc.rs:479-535).  Momentum comes from p0_scratch [d][C] when
// `injected`, else from the chain's (instance) EPS stream and is written there.
template <bool GT>
__global__ __launch_bounds__(FG_WAVE, FG_MIN_WAVES) void k_hmc_find_eps(FgProgramDev P, FgChainCtx X, FgHmcDev H, uint32_t instance, int injected,
                                                          double *eps_out) {
    extern __shared__ double lds_[];
    double *lds = GT ? X.gtile + (size_t)blockIdx.x * X.gtile_rows * FG_WAVE : lds_;        // GT: the tile lives in global memory (programs beyond 160 KB of LDS)
    constexpr int tw = FG_WAVE;                        // tile width: every lane of the wave owns a chain
    const long long chain = (long long)blockIdx.x * tw + threadIdx.x;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    double *slots = lds + threadIdx.x;
    double *pl = lds + (long long)P.n_slots * tw + threadIdx.x;
    const double *mi = H.use_mass ? H.m_inv + c : nullptr;
    const double *ms = H.use_mass ? H.mass_sqrt + c : nullptr;
    const bool sparse = H.grad_mode != FG_GRAD_FD_DENSE;       // the step-size search of FG_GRAD_ANALYTIC uses the sparse finite difference
    fg_load_values(P, X, c, slots, tw);
    if (!injected) {
        FgStream rng = fg_stream(X.seed, X.chain0 + (uint32_t)c, instance, FG_RNG_EPS);
        fg_draw_momentum(P, rng, pl, tw, ms, X.C);
        if (live) for (int i = 0; i < P.d; ++i) H.p0_scratch[(long long)i * X.C + c] = pl[i * tw];
    } else {
        for (int i = 0; i < P.d; ++i) pl[i * tw] = H.p0_scratch[(long long)i * X.C + c];
    }
    // keep p0 in registers-free storage: re-read it from LDS-resident copy is impossible once
    // the trajectory overwrites pl, so dead lanes (which did not write p0_scratch) re-derive
    // it the same way live lanes re-read it -- from the column of the chain they mirror.
    const double lj_q = H.lj[c];
    const double h0 = -lj_q + fg_kinetic(P, pl, tw, mi, X.C);
    const double ln_half = log(0.5), ln2 = log(2.0);
    double eps = 1.0, lr = 0.0, a = 1.0;
    bool active = true;     // lanes still inside the doubling/halving loop
    bool first = true;
    unsigned iters = 0;
    while (__any(active)) {
        // log_ratio_at(eps_try): one leapfrog step from (q, p0)      hmc.rs:500-511
        const double eps_try = first ? 1.0 : eps * ((a > 0.0) ? 2.0 : 0.5);   // eps * 2^a
        double lj1;
        const bool div = fg_trajectory(P, slots, pl, tw, eps_try, 1, H.h, sparse, mi, X.C, lj1);
        double lr_try = FG_NEG_INF;
        if (!div) lr_try = h0 - (-lj1 + fg_kinetic(P, pl, tw, mi, X.C));
        // restore (q, p0) for the next trial
        for (int i = 0; i < P.d; ++i) {
            slots[i * tw] = fg_as_double(X.values[(long long)P.f64_site[i] * X.C + c]);
            pl[i * tw] = H.p0_scratch[(long long)i * X.C + c];
        }
        if (first) {
            lr = lr_try;
            a = (lr > ln_half) ? 1.0 : -1.0;
            active = a * lr > -a * ln2;
            first = false;
        } else if (active) {
            eps = eps_try; lr = lr_try; iters += 1;
            if (iters > 100 || !(eps >= 1e-12 && eps <= 1e12)) active = false;
            else if (a > 0.0 && lr == FG_NEG_INF) { eps /= 2.0; active = false; }
            else active = a * lr > -a * ln2;
        }
    }
    if (live) eps_out[c] = fmin(fmax(eps, 1e-6), 1e3);
}

// mass-matrix reset at the warmup midpoint: Welford::variances + hmc.rs:886-894
__global__ void k_hmc_mass_reset(FgHmcDev H, long long C, int d) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const unsigned long long n = H.w_n[c];
    for (int i = 0; i < d; ++i) {
        const long long g = (long long)i * C + c;
        double v = 1.0;
        if (n >= 2) { const double vv = H.w_m2[g] / (double)(n - 1); v = (fg_finite(vv) && vv > 1e-8) ? vv : 1.0; }
        H.m_inv[g] = v;
        H.mass_sqrt[g] = sqrt(1.0 / v);
    }
}
// DualAveraging::new for every chain: hmc.rs:153-164
__global__ void k_hmc_da_new(FgHmcDev H, long long C, const double *eps0) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double e = eps0[c];
    H.eps[c] = e; H.da_mu[c] = log(10.0 * e); H.da_leb[c] = 0.0; H.da_hbar[c] = 0.0; H.da_m[c] = 0ull;
}


// ======================================================================================
// single-site adaptive Metropolis-Hastings (src/inference/mh.rs:698-744, 938-1014)
// ======================================================================================
// Per-chain state in HBM:  lw [C] = log-weight of the current trace;  per (site, chain):
// scale, log_scale [S][C] f64, acc, tot [S][C] u32 (DiminishingAdaptation, mcmc_utils.rs:30-42),
// kind [S][C] i32 (the f64 proposal kind cache, mh.rs:330,947).

// adaptive_mcmc_chain's step loop: n_steps x single_site_mh_step (mh.rs:698-744), exactly one
// model run per step (mh.rs:1186-1202).  Recorded draws: [t][r][C] cells of the CURRENT state
// after each sampling-phase step (mh.rs:1010).
template <bool GT>
__global__ __launch_bounds__(FG_WAVE, FG_MIN_WAVES) void k_mh_steps(FgProgramDev P, FgChainCtx X, FgMhDev M, int iter0, int n_steps,
                                                                    int n_warmup, long long *draws, int first_sample_t) {
    extern __shared__ double lds_[];
    double *lds = GT ? X.gtile + (size_t)blockIdx.x * X.gtile_rows * FG_WAVE : lds_;        // GT: the tile lives in global memory (programs beyond 160 KB of LDS)
    constexpr int tw = FG_WAVE;
    const long long chain = (long long)blockIdx.x * tw + threadIdx.x;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    double *slots = lds + threadIdx.x;
    fg_load_values(P, X, c, slots, tw);
    double lw = M.lw[c];
    unsigned long long nacc = 0;
    for (int t = 0; t < n_steps; ++t) {
        const int iter = iter0 + t;
        const bool adapt = iter < n_warmup;
        FgMhCtx mh;
        FgStream rng = fg_stream(X.seed, X.chain0 + (uint32_t)c, (uint32_t)iter, FG_RNG_MH);
        unsigned long long ra, rb;
        fg_rng_block(rng, ra, rb);
        const int target = (int)fg_pick(ra, (uint32_t)P.S);               // sites[rng.gen_range(0..len)]  mh.rs:716
        const long long g = (long long)target * X.C + c;
        const int tslot = P.site_slot[target];                             // per-lane gather (site -> LDS slot)
        mh.target = tslot;
        { const fg_u32x4 a0 = *(const fg_u32x4 *)(M.ad + g); mh.scale = fg_dbl(a0[0], a0[1]); mh.kind = (int)a0[2]; }   // get_scale  mcmc_utils.rs:70-77
        const int kind0 = mh.kind;
        mh.rng = rng;                                                      // at block 1
        fg_rng_block(rng, ra, rb);
        mh.z = fg_cold_gaussian_z(ra, rb);
        mh.next_block = 2;
        mh.lqf = 0.0; mh.lqr = 0.0;
        mh.ov_kind = M.ov_kind; mh.ov_lo = M.ov_lo; mh.ov_hi = M.ov_hi;
        mh.old_cell = slots[tslot * tw];
        FgAcc3 A = {0.0, 0.0, 0.0};
        // Random-walk proposals (mh.rs:183-294, 557-567) need nothing from the model: the new value, log q(x'|x) and
        // log q(x|x') depend only on the current value, the adapted scale, the decided kind and (Reflect) the
        // override bounds.  When EVERY lane of the wave holds such a site the proposal is made here, per lane, and
        // the model is then only SCORED (score stream / score-only opcodes) instead of being driven through the
        // propose-and-score mode, whose proposal machinery would otherwise run under a partial mask at almost every
        // sample site (64 lanes pick 64 different sites).  Undecided kinds, prior-resample kinds and Categorical
        // sites take the general path below; both paths produce the same values.
        const uint32_t tv = (uint32_t)P.site_vtype[target];
        int kind_eff = FG_PROP_AUTO;
        if (tv == 0u) { kind_eff = mh.ov_kind ? mh.ov_kind[tslot] : FG_PROP_AUTO; if (kind_eff == FG_PROP_AUTO) kind_eff = mh.kind; }
        const int cat_base = P.site_cat[2 * target], cat_K = P.site_cat[2 * target + 1];
        const bool walk = tv == 0u ? (kind_eff == FG_PROP_GAUSSIAN || kind_eff == FG_PROP_LOGSPACE || kind_eff == FG_PROP_REFLECT)
                                   : (tv == 1u || tv == 2u || tv == 4u || (tv == 3u && cat_K > 0));
        if (__all(walk)) {
            if (tv == 3u) {                                   // usize target: resample from the constant prior table (mh.rs:516-530)
                FgStream s1 = mh.rng;
                const double uu = fg_rng_u01(s1);
                double cum = 0.0; int idx = cat_K;
                for (int i = 0; i < cat_K; ++i) { cum += P.pool[cat_base + i]; if (idx == cat_K && !(cum < uu)) idx = i; }
                const long long prop = idx < cat_K - 1 ? idx : cat_K - 1;
                const long long cur = fg_as_i64(mh.old_cell);
                // prior log-probabilities of the proposed and the current index: the table's precomputed ln p (-inf for p <= 0)
                mh.lqf += P.pool[cat_base + cat_K + (int)prop];
                mh.lqr += (cur < 0 || cur >= (long long)cat_K) ? FG_NEG_INF : P.pool[cat_base + cat_K + (int)cur];
                mh.next_block = (int)s1.c1;
                slots[tslot * tw] = fg_as_double(prop);
            } else fg_mh_walk_proposal(mh, tv, kind_eff, tslot, slots, tw);
            if (P.sstream && P.sstream_kinds == 0) fg_score_stream<0>(P.sstream, P.n_sstream, P.pool, slots, tw, A);
            else if (P.sstream) fg_score_stream<2>(P.sstream, P.n_sstream, P.pool, slots, tw, A);
            else A = fg_cold_score_exec(P.ins_fast, P.n_ins, P.pool, slots, tw);
        } else
            A = fg_cold_mh_exec(P.ins, P.n_ins, P.pool, slots, tw, live, &mh);   // propose_and_score
        const double prop_lw = fg_total(A);
        const double log_alpha = prop_lw - lw + (mh.lqr - mh.lqf);         // + dim_term == 0 (fixed structure)  mh.rs:731-732
        const double u = fg_cold_u01_pair((uint32_t)X.seed, (uint32_t)(X.seed >> 32), X.chain0 + (uint32_t)c, (uint32_t)mh.next_block, (uint32_t)iter, FG_RNG_MH).a;   // only consulted when log_alpha < 0
        const bool accept = (log_alpha >= 0.0) || (u < fg_cold_exp(log_alpha));    // mh.rs:733
        if (adapt) {                                                       // DiminishingAdaptation::update  mcmc_utils.rs:88-150
            const fg_u32x4 a1 = *(const fg_u32x4 *)((const char *)(M.ad + g) + 16);
            const uint32_t tot = a1[2] + 1u;
            const uint32_t acn = a1[3] + (accept ? 1u : 0u);
            double sc = mh.scale, ls = fg_dbl(a1[0], a1[1]);
            if (tot >= 10u) { const FgD2 r = fg_cold_mh_adapt(ls, acn, tot, M.step_tab, M.step_n); sc = r.a; ls = r.b; }
            if (live) {
                const unsigned long long lb = (unsigned long long)__double_as_longlong(ls);
                const fg_u32x4 w1 = { (uint32_t)lb, (uint32_t)(lb >> 32), tot, acn };
                *(fg_u32x4 *)((char *)(M.ad + g) + 16) = w1;
                M.ad[g].scale = sc;
            }
        }
        if (live && mh.kind != kind0) M.ad[g].kind = mh.kind;
        if (accept) { lw = prop_lw; nacc += 1ull; if (live) X.values[g] = fg_as_i64(slots[tslot * tw]); }
        else slots[tslot * tw] = mh.old_cell;
        if ((!adapt || M.rec_all) && draws && live) {
            long long *row = draws + (long long)(t - first_sample_t) * M.n_rec * X.C + c;
            for (int r = 0; r < M.n_rec; ++r) row[(long long)r * X.C] = fg_as_i64(slots[M.rec[r] * tw]);
        }
    }
    if (live) { M.lw[c] = lw; M.n_acc[c] += nacc; }
}

// ======================================================================================
// engine
// ======================================================================================
// one-wave-per-tile kernels: the tile in LDS, or -- programs whose tile exceeds the 160 KB of a CU -- in a global scratch
// [tiles][rows][64] (same addressing through generic pointers; L2-resident for moderate chain counts).  Such programs run
// everything on these kernels: the reference's hmc_chain / adaptive_mcmc_chain have no size limit (hmc.rs:238-260).
#define FG_LAUNCH_GT(e, K, GRID, BLOCK, LDS, STREAM, ...) do { if ((e)->gt) hipLaunchKernelGGL((K<true>), GRID, BLOCK, 0, STREAM, __VA_ARGS__); \
                                                               else hipLaunchKernelGGL((K<false>), GRID, BLOCK, LDS, STREAM, __VA_ARGS__); } while (0)
extern "C" {

void fg_hmc_config_default(fg_hmc_config *c) {     // HMCConfig::default, hmc.rs:125-135
    if (!c) return;
    c->n_leapfrog = 16; c->target_accept = 0.8; c->init_step_size = NAN; c->finite_diff_eps = 1e-5;
    c->adapt_mass = 0; c->grad_mode = FG_GRAD_FD_SPARSE;    // the engine's one default (C ABI, Python mirror, bench); FG_GRAD_FD_DENSE = the reference verbatim
}

fg_engine *fg_engine_new(const fg_program *p, int64_t n_chains, uint64_t seed, uint32_t chain_offset, int device) {
    if (!p || !p->finalized) { fg_set_error("fg_engine_new: program is not finalized"); return nullptr; }
    if (n_chains <= 0) { fg_set_error("fg_engine_new: n_chains must be positive"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fg_set_error("no HIP device available: the engine has no CPU fallback (FG_E_NO_DEVICE)"); return nullptr; }
    if (device < 0 || device >= ndev) { fg_set_error("fg_engine_new: bad device ordinal"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fg_set_error("hipSetDevice failed"); return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { fg_set_error("hipGetDeviceProperties failed"); return nullptr; }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fg_set_error(std::string("device arch ") + prop.gcnArchName + " is not gfx950; kernels are built for MI355X only");
        return nullptr; }
    fg_engine *e = new fg_engine();
    e->prog = p; e->device = device; e->C = n_chains; e->seed = seed; e->chain0 = chain_offset;
    e->S = (int)p->sorted_stmt.size(); e->d = (int)p->f64_slot.size(); e->n_slots = p->n_slots;
    e->tw = tile_width_for(e->C);
    e->n_simd = 4 * prop.multiProcessorCount;
    if (const char *mw = std::getenv("FG_HMC_WAVES")) { const int w = std::atoi(mw); if (w == 1 || w == 2 || w == 4 || w == 8 || w == 16) e->mw_override = w; }
    if (const char *sp = std::getenv("FG_HMC_SEP")) e->sep_disabled = std::atoi(sp) == 0;
    if (const char *sp = std::getenv("FG_HMC_LIN")) e->lin_disabled = std::atoi(sp) == 0;
    if (const char *sp = std::getenv("FG_HMC_INTERP_MW")) e->interp_mw_disabled = std::atoi(sp) == 0;
    if (const char *sp = std::getenv("FG_MH_MW")) e->mh_mw_disabled = std::atoi(sp) == 0;
    // LDS tile of 64 chains: the score / prior / MH / SMC kernels need the slot rows only, the HMC kernels also the
    // momentum and the multi-wave exchange rows.  A model whose slots alone exceed the 160 KB of a CU cannot run at all;
    // one that only fits without the momentum is refused by fg_hmc_init.
    e->lds_score = (size_t)e->n_slots * e->tw * sizeof(double);
    e->lds_bytes = (size_t)(e->n_slots + e->d + 2 + FG_MW_MAX) * e->tw * sizeof(double);
    // beyond one CU's LDS: the tile goes to a global scratch and every kernel of this engine is the one-wave-per-tile form (GT)
    e->gt = e->lds_bytes > 160 * 1024;
    if (const char *gv = std::getenv("FG_GLOBAL_TILE")) e->gt = e->gt || std::atoi(gv) != 0;      // tests: any program through the global-tile instantiations
    const size_t lds_hmc = e->gt ? 0 : std::min<size_t>(e->lds_bytes, 160 * 1024), lds_sc = e->gt ? 0 : e->lds_score;
    auto fail = [&](const char *what) { fg_set_error(std::string("fg_engine_new: ") + what + ": " + fg_last_error()); fg_engine_free(e); return (fg_engine *)nullptr; };
    if (hipStreamCreate(&e->stream) != hipSuccess) return fail("hipStreamCreate");
    if (e->gt) {
        e->X.gtile_rows = e->n_slots + e->d + 2 + FG_MW_MAX;
        if (dev_alloc(&e->d_gtile, (size_t)((e->C + e->tw - 1) / e->tw) * e->X.gtile_rows * e->tw)) return fail("alloc global tile");
        e->X.gtile = e->d_gtile;
    }
    if (dev_upload(&e->d_ins, p->ins)) return fail("upload ins");
    if (dev_upload(&e->d_sub, p->sub)) return fail("upload sub");
    if (dev_upload(&e->d_ins_fast, p->ins_fast)) return fail("upload ins_fast");
    if (dev_upload(&e->d_coord, p->coord)) return fail("upload coord");
    if (dev_upload(&e->d_gstream, p->gstream)) return fail("upload gstream");
    if (dev_upload(&e->d_sstream, p->sstream)) return fail("upload sstream");
    if (dev_upload(&e->d_sep, p->sep) || dev_upload(&e->d_sep_coord, p->sep_coord) || dev_upload(&e->d_sep_free, p->sep_free) || dev_upload(&e->d_site_rec, p->site_rec) || dev_upload(&e->d_sobs, p->sobs)) return fail("upload sep");
    if (!p->lin_tab.empty() && (dev_upload(&e->d_lin_tab, p->lin_tab) || dev_upload(&e->d_lin_meta, p->lin_meta))) return fail("upload lin");
    if (dev_upload(&e->d_sub_off, p->sub_off)) return fail("upload sub_off");
    if (dev_upload(&e->d_f64_slot, p->f64_slot)) return fail("upload f64_slot");
    if (dev_upload(&e->d_site_slot, p->site_slot)) return fail("upload site_slot");
    if (dev_upload(&e->d_vtype, p->site_vtype)) return fail("upload vtype");
    if (dev_upload(&e->d_site_cat, p->site_cat)) return fail("upload site_cat");
    if (dev_upload(&e->d_pool, p->pool)) return fail("upload pool");
    if (dev_alloc(&e->d_values, (size_t)std::max(1, e->S) * e->C)) return fail("alloc values");
    if (dev_alloc(&e->d_acc, (size_t)3 * e->C)) return fail("alloc acc");
    if (dev_alloc(&e->d_tmp, (size_t)e->C)) return fail("alloc tmp");
    if (dev_alloc(&e->d_itmp, (size_t)3 * e->C)) return fail("alloc itmp");
    e->P.ins = e->d_ins; e->P.ins_fast = e->d_ins_fast; e->P.coord = e->d_coord; e->P.gstream = p->n_gstream > 0 ? e->d_gstream : nullptr; e->P.n_gstream = p->n_gstream;
    e->P.sep = p->sep_coord.empty() ? nullptr : e->d_sep; e->P.sep_coord = e->d_sep_coord; e->P.sobs = e->d_sobs;
    e->P.sep_free = e->d_sep_free; e->P.n_sep_free = (int)p->sep_free.size(); e->P.n_prior_terms = p->n_prior_terms; e->P.site_rec = e->d_site_rec;
    e->P.sstream = p->n_sstream > 0 ? e->d_sstream : nullptr; e->P.n_sstream = p->n_sstream; e->P.sstream_kinds = p->sstream_has_gen ? 2 : (p->sstream_has_lin ? 1 : 0); e->P.sstream_gen = p->sstream_has_genrec ? 1 : 0; e->P.sub = e->d_sub; e->P.sub_off = e->d_sub_off; e->P.pool = e->d_pool;
    e->P.f64_site = e->d_f64_slot; e->P.site_slot = e->d_site_slot; e->P.site_vtype = e->d_vtype; e->P.site_cat = e->d_site_cat;
    e->P.n_ins = p->n_ins; e->P.n_slots = e->n_slots; e->P.S = e->S; e->P.d = e->d;
    e->P.lin_tab = e->d_lin_tab; e->P.lin_meta = e->d_lin_meta; e->P.lin_n = p->lin_n; e->P.lin_p2 = p->lin_p2;
    e->X.C = e->C; e->X.chain0 = e->chain0; e->X.seed = e->seed; e->X.values = e->d_values;
    if (set_lds(k_prior_init<false>, lds_sc) || set_lds(k_log_joint<false>, lds_sc) || set_lds(k_log_joint_stream<false>, lds_sc) ||
        set_lds(k_hmc_steps<false>, lds_hmc) || set_lds(k_hmc_stream_steps<0, false, true>, lds_hmc) || set_lds(k_hmc_stream_steps<1, false, true>, lds_hmc) || set_lds(k_hmc_stream_steps<2, false, true>, lds_hmc) ||
        set_lds(k_hmc_stream_steps<0, true, true>, lds_hmc) || set_lds(k_hmc_stream_steps<1, true, true>, lds_hmc) ||
        set_lds(k_hmc_stream_steps<0, false, false>, lds_hmc) || set_lds(k_hmc_stream_steps<1, false, false>, lds_hmc) || set_lds(k_hmc_stream_steps<2, false, false>, lds_hmc) ||
        set_lds(k_hmc_stream_steps<0, true, false>, lds_hmc) || set_lds(k_hmc_stream_steps<1, true, false>, lds_hmc) || set_lds(k_hmc_transition_injected<false>, lds_hmc) ||
        set_lds(k_hmc_grad<false>, lds_hmc) || set_lds(k_hmc_find_eps<false>, lds_hmc) ||
        set_lds(k_mh_steps<false>, lds_sc))
        return fail("hipFuncSetAttribute");
    return e;
}

void fg_engine_free(fg_engine *e) {
    if (!e) return;
    hipSetDevice(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    for (void *q : e->hmc_allocs) hipFree(q);
    for (void *q : e->mh_allocs) hipFree(q);
    if (e->d_rec) hipFree(e->d_rec);
    if (e->smc_arena) hipFree(e->smc_arena);
    if (e->smc_host) (void)hipHostFree(e->smc_host);
    if (e->jit_mod) (void)hipModuleUnload(e->jit_mod);
    if (e->jit_mh_mod) (void)hipModuleUnload(e->jit_mh_mod);
    if (e->jit_mhmw_mod) (void)hipModuleUnload(e->jit_mhmw_mod);
    if (e->jit_mhns_mod) (void)hipModuleUnload(e->jit_mhns_mod);
    void *ptrs[] = { e->d_mh_catu_c, e->d_mh_catu, e->d_jit_tab, e->d_jit_mh_tab, e->d_jit_mhmw_tab, e->d_jit_mhns_tab, e->d_mhi_acc, e->d_mhi_site_ins, e->d_mwi_order, e->d_mwi_prof, e->d_gtile, e->d_lin_tab, e->d_lin_meta, e->d_mh_srt, e->d_sep, e->d_sep_coord, e->d_sep_free, e->d_site_rec, e->d_sobs, e->d_ins, e->d_ins_fast, e->d_coord, e->d_gstream, e->d_sstream, e->d_sub, e->d_sub_off, e->d_f64_slot, e->d_site_slot, e->d_vtype, e->d_site_cat, e->d_pool, e->d_values, e->d_acc, e->d_logp,
                     e->d_tmp, e->d_itmp };
    for (void *q : ptrs) if (q) hipFree(q);
    if (e->stream && e->own_stream) hipStreamDestroy(e->stream);
    delete e;
}


int fg_engine_synchronize(fg_engine *e) { NEED_ENGINE(e); HIPCHK(hipStreamSynchronize(e->stream)); return FG_OK; }
void *fg_engine_stream(fg_engine *e) { return e ? (void *)e->stream : nullptr; }
int fg_engine_set_stream(fg_engine *e, void *hip_stream) {
    NEED_ENGINE(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->own_stream && e->stream) HIPCHK(hipStreamDestroy(e->stream));
    e->stream = (hipStream_t)hip_stream; e->own_stream = false;
    return FG_OK;
}
int64_t fg_engine_n_chains(const fg_engine *e) { return e ? e->C : 0; }
void *fg_engine_values_device(fg_engine *e) { return e ? (void *)e->d_values : nullptr; }

int fg_engine_set_values(fg_engine *e, const void *h) {
    NEED_ENGINE(e);
    if (!h) return FG_E_BAD_ARG;
    HIPCHK(hipMemcpyAsync(e->d_values, h, (size_t)e->S * e->C * 8, hipMemcpyHostToDevice, e->stream));
    // a live sampler session caches the log-joint of its current state: re-score it at the new values (what the reference's
    // callers do after editing a trace: crates/fugue-wasm/src/mh.rs:239-255 runs ScoreGivenTrace)
    for (double *lj : { e->mh_ready ? e->M.lw : (double *)nullptr, e->hmc_ready ? e->H.lj : (double *)nullptr })
        if (lj && (e->gt || fg_jit_log_joint_launch(e, e->d_acc, lj, false) != FG_OK))
            FG_LAUNCH_GT(e, k_log_joint, dim3((unsigned)((e->C + e->tw - 1) / e->tw)), dim3(e->tw), e->lds_score, e->stream, e->P, e->X, e->d_acc,
                                   (double *)nullptr, lj);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return FG_OK;
}
int fg_engine_get_values(fg_engine *e, void *h) {
    NEED_ENGINE(e);
    if (!h) return FG_E_BAD_ARG;
    HIPCHK(hipMemcpyAsync(h, e->d_values, (size_t)e->S * e->C * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return FG_OK;
}

void *fg_device_alloc(fg_engine *e, size_t bytes) {
    if (!e || hipSetDevice(e->device) != hipSuccess) return nullptr;
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { fg_set_error("hipMalloc failed"); return nullptr; }
    return p;
}
int fg_device_free(fg_engine *e, void *p) { NEED_ENGINE(e); HIPCHK(hipStreamSynchronize(e->stream)); HIPCHK(hipFree(p)); return FG_OK; }
int fg_device_download(fg_engine *e, void *h, const void *d, size_t bytes) {
    NEED_ENGINE(e);
    HIPCHK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return FG_OK;
}
int fg_device_upload(fg_engine *e, void *d, const void *h, size_t bytes) {
    NEED_ENGINE(e);
    HIPCHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return FG_OK;
}

int fg_launch_prior(fg_engine *e, uint32_t iteration, uint32_t purpose, double *d_acc, double *d_lj) {
    // the compiled model where there is one (fg_hmc_jit_body.h: k_prior_jit -- the same draws).  A program without a record stream will
    // step through the compiled unit anyway: it is built here, one call earlier; other programs use it only once it exists (adaptive_smc
    // asks for it when the population is large: fg_smc.hip)
    if (!e->gt) {
        const int rj = fg_jit_prior_launch(e, iteration, purpose, d_acc, d_lj, e->prog->n_gstream == 0 && e->prog->n_sstream == 0);
        if (rj == FG_OK) return FG_OK;
        if (rj != FG_E_UNSUPPORTED) return rj;
    }
    FG_LAUNCH_GT(e, k_prior_init, dim3((unsigned)((e->C + e->tw - 1) / e->tw)), dim3(e->tw), e->lds_score, e->stream, e->P, e->X, iteration,
                       purpose, d_acc, d_lj);
    HIPCHK(hipGetLastError());
    return FG_OK;
}

int fg_prior_init(fg_engine *e, uint32_t iteration, double *h_acc) {
    NEED_ENGINE(e);
    int rc = fg_launch_prior(e, iteration, FG_RNG_PRIOR, e->d_acc, nullptr);
    if (rc) return rc;
    if (h_acc) HIPCHK(hipMemcpyAsync(h_acc, e->d_acc, (size_t)3 * e->C * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return FG_OK;
}

int fg_log_joint(fg_engine *e, double *h_acc, double *h_logp) {
    NEED_ENGINE(e);
    if (h_logp && !e->d_logp) { int rc = dev_alloc(&e->d_logp, (size_t)std::max(1, e->S) * e->C); if (rc) return rc; }
    FG_LAUNCH_GT(e, k_log_joint, dim3((unsigned)((e->C + e->tw - 1) / e->tw)), dim3(e->tw), e->lds_score, e->stream, e->P, e->X, e->d_acc,
                       h_logp ? e->d_logp : nullptr, (double *)nullptr);
    HIPCHK(hipGetLastError());
    if (h_acc) HIPCHK(hipMemcpyAsync(h_acc, e->d_acc, (size_t)3 * e->C * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (h_logp)     // the kernel indexes rows by LDS slot; hand them back in site order
        for (int j = 0; j < e->S; j++)
            HIPCHK(hipMemcpy(h_logp + (size_t)j * e->C, e->d_logp + (size_t)e->prog->site_slot[j] * e->C, (size_t)e->C * 8, hipMemcpyDeviceToHost));
    return FG_OK;
}

int fg_log_joint_stream(fg_engine *e, double *h_acc, double *h_rec_lp) {
    NEED_ENGINE(e);
    if (!e->P.sstream) { fg_set_error("fg_log_joint_stream: the program has no score stream (some statement needs the interpreter)"); return FG_E_UNSUPPORTED; }
    double *d_rec = nullptr;
    const size_t nrec = (size_t)e->P.n_sstream * e->C;
    if (h_rec_lp) { int rc = dev_alloc(&d_rec, nrec); if (rc) return rc; }
    FG_LAUNCH_GT(e, k_log_joint_stream, dim3((unsigned)((e->C + e->tw - 1) / e->tw)), dim3(e->tw), e->lds_score, e->stream, e->P, e->X, e->d_acc, d_rec);
    hipError_t he = hipGetLastError();
    if (he == hipSuccess && h_acc) he = hipMemcpyAsync(h_acc, e->d_acc, (size_t)3 * e->C * 8, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess && h_rec_lp) he = hipMemcpyAsync(h_rec_lp, d_rec, nrec * 8, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (d_rec) (void)hipFree(d_rec);
    if (he != hipSuccess) { fg_set_error(hipGetErrorString(he)); return FG_E_HIP; }
    return FG_OK;
}

// ------------------------------------------------------------------ HMC host side
int fg_internal_hmc_alloc(fg_engine *e, bool mass) {
    auto A = [&](auto **p, size_t n) { int rc = dev_alloc(p, n); if (!rc) e->hmc_allocs.push_back((void *)*p); return rc; };
    size_t C = (size_t)e->C, d = (size_t)std::max(1, e->d);
    if (!e->H.lj) {
        if (A(&e->H.lj, C) || A(&e->H.eps, C) || A(&e->H.frozen, C) || A(&e->H.da_mu, C) || A(&e->H.da_leb, C) ||
            A(&e->H.da_hbar, C) || A(&e->H.da_m, C) || A(&e->H.alpha_sum, C) || A(&e->H.n_div, C) || A(&e->H.p0_scratch, d * C))
            return FG_E_HIP;
    }
    if (mass && !e->H.m_inv) {
        if (A(&e->H.m_inv, d * C) || A(&e->H.mass_sqrt, d * C) || A(&e->H.w_mean, d * C) || A(&e->H.w_m2, d * C) || A(&e->H.w_n, C))
            return FG_E_HIP;
    }
    return FG_OK;
}

void fg_internal_hmc_set_cfg(fg_engine *e, const fg_hmc_config *cfg) {
    e->cfg = *cfg;
    e->H.L = cfg->n_leapfrog > 1 ? cfg->n_leapfrog : 1;        // hmc.rs:684
    e->H.h = cfg->finite_diff_eps; e->H.target = cfg->target_accept;
    e->H.grad_mode = cfg->grad_mode;
}

// does fg_hmc_step run this program through the kernel compiled at run time (hmc_launch_steps' order: independent sites, dense regressions,
// then the compiled form where it is the faster one, the stream kernel, the compiled form, the interpreter)?
static bool hmc_jit_preferred(const fg_engine *e) {
    if (e->cfg.grad_mode != FG_GRAD_FD_SPARSE || e->jit_state < 0) return false;
    if (!e->P.gstream) return true;
    if (e->gt || e->tw != FG_WAVE) return false;
    if (e->P.sep && !e->sep_disabled && e->d >= 1) return false;                                     // fg_hmc_sep_launch takes it
    if (e->P.lin_tab && !e->lin_disabled && e->d >= 2 && e->d <= 64) return false;   // fg_hmc_lin_launch takes it
    // Every other gradient-stream program, at every chain count.  Round 3 sent only linear-predictor / general / option-select records here, round 4 first
    // added programs of fewer than eight coordinates and launches of two tiles per CU or fewer (the stream kernel kept 16 % on reference_model(8) at
    // 65 536 chains).  Since the unit holds its task split as straight-line code per wave (fg_jit_wave_tasks: no task list in memory, no dispatch on the
    // coordinate, short sub-programs inlined) it wins everywhere measured -- reference_model(8) 1.85e10 -> 2.26e10 leapfrog-steps/s at 65 536 chains,
    // 2.48e10 -> 2.57e10 at 524 288; reference_model(20) 8.4e9 -> 1.07e10; reference_model(32) 4.5e9 -> 7.2e9 (profiles/round4_jit_vs_stream_tasks.txt).
    // FG_JIT=0 keeps k_hmc_stream_steps (bit-identity tests, a box without hiprtc).
    return true;
}

static int hmc_find_eps(fg_engine *e, uint32_t instance, int injected, double *d_eps_out) {
    if (hmc_jit_preferred(e)) {                              // the step-size search on the compiled kernel too (k_hmc_jit_find_eps)
        const int rc = fg_hmc_jit_find_eps(e, instance, injected, d_eps_out);
        if (rc != FG_E_UNSUPPORTED) return rc;
    }
    FG_LAUNCH_GT(e, k_hmc_find_eps, dim3((unsigned)((e->C + e->tw - 1) / e->tw)), dim3(e->tw), e->lds_bytes, e->stream, e->P, e->X, e->H, instance,
                       injected, d_eps_out);
    HIPCHK(hipGetLastError());
    return FG_OK;
}

static int hmc_lds_ok(const fg_engine *e) {
    if (e->lds_bytes <= 160 * 1024 || e->gt) return FG_OK;
    fg_set_error("HMC needs sites + temporaries + momentum in one 160 KB LDS tile (2 * f64 sites + other slots + 6 > 320 cells); "
                 "adaptive_mcmc_chain / adaptive_smc still run for this model");
    return FG_E_LIMIT;
}

int fg_hmc_init(fg_engine *e, const fg_hmc_config *cfg, int n_warmup) {
    NEED_ENGINE(e);
    if (!cfg || n_warmup < 0) return FG_E_BAD_ARG;
    if (int rc = hmc_lds_ok(e)) return rc;
    if (cfg->grad_mode != FG_GRAD_FD_DENSE && cfg->grad_mode != FG_GRAD_FD_SPARSE && cfg->grad_mode != FG_GRAD_ANALYTIC) { fg_set_error("unknown grad_mode"); return FG_E_BAD_ARG; }
    // FG_GRAD_ANALYTIC: closed forms in the stream / register-resident kernels when every force term is a Normal with constant sigma;
    // any other program through the forward-mode derivative of its sub-programs in the unit compiled at run time (fg_jit.cpp: Gen::ins_ad)
    e->an_jit = false;
    if (cfg->grad_mode == FG_GRAD_ANALYTIC && (e->d == 0 || !e->P.gstream || fg_program_stream_records(e->prog, 2) > 1 || e->tw != FG_WAVE)) {
        if (e->d == 0 || e->tw != FG_WAVE || e->gt || !fg_hmc_jit_has_ad(e)) {
            fg_set_error("FG_GRAD_ANALYTIC: no closed form for this program (force terms other than Normals with constant sigma need the run-time compiler: hiprtc / hipcc)");
            return FG_E_UNSUPPORTED; }
        e->an_jit = true;
    }
    if (cfg->n_leapfrog < 1 || cfg->n_leapfrog > 100000) { fg_set_error("n_leapfrog must be in [1, 100000] (a trajectory of 0 steps never moves: hmc.rs:385)"); return FG_E_BAD_ARG; }
    if (!(cfg->finite_diff_eps > 0.0) || !std::isfinite(cfg->finite_diff_eps)) { fg_set_error("finite_diff_eps must be positive and finite"); return FG_E_BAD_ARG; }
    const bool mass = cfg->adapt_mass && n_warmup >= 4;                 // hmc.rs:704-708
    int rc = fg_internal_hmc_alloc(e, mass);
    if (rc) return rc;
    fg_internal_hmc_set_cfg(e, cfg);
    e->H.use_mass = mass ? 1 : 0;
    e->n_warmup = n_warmup; e->iter = 0; e->mass_adapt_at = mass ? n_warmup / 2 : -1;
    const int TB = 256, NB = (int)((e->C + TB - 1) / TB);
    size_t C = (size_t)e->C;
    HIPCHK(hipMemsetAsync(e->H.alpha_sum, 0, C * 8, e->stream));
    HIPCHK(hipMemsetAsync(e->H.n_div, 0, C * 8, e->stream));
    hipLaunchKernelGGL(k_fill, dim3(NB), dim3(TB), 0, e->stream, e->H.frozen, (long long)C, (double)NAN);
    if (mass) {
        size_t dC = (size_t)e->d * C;
        const int NBd = (int)((dC + TB - 1) / TB);
        hipLaunchKernelGGL(k_fill, dim3(NBd), dim3(TB), 0, e->stream, e->H.m_inv, (long long)dC, 1.0);
        hipLaunchKernelGGL(k_fill, dim3(NBd), dim3(TB), 0, e->stream, e->H.mass_sqrt, (long long)dC, 1.0);
        HIPCHK(hipMemsetAsync(e->H.w_mean, 0, dC * 8, e->stream));
        HIPCHK(hipMemsetAsync(e->H.w_m2, 0, dC * 8, e->stream));
        HIPCHK(hipMemsetAsync(e->H.w_n, 0, C * 8, e->stream));
    }
    rc = fg_launch_prior(e, 0, FG_RNG_PRIOR, nullptr, e->H.lj);             // hmc.rs:673-687
    if (rc) return rc;
    if (e->d == 0) hipLaunchKernelGGL(k_fill, dim3(NB), dim3(TB), 0, e->stream, e->d_tmp, (long long)C, 1.0);
    else if (!std::isnan(cfg->init_step_size)) hipLaunchKernelGGL(k_fill, dim3(NB), dim3(TB), 0, e->stream, e->d_tmp, (long long)C, cfg->init_step_size);
    else { rc = hmc_find_eps(e, 0, 0, e->d_tmp); if (rc) return rc; }
    hipLaunchKernelGGL(k_hmc_da_new, dim3(NB), dim3(TB), 0, e->stream, e->H, (long long)C, (const double *)e->d_tmp);
    HIPCHK(hipGetLastError());
    e->hmc_ready = true;
    return FG_OK;
}

static int hmc_launch_steps(fg_engine *e, int iter0, int n, int welford_on, double *draws, int first_sample_t,
                            double *pos_all = nullptr, double *info = nullptr) {
    const unsigned tiles = (unsigned)((e->C + e->tw - 1) / e->tw);
    const bool dense_stream = e->cfg.grad_mode == FG_GRAD_FD_DENSE && e->P.sstream != nullptr && e->P.sstream_kinds == 0;
    const bool analytic = e->cfg.grad_mode == FG_GRAD_ANALYTIC;
    if (analytic && e->an_jit) {                              // the analytic gradient of a program without closed-form records: the compiled unit
        const int rc = fg_hmc_jit_launch(e, iter0, n, welford_on, draws, first_sample_t, pos_all, info);
        if (rc != FG_OK) fg_set_error("FG_GRAD_ANALYTIC: the compiled kernel is not available for this launch");
        return rc == FG_E_UNSUPPORTED ? FG_E_STATE : rc;
    }
    {   // independent-sites programs: whole trajectories in registers (fg_hmc_sep.hip)
        const int rc = fg_hmc_sep_launch(e, iter0, n, welford_on, draws, first_sample_t, pos_all, info);
        if (rc != FG_E_UNSUPPORTED) return rc;
    }
    {   // dense regressions: observation-major finite difference (fg_hmc_lin.hip)
        const int rc = fg_hmc_lin_launch(e, iter0, n, welford_on, draws, first_sample_t, pos_all, info);
        if (rc != FG_E_UNSUPPORTED) return rc;
    }
    if (e->cfg.grad_mode == FG_GRAD_FD_SPARSE && e->P.gstream && e->jit_state >= 0) {
        // Gradient-stream programs: the program compiled at run time (fg_jit.cpp) is faster than the stream kernel wherever records are more
        // than fast Normals (linear predictors, general distributions, option selects: hier_scale 1.2e9 -> 3.7e9, linreg 2.8e9 -> 2.0e10
        // leapfrog-steps/s, bit-identical -- tools/bench_jit_vs_stream.py), and for fast-Normal programs when the tiles do not fill the GPU
        // (reference_model(8) at 8 192 chains: 3.6e9 -> 6.2e9; at 65 536 chains the stream kernel keeps 13 %).  FG_JIT=2 forces it.
        if (hmc_jit_preferred(e)) {
            const int rc = fg_hmc_jit_launch(e, iter0, n, welford_on, draws, first_sample_t, pos_all, info);
            if (rc != FG_E_UNSUPPORTED) return rc;
        }
    }
    if (e->cfg.grad_mode == FG_GRAD_FD_DENSE && e->P.gstream && e->jit_state >= 0) {
        // The dense mode of gradient-stream programs: the whole program per (coordinate, sign) as generated code (fg_jit_full_k) beats the dense
        // stream at every size measured (reference_model(32) 3.6e8 -> 7.6e8 leapfrog-steps/s, reference_model(8) 4.6e9 -> 8.9e9, hier 2.6e9 ->
        // 5.5e9; at 8 192 chains 1.7e8 -> 3.7e8, 1.1e9 -> 3.6e9, 4.4e8 -> 2.5e9: profiles/round4_jit_dense.txt), bit-identical; FG_JIT=0 (or
        // a program whose d copies are too much to compile) keeps the dense stream / the interpreter kernels.
        const int rc = fg_hmc_jit_launch(e, iter0, n, welford_on, draws, first_sample_t, pos_all, info);
        if (rc != FG_E_UNSUPPORTED) return rc;
    }
    if ((((e->cfg.grad_mode == FG_GRAD_FD_SPARSE || analytic) && e->P.gstream) || dense_stream) && e->tw == FG_WAVE && !e->gt) {
        // waves per tile: aim at 4 waves per SIMD (16 per CU, see k_hmc_stream_steps).  The LDS tile caps the tiles
        // resident on a CU (160 KB / lds_bytes -- 4 for the 32-site model), so the waves have to come from sharing
        // a tile, whatever the chain count; each wave should still own at least 2 coordinates
        int W = e->mw_override > 0 ? std::min(e->mw_override, FG_MW_MAX) : 1;
        if (e->mw_override <= 0) {
            const long long n_cu = std::max(1, e->n_simd / 4);
            const long long resident = std::max(1LL, std::min<long long>((160 * 1024) / (long long)e->lds_bytes, ((long long)tiles + n_cu - 1) / n_cu));
            while (W < FG_MW_MAX && resident * W < 16 && e->d >= 4 * W) W *= 2;
        }
        FgSeg seg;
        const std::vector<FgGradRec> &gs = e->prog->gstream;
        const int nrec = e->prog->n_gstream;
        std::vector<int> cstart(e->d + 1, nrec);                 // first record of each coordinate
        for (int k = nrec - 1; k >= 0; --k) cstart[gs[k].coord] = k;
        std::vector<long long> cum(nrec + 1, 0);                 // work before record k: a linear predictor costs its terms
        for (int k = 0; k < nrec; ++k) cum[k + 1] = cum[k] + ((gs[k].flags & FG_G_LIN) ? 1 + gs[k].maskm / 2 : 1);
        for (int w = 0; w <= FG_MW_MAX; ++w) { seg.c[w] = e->d; seg.g[w] = nrec; }
        seg.c[0] = 0; seg.g[0] = 0;
        for (int w = 1, k = 0; w < W; ++w) {
            if (dense_stream) { seg.c[w] = (int)((long long)e->d * w / W); seg.g[w] = 0; continue; }   // every coordinate costs one whole-program pass
            const long long target = cum[nrec] * w / W;           // cut at the coordinate boundary nearest to w/W of the work
            while (k < e->d && cum[cstart[k]] < target) ++k;
            seg.c[w] = k; seg.g[w] = cstart[k];
        }
        // do the waves interact inside a trajectory?  Not when every record only reads coordinates of its own wave.
        seg.separable = 1;
        for (int w = 0; w < W && !dense_stream; ++w)
            for (int k = seg.g[w]; k < seg.g[w + 1]; ++k) {
                const FgGradRec &r = gs[k];
                const bool x_ok = (r.flags & FG_G_X_CONST) || ((int)r.xi >= seg.c[w] && (int)r.xi < seg.c[w + 1]);
                const bool m_ok = (r.flags & FG_G_M_CONST) || ((int)r.mi >= seg.c[w] && (int)r.mi < seg.c[w + 1]);
                if (!x_ok || !m_ok || (r.flags & FG_G_LIN)) seg.separable = 0;   // a linear predictor reads many coordinates
            }
        int rk = e->P.sstream_kinds;                             // record kinds present in either stream
        for (int k = 0; k < nrec && rk < 2; ++k) rk = std::max(rk, (gs[k].flags & FG_G_GEN) ? 2 : ((gs[k].flags & FG_G_LIN) ? 1 : 0));
#define FG_LAUNCH_STREAM(RK, AN, SS) hipLaunchKernelGGL((k_hmc_stream_steps<RK, AN, SS>), dim3(tiles), dim3(FG_WAVE * W), e->lds_bytes, e->stream, e->P, e->X, e->H, seg, \
                                                        iter0, n, e->n_warmup, welford_on, draws, first_sample_t, pos_all, info)
        const bool ss = e->P.sstream != nullptr;
        if (!analytic) {
            if (ss) { if (rk == 2) FG_LAUNCH_STREAM(2, false, true); else if (rk == 1) FG_LAUNCH_STREAM(1, false, true); else FG_LAUNCH_STREAM(0, false, true); }
            else { if (rk == 2) FG_LAUNCH_STREAM(2, false, false); else if (rk == 1) FG_LAUNCH_STREAM(1, false, false); else FG_LAUNCH_STREAM(0, false, false); }
        } else {
            if (ss) { if (rk == 1) FG_LAUNCH_STREAM(1, true, true); else FG_LAUNCH_STREAM(0, true, true); }
            else { if (rk == 1) FG_LAUNCH_STREAM(1, true, false); else FG_LAUNCH_STREAM(0, true, false); }
        }
#undef FG_LAUNCH_STREAM
        HIPCHK(hipGetLastError());
        e->last_hmc_kernel = std::string(dense_stream ? "k_hmc_stream_steps (dense stream) W=" : "k_hmc_stream_steps W=") + std::to_string(W);
        return FG_OK;
    }
    {   // programs without a record stream, compiled at run time (fg_jit.cpp)
        const int rc = fg_hmc_jit_launch(e, iter0, n, welford_on, draws, first_sample_t, pos_all, info);
        if (rc != FG_E_UNSUPPORTED) return rc;
    }
    {   // interpreter programs: the tile shared by W waves, each on its own copy of the slots (fg_hmc_interp.hip)
        const int rc = fg_hmc_interp_launch(e, iter0, n, welford_on, draws, first_sample_t, pos_all, info);
        if (rc != FG_E_UNSUPPORTED) return rc;
    }
    e->last_hmc_kernel = "k_hmc_steps W=1";
    FG_LAUNCH_GT(e, k_hmc_steps, dim3(tiles), dim3(e->tw), e->lds_bytes, e->stream, e->P, e->X, e->H, iter0, n,
                       e->n_warmup, welford_on, draws, first_sample_t, pos_all, info);
    HIPCHK(hipGetLastError());
    return FG_OK;
}

int fg_internal_hmc_step(fg_engine *e, int n_transitions, double *d_draws, double *d_pos_all, double *d_info) {
    NEED_ENGINE(e);
    if (!e->hmc_ready) { fg_set_error("fg_hmc_step before fg_hmc_init"); return FG_E_STATE; }
    if (n_transitions < 0) return FG_E_BAD_ARG;
    int done = 0;
    long long rows_written = 0;          // post-warmup rows appended to d_draws by this call
    while (done < n_transitions) {
        const int iter = e->iter;
        if (e->d == 0) {                                   // fresh prior draw per step: hmc.rs:826-845
            if (d_pos_all || d_info) { fg_set_error("fg_hmc_step_info: model has no continuous sites"); return FG_E_UNSUPPORTED; }
            int rc = fg_launch_prior(e, (uint32_t)(iter + 1), FG_RNG_PRIOR, nullptr, e->H.lj);
            if (rc) return rc;
            e->iter += 1; done += 1;
            continue;
        }
        int n = n_transitions - done;
        // stop a launch at the mass-adaptation iteration (its reset runs between launches)
        if (e->mass_adapt_at >= 0 && iter < e->mass_adapt_at) n = std::min(n, e->mass_adapt_at - iter);
        const int welford_on = (e->mass_adapt_at >= 0) ? 1 : 0;
        const int first_sample_t = std::max(iter, e->n_warmup) - iter;      // first post-warmup t of this launch
        const int n_rows = std::max(0, n - first_sample_t);
        double *draws = (d_draws && n_rows > 0) ? d_draws + rows_written * (long long)e->d * e->C : nullptr;
        int rc = hmc_launch_steps(e, iter, n, welford_on, draws, first_sample_t,
                                  d_pos_all ? d_pos_all + (long long)done * e->d * e->C : nullptr,
                                  d_info ? d_info + (long long)done * 4 * e->C : nullptr);
        if (rc) return rc;
        rows_written += n_rows;
        e->iter += n; done += n;
        if (e->mass_adapt_at >= 0 && e->iter == e->mass_adapt_at) {    // hmc.rs:885-908
            const int TB = 256, NB = (int)((e->C + TB - 1) / TB);
            hipLaunchKernelGGL(k_hmc_mass_reset, dim3(NB), dim3(TB), 0, e->stream, e->H, e->C, e->d);
            rc = hmc_find_eps(e, 1, 0, e->d_tmp);
            if (rc) return rc;
            hipLaunchKernelGGL(k_hmc_da_new, dim3(NB), dim3(TB), 0, e->stream, e->H, e->C, (const double *)e->d_tmp);
            HIPCHK(hipGetLastError());
        }
    }
    return FG_OK;
}

int fg_hmc_step(fg_engine *e, int n_transitions, double *d_draws) { return fg_internal_hmc_step(e, n_transitions, d_draws, nullptr, nullptr); }
int fg_hmc_step_info(fg_engine *e, int n_transitions, double *d_positions, double *d_info) {
    return fg_internal_hmc_step(e, n_transitions, nullptr, d_positions, d_info);
}

int fg_hmc_get_stats(fg_engine *e, fg_hmc_stats *st) {
    NEED_ENGINE(e);
    if (!st || !e->hmc_ready) return FG_E_BAD_ARG;
    std::vector<double> a((size_t)e->C), eps((size_t)e->C), fr((size_t)e->C), leb((size_t)e->C);
    std::vector<unsigned long long> nd((size_t)e->C);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(a.data(), e->H.alpha_sum, a.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nd.data(), e->H.n_div, nd.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(eps.data(), e->H.eps, eps.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(fr.data(), e->H.frozen, fr.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(leb.data(), e->H.da_leb, leb.size() * 8, hipMemcpyDeviceToHost));
    double as = 0.0, es = 0.0; long long dv = 0;
    for (size_t i = 0; i < a.size(); i++) {
        as += a[i]; dv += (long long)nd[i];
        double cur = eps[i];
        if (e->iter >= e->n_warmup) cur = !std::isnan(fr[i]) ? fr[i] : (e->n_warmup > 0 ? std::exp(leb[i]) : eps[i]);
        es += cur;
    }
    st->n_transitions = (long long)e->iter * e->C;
    st->accept_rate = (e->d == 0) ? 1.0 : (st->n_transitions > 0 ? as / (double)st->n_transitions : 0.0);
    st->mean_step_size = es / (double)e->C;
    st->n_divergent = dv;
    return FG_OK;
}

int fg_hmc_run(fg_engine *e, const fg_hmc_config *cfg, int n_samples, int n_warmup, double *d_draws, fg_hmc_stats *st) {
    int rc = fg_hmc_init(e, cfg, n_warmup);
    if (rc) return rc;
    rc = fg_hmc_step(e, n_warmup, nullptr);
    if (rc) return rc;
    rc = fg_hmc_step(e, n_samples, d_draws);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (st) return fg_hmc_get_stats(e, st);
    return FG_OK;
}

int fg_hmc_get_step_sizes(fg_engine *e, double *h_eps) {
    NEED_ENGINE(e);
    if (!h_eps || !e->hmc_ready) return FG_E_BAD_ARG;
    std::vector<double> eps((size_t)e->C), fr((size_t)e->C), leb((size_t)e->C);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(eps.data(), e->H.eps, eps.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(fr.data(), e->H.frozen, fr.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(leb.data(), e->H.da_leb, leb.size() * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < eps.size(); i++)     // HmcSession::step_size, hmc.rs:771-777
        h_eps[i] = (e->iter < e->n_warmup) ? eps[i] : (!std::isnan(fr[i]) ? fr[i] : (e->n_warmup > 0 ? std::exp(leb[i]) : eps[i]));
    return FG_OK;
}
int fg_hmc_get_log_joint(fg_engine *e, double *h_lj) {
    NEED_ENGINE(e);
    if (!h_lj || !e->hmc_ready) return FG_E_BAD_ARG;
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(h_lj, e->H.lj, (size_t)e->C * 8, hipMemcpyDeviceToHost));
    return FG_OK;
}
int fg_hmc_get_mass(fg_engine *e, double *h_m_inv) {
    NEED_ENGINE(e);
    if (!h_m_inv || !e->hmc_ready) return FG_E_BAD_ARG;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->H.use_mass) HIPCHK(hipMemcpy(h_m_inv, e->H.m_inv, (size_t)e->d * e->C * 8, hipMemcpyDeviceToHost));
    else for (size_t i = 0; i < (size_t)e->d * e->C; i++) h_m_inv[i] = 1.0;
    return FG_OK;
}
int fg_hmc_set_step_size(fg_engine *e, double eps) {    // hmc.rs:741-747
    NEED_ENGINE(e);
    if (!e->hmc_ready) return FG_E_STATE;
    eps = std::max(eps, 1e-12);
    const int TB = 256, NB = (int)((e->C + TB - 1) / TB);
    hipLaunchKernelGGL(k_fill, dim3(NB), dim3(TB), 0, e->stream, e->H.eps, e->C, eps);
    hipLaunchKernelGGL(k_fill, dim3(NB), dim3(TB), 0, e->stream, e->H.frozen, e->C, eps);
    HIPCHK(hipGetLastError());
    e->n_warmup = std::min(e->n_warmup, e->iter);
    // leaving warmup also ends mass adaptation: the reference only adapts the mass while `warming` (hmc.rs:877-908)
    if (e->mass_adapt_at >= e->iter) e->mass_adapt_at = -1;
    return FG_OK;
}
int fg_hmc_set_n_leapfrog(fg_engine *e, int n_leapfrog) {     // hmc.rs:751-753
    NEED_ENGINE(e);
    if (!e->hmc_ready) return FG_E_STATE;
    e->cfg.n_leapfrog = n_leapfrog > 1 ? n_leapfrog : 1;
    e->H.L = e->cfg.n_leapfrog;
    return FG_OK;
}
const char *fg_hmc_last_kernel(const fg_engine *e) { return e ? e->last_hmc_kernel.c_str() : ""; }
const char *fg_mh_last_kernel(const fg_engine *e) { return e ? e->last_mh_kernel.c_str() : ""; }
int fg_hmc_is_warming_up(const fg_engine *e) { return (e && e->hmc_ready && e->iter < e->n_warmup) ? 1 : 0; }   // hmc.rs:780-782
int64_t fg_hmc_iterations(const fg_engine *e) { return (e && e->hmc_ready) ? (int64_t)e->iter : 0; }             // hmc.rs:785-787

int fg_hmc_grad(fg_engine *e, double h, int grad_mode, double *h_grad, int32_t *h_ok) {
    NEED_ENGINE(e);
    if (!h_grad) return FG_E_BAD_ARG;
    if (int rc = hmc_lds_ok(e)) return rc;
    if (grad_mode == FG_GRAD_ANALYTIC && (!e->P.gstream || fg_program_stream_records(e->prog, 2) > 1)) {
        fg_set_error("FG_GRAD_ANALYTIC is not available for this program"); return FG_E_UNSUPPORTED; }
    double *d_g = nullptr;
    int rc = dev_alloc(&d_g, (size_t)std::max(1, e->d) * e->C);
    if (rc) return rc;
    FG_LAUNCH_GT(e, k_hmc_grad, dim3((unsigned)((e->C + e->tw - 1) / e->tw)), dim3(e->tw), e->lds_bytes, e->stream, e->P, e->X, h,
                       grad_mode == FG_GRAD_ANALYTIC ? 2 : (grad_mode == FG_GRAD_FD_SPARSE ? 1 : 0), d_g, e->d_itmp);
    hipError_t le = hipGetLastError();
    if (le == hipSuccess) le = hipMemcpyAsync(h_grad, d_g, (size_t)e->d * e->C * 8, hipMemcpyDeviceToHost, e->stream);
    if (le == hipSuccess && h_ok) le = hipMemcpyAsync(h_ok, e->d_itmp, (size_t)e->C * 4, hipMemcpyDeviceToHost, e->stream);
    if (le == hipSuccess) le = hipStreamSynchronize(e->stream);
    hipFree(d_g);
    if (le != hipSuccess) { fg_set_error(hipGetErrorString(le)); return FG_E_HIP; }
    return FG_OK;
}

static int hmc_prepare_injected(fg_engine *e, const fg_hmc_config *cfg) {
    if (!cfg) return FG_E_BAD_ARG;
    if (int rc0 = hmc_lds_ok(e)) return rc0;
    int rc = fg_internal_hmc_alloc(e, false);
    if (rc) return rc;
    if (!e->hmc_ready) {      // standalone use: lj of the current values, identity mass
        e->H.use_mass = 0;
        FG_LAUNCH_GT(e, k_log_joint, dim3((unsigned)((e->C + e->tw - 1) / e->tw)), dim3(e->tw), e->lds_score, e->stream, e->P, e->X, (double *)nullptr,
                           (double *)nullptr, e->H.lj);
        HIPCHK(hipGetLastError());
    }
    fg_internal_hmc_set_cfg(e, cfg);
    return FG_OK;
}

int fg_hmc_transition_injected(fg_engine *e, const fg_hmc_config *cfg, double eps, const double *h_p0, const double *h_u,
                               int32_t *h_acc, double *h_alpha, int32_t *h_div, double *h_lj) {
    NEED_ENGINE(e);
    if (!h_p0 || !h_u) return FG_E_BAD_ARG;
    int rc = hmc_prepare_injected(e, cfg);
    if (rc) return rc;
    size_t C = (size_t)e->C;
    HIPCHK(hipMemcpyAsync(e->H.p0_scratch, h_p0, (size_t)e->d * C * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_tmp, h_u, C * 8, hipMemcpyHostToDevice, e->stream));
    int *ia = e->d_itmp, *idv = e->d_itmp + C;
    double *al = e->d_acc;
    FG_LAUNCH_GT(e, k_hmc_transition_injected, dim3((unsigned)((e->C + e->tw - 1) / e->tw)), dim3(e->tw), e->lds_bytes, e->stream, e->P, e->X, e->H,
                       eps, (const double *)e->H.p0_scratch, (const double *)e->d_tmp, ia, al, idv);
    HIPCHK(hipGetLastError());
    if (h_acc) HIPCHK(hipMemcpyAsync(h_acc, ia, C * 4, hipMemcpyDeviceToHost, e->stream));
    if (h_div) HIPCHK(hipMemcpyAsync(h_div, idv, C * 4, hipMemcpyDeviceToHost, e->stream));
    if (h_alpha) HIPCHK(hipMemcpyAsync(h_alpha, al, C * 8, hipMemcpyDeviceToHost, e->stream));
    if (h_lj) HIPCHK(hipMemcpyAsync(h_lj, e->H.lj, C * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return FG_OK;
}

int fg_hmc_find_eps_injected(fg_engine *e, const fg_hmc_config *cfg, const double *h_p0, double *h_eps) {
    NEED_ENGINE(e);
    if (!h_p0 || !h_eps) return FG_E_BAD_ARG;
    int rc = hmc_prepare_injected(e, cfg);
    if (rc) return rc;
    size_t C = (size_t)e->C;
    HIPCHK(hipMemcpyAsync(e->H.p0_scratch, h_p0, (size_t)e->d * C * 8, hipMemcpyHostToDevice, e->stream));
    rc = hmc_find_eps(e, 0, 1, e->d_tmp);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h_eps, e->d_tmp, C * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return FG_OK;
}


__global__ void k_mh_adapt_init(FgMhAdapt *ad, long long n) {        // DiminishingAdaptation::new: scale 1.0, ln scale 0.0, no counts (mcmc_utils.rs:51-62)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { FgMhAdapt a; a.scale = 1.0; a.kind = 0; a.pad = 0; a.log_scale = 0.0; a.tot = 0u; a.acc = 0u; ad[i] = a; }
}

// ------------------------------------------------------------------ MH host side
int fg_internal_mh_alloc(fg_engine *e) {
    size_t C = (size_t)e->C, S = (size_t)std::max(1, e->S);
    auto A = [&](auto **p, size_t n) { int rc = dev_alloc(p, n); if (!rc) e->mh_allocs.push_back((void *)*p); return rc; };
    if (!e->M.lw) {
        if (A(&e->M.lw, C) || A(&e->M.ad, S * C) || A(&e->M.n_acc, C))
            return FG_E_HIP;
    }
    return FG_OK;
}
int fg_internal_mh_set_overrides(fg_engine *e, const fg_site_proposal *overrides) {
    const size_t S = (size_t)std::max(1, e->S);
    e->M.ov_kind = nullptr; e->M.ov_lo = nullptr; e->M.ov_hi = nullptr;
    e->mh_has_prior_resample = false;
    e->mh_overrides.clear();
    if (!overrides) return FG_OK;
    e->mh_overrides.assign(overrides, overrides + e->S);
    for (int j = 0; j < e->S; j++) if (overrides[j].kind == FG_PROP_PRIOR_RESAMPLE) e->mh_has_prior_resample = true;
    std::vector<int> k(S); std::vector<double> lo(S), hi(S);
    for (int j = 0; j < e->S; j++) {            // device tables are indexed by LDS slot
        const int sl = e->prog->site_slot[j];
        k[sl] = overrides[j].kind; lo[sl] = overrides[j].lower; hi[sl] = overrides[j].upper;
        if (k[sl] < 0 || k[sl] > 4) { fg_set_error("fg_mh_init: unknown proposal kind"); return FG_E_BAD_ARG; }
    }
    int *dk = nullptr; double *dlo = nullptr, *dhi = nullptr;
    if (dev_upload(&dk, k) || dev_upload(&dlo, lo) || dev_upload(&dhi, hi)) return FG_E_HIP;
    e->mh_allocs.push_back(dk); e->mh_allocs.push_back(dlo); e->mh_allocs.push_back(dhi);
    e->M.ov_kind = dk; e->M.ov_lo = dlo; e->M.ov_hi = dhi;
    return FG_OK;
}
int fg_mh_init(fg_engine *e, int n_warmup, const fg_site_proposal *overrides) {
    NEED_ENGINE(e);
    if (n_warmup < 0) return FG_E_BAD_ARG;
    size_t C = (size_t)e->C, S = (size_t)std::max(1, e->S);
    if (int rc0 = fg_internal_mh_alloc(e)) return rc0;
    HIPCHK(hipMemsetAsync(e->M.n_acc, 0, C * 8, e->stream));
    const int TB = 256;
    hipLaunchKernelGGL(k_mh_adapt_init, dim3((unsigned)((S * C + TB - 1) / TB)), dim3(TB), 0, e->stream, e->M.ad, (long long)(S * C));   // scale 1, log_scale 0, no counts, kind undecided
    if (int rc1 = fg_internal_mh_set_overrides(e, overrides)) return rc1;
    int rc = fg_launch_prior(e, 0, FG_RNG_PRIOR, nullptr, e->M.lw);       // mh.rs:950-957
    if (rc) return rc;
    e->mh_warmup = n_warmup; e->mh_iter = 0; e->mh_ready = true;
    return FG_OK;
}

int fg_mh_set_recording(fg_engine *e, int during_adaptation) {
    NEED_ENGINE(e);
    e->M.rec_all = during_adaptation ? 1 : 0;
    return FG_OK;
}

// 1 / n^0.7 for every proposal count an adapting session can reach (a site is proposed at most once per step), capped at 2^20
// entries; built by the host's pow when a session first steps inside its warmup (fg_mh_init and fg_state_import both end here)
static int fg_mh_step_table(fg_engine *e) {
    const uint32_t want = (uint32_t)std::min<long long>((long long)e->mh_warmup + 1, 1LL << 20);
    if (e->M.step_n >= want) return FG_OK;
    std::vector<double> tab((size_t)want);
    for (uint32_t n = 0; n < want; ++n) tab[n] = 1.0 / std::pow((double)n, 0.7);
    double *d = nullptr;
    if (dev_upload(&d, tab)) return FG_E_HIP;
    e->mh_allocs.push_back(d);
    e->M.step_tab = d; e->M.step_n = want;
    return FG_OK;
}

int fg_mh_step(fg_engine *e, int n_steps, const int32_t *h_rec_sites, int n_rec, void *d_draws) {
    NEED_ENGINE(e);
    if (!e->mh_ready) { fg_set_error("fg_mh_step before fg_mh_init"); return FG_E_STATE; }
    if (n_steps < 0 || n_rec < 0 || (n_rec > 0 && !h_rec_sites)) return FG_E_BAD_ARG;
    if (n_steps > 0 && e->mh_iter < e->mh_warmup) { if (int rc = fg_mh_step_table(e)) return rc; }
    if (e->S == 0) { e->mh_iter += n_steps; return FG_OK; }             // no latent sites: nothing to move (mh.rs:713-715)
    if (n_rec > 0) {
        for (int r = 0; r < n_rec; r++)
            if (h_rec_sites[r] < 0 || h_rec_sites[r] >= e->S) { fg_set_error("fg_mh_step: recorded site out of range"); return FG_ERR_ADDRESS_NOT_FOUND; }
        if (n_rec > e->rec_cap) {
            if (e->d_rec) { HIPCHK(hipStreamSynchronize(e->stream)); HIPCHK(hipFree(e->d_rec)); }
            HIPCHK(hipMalloc((void **)&e->d_rec, (size_t)n_rec * 4)); e->rec_cap = n_rec;
        }
        std::vector<int> rec_slots((size_t)n_rec);
        for (int r = 0; r < n_rec; r++) rec_slots[r] = e->prog->site_slot[h_rec_sites[r]];
        HIPCHK(hipMemcpy(e->d_rec, rec_slots.data(), (size_t)n_rec * 4, hipMemcpyHostToDevice));
    }
    e->M.rec = e->d_rec; e->M.n_rec = n_rec;
    const int iter = e->mh_iter;
    const int first_sample_t = e->M.rec_all ? 0 : std::max(iter, e->mh_warmup) - iter;
    if (n_steps > 0 && std::getenv("FG_JIT") && std::atoi(std::getenv("FG_JIT")) == 2) {
        const int rc2 = fg_mh_interp_launch(e, iter, n_steps, (n_rec > 0) ? (long long *)d_draws : (long long *)nullptr, first_sample_t);
        if (rc2 == FG_OK) { e->mh_iter += n_steps; return FG_OK; }
        if (rc2 != FG_E_UNSUPPORTED) return rc2;
    }
    if (n_steps > 0) {                                                  // multi-wave tiles when every statement has a score-stream record
        const int rc = fg_mh_mw_launch(e, iter, n_steps, (n_rec > 0) ? (long long *)d_draws : (long long *)nullptr, first_sample_t);
        if (rc == FG_OK) { e->mh_iter += n_steps; return FG_OK; }
        if (rc != FG_E_UNSUPPORTED) return rc;
        const int rc2 = fg_mh_interp_launch(e, iter, n_steps, (n_rec > 0) ? (long long *)d_draws : (long long *)nullptr, first_sample_t);   // interpreter programs: statements split over waves
        if (rc2 == FG_OK) { e->mh_iter += n_steps; return FG_OK; }
        if (rc2 != FG_E_UNSUPPORTED) return rc2;
    }
    e->last_mh_kernel = "k_mh_steps W=1";
    FG_LAUNCH_GT(e, k_mh_steps, dim3((unsigned)((e->C + e->tw - 1) / e->tw)), dim3(e->tw), e->lds_score, e->stream, e->P, e->X, e->M,
                       iter, n_steps, e->mh_warmup, (n_rec > 0) ? (long long *)d_draws : (long long *)nullptr, first_sample_t);
    HIPCHK(hipGetLastError());
    e->mh_iter += n_steps;
    return FG_OK;
}

int fg_mh_get_stats(fg_engine *e, fg_mh_stats *st) {
    NEED_ENGINE(e);
    if (!st || !e->mh_ready) return FG_E_BAD_ARG;
    std::vector<unsigned long long> a((size_t)e->C);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(a.data(), e->M.n_acc, a.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long tot = 0;
    for (auto v : a) tot += v;
    st->n_steps = (long long)e->mh_iter * e->C;
    st->accept_rate = st->n_steps > 0 ? (double)tot / (double)st->n_steps : 0.0;
    return FG_OK;
}

int fg_mh_run(fg_engine *e, int n_samples, int n_warmup, const fg_site_proposal *overrides, const int32_t *h_rec_sites, int n_rec,
              void *d_draws, fg_mh_stats *st) {
    int rc = fg_mh_init(e, n_warmup, overrides);
    if (rc) return rc;
    rc = fg_mh_step(e, n_warmup, nullptr, 0, nullptr);
    if (rc) return rc;
    rc = fg_mh_step(e, n_samples, h_rec_sites, n_rec, d_draws);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (st) return fg_mh_get_stats(e, st);
    return FG_OK;
}

int fg_mh_get_scales(fg_engine *e, double *h_scales) {
    NEED_ENGINE(e);
    if (!h_scales || !e->mh_ready) return FG_E_BAD_ARG;
    HIPCHK(hipStreamSynchronize(e->stream));
    std::vector<FgMhAdapt> ad((size_t)e->S * e->C);
    HIPCHK(hipMemcpy(ad.data(), e->M.ad, ad.size() * sizeof(FgMhAdapt), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < ad.size(); ++k) h_scales[k] = ad[k].scale;
    return FG_OK;
}
int fg_mh_get_log_weight(fg_engine *e, double *h_lw) {
    NEED_ENGINE(e);
    if (!h_lw || !e->mh_ready) return FG_E_BAD_ARG;
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(h_lw, e->M.lw, (size_t)e->C * 8, hipMemcpyDeviceToHost));
    return FG_OK;
}

}  // extern "C"
