// fg_smc.hip -- likelihood-tempered SMC (adaptive_smc, src/inference/smc.rs:455-581) on gfx950.
//
// N particles = the engine's N "chains": values [S][N] in HBM, one lane per particle for model
// runs.  The population-wide steps are streaming kernels over [N] arrays:
//   * log-sum-exp / ESS(b)  (numerical.rs:15-38, smc.rs:588-622): block max -> block sums ->
//     one-thread finish; partials are combined in a fixed order, so results are reproducible.
//     The 64-step ESS bisection of next_beta runs entirely on the device (its state lives in
//     HBM; no host round-trip per iteration).
//   * systematic / stratified / multinomial resampling (smc.rs:255-314): chunked inclusive
//     prefix sum of the weights (LDS block scan + sequential scan of the chunk totals) and a
//     binary search per output slot for the first index whose cumulative weight reaches the
//     threshold -- the index the reference's sequential walk stops at.
//   * gather of the resampled particles into a second [S][N] buffer (buffers swap).
//   * rejuvenation: one tempered single-site MH move per particle per sweep
//     (smc.rs:631-688: two model runs, accept on d(log_prior) + beta * d(loglik)).
// Deviation (documented in DESIGN.md): the reference threads ONE DiminishingAdaptation through
// all particles sequentially (smc.rs:482,544-553); here every particle of a sweep uses the
// scales from the start of the sweep and the per-site counts are folded in once per sweep.
#include "fg_engine_internal.h"
#include "fg_gradstream.h"

#define RED_BLOCKS 512
#define RED_THREADS 256
#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_CHUNK (SCAN_THREADS * SCAN_ITEMS)

struct FgSmcScalars {      // device-resident scalars of one SMC run
    double beta, bnew, lo, hi, mid, one, target_ess;
    double log_evidence, log_norm, lse1, lse2, ess;
    int done, force_one;
};

// ---------------------------------------------------------------------------------------
// reductions:  v_i = lw_i + (b - beta) * ll_i     (smc.rs:590-594 / :512-516)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double smc_v(const double *lw, const double *ll, long long i, double b, double beta) {
    return lw[i] + (b - beta) * ll[i];
}
__device__ __forceinline__ double block_reduce_max(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) { double m = sh[0]; for (int k = 1; k < (int)(blockDim.x >> 6); ++k) m = fmax(m, sh[k]); sh[0] = m; }
    __syncthreads();
    const double r = sh[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_reduce_sum(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) { double m = sh[0]; for (int k = 1; k < (int)(blockDim.x >> 6); ++k) m += sh[k]; sh[0] = m; }
    __syncthreads();
    const double r = sh[0];
    __syncthreads();
    return r;
}
__global__ __launch_bounds__(RED_THREADS) void k_smc_red_max(const double *lw, const double *ll, long long n, const double *b_ptr,
                                                              const double *beta_ptr, double *part_max) {
    __shared__ double sh[RED_THREADS / 64];
    const double b = *b_ptr, beta = *beta_ptr;
    double m = -INFINITY;                                   // fold(NEG_INFINITY, max)  numerical.rs:21-23
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        m = fmax(m, ll ? smc_v(lw, ll, i, b, beta) : lw[i]);
    m = block_reduce_max(m, sh);
    if (threadIdx.x == 0) part_max[blockIdx.x] = m;
}
__global__ __launch_bounds__(RED_THREADS) void k_smc_red_sum(const double *lw, const double *ll, long long n, const double *b_ptr,
                                                              const double *beta_ptr, const double *part_max, double *part_sum) {
    __shared__ double sh[RED_THREADS / 64];
    const double b = *b_ptr, beta = *beta_ptr;
    double m = -INFINITY;                                   // max of the block maxima (max is exact: any order)
    for (int k = threadIdx.x; k < (int)gridDim.x; k += blockDim.x) m = fmax(m, part_max[k]);
    m = block_reduce_max(m, sh);
    double s1 = 0.0, s2 = 0.0;
    if (!(isinf(m) && m < 0.0)) {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            const double t = (ll ? smc_v(lw, ll, i, b, beta) : lw[i]) - m;
            s1 += exp(t);                                    // sum(exp(x - max))  numerical.rs:31
            s2 += exp(2.0 * t);                              // the same for 2v (max(2v) = 2 max(v))  smc.rs:596-597
        }
    }
    s1 = block_reduce_sum(s1, sh);
    s2 = block_reduce_sum(s2, sh);
    if (threadIdx.x == 0) { part_sum[2 * blockIdx.x] = s1; part_sum[2 * blockIdx.x + 1] = s2; }
}
// phase 0: ESS at b = 1 (smc.rs:604-607);  phase 1: one bisection step (:612-619);
// phase 2: finish next_beta (:620-621);  phase 3: log_norm of the reweight (:517-518);
// phase 4: plain log-sum-exp (lse1 only)
// One block of RED_THREADS threads: the block partials are combined by a fixed tree (thread t takes partials t, t + 256,
// ...; wave shuffles; waves in order), so the result is reproducible from run to run.
__global__ __launch_bounds__(RED_THREADS) void k_smc_finish(FgSmcScalars *st, const double *part_max, const double *part_sum, int nb, long long n, int phase) {
    __shared__ double sh[RED_THREADS / 64];
    if (blockIdx.x != 0) return;
    if (phase == 2) {
        if (threadIdx.x == 0) {
            if (!st->done) st->bnew = fmin(fmax(st->hi, st->beta + 1e-9), 1.0);
            if (st->force_one) st->bnew = 1.0;
        }
        return;
    }
    double m = -INFINITY, s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) { m = fmax(m, part_max[k]); s1 += part_sum[2 * k]; s2 += part_sum[2 * k + 1]; }
    m = block_reduce_max(m, sh);
    s1 = block_reduce_sum(s1, sh);
    s2 = block_reduce_sum(s2, sh);
    if (threadIdx.x != 0) return;
    const bool empty = isinf(m) && m < 0.0;
    const double lse1 = (empty || s1 == 0.0) ? -INFINITY : m + log(s1);          // numerical.rs:33-37
    const double lse2 = (empty || s2 == 0.0) ? -INFINITY : 2.0 * m + log(s2);
    st->lse1 = lse1; st->lse2 = lse2;
    if (phase == 3) { st->log_norm = lse1; st->log_evidence += lse1; return; }
    if (phase == 4) return;
    const double ess = (!isfinite(lse1) || !isfinite(lse2)) ? (double)n : exp(2.0 * lse1 - lse2);   // smc.rs:598-601
    st->ess = ess;
    if (phase == 0) {
        st->done = ess >= st->target_ess;
        if (st->done) st->bnew = 1.0;
        st->lo = st->beta; st->hi = 1.0; st->mid = 0.5 * (st->lo + st->hi);
    } else if (!st->done) {
        if (ess < st->target_ess) st->hi = st->mid; else st->lo = st->mid;
        st->mid = 0.5 * (st->lo + st->hi);
    }
}
// lw <- combined - log_norm (or uniform), w <- exp(lw)      smc.rs:520-528,535
__global__ void k_smc_apply(double *lw, const double *ll, double *w, long long n, const FgSmcScalars *st) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = lw[i] + (st->bnew - st->beta) * ll[i];
    const double nl = isfinite(st->log_norm) ? v - st->log_norm : -log((double)n);
    lw[i] = nl;
    if (w) w[i] = exp(nl);
}
__global__ void k_smc_set_beta(FgSmcScalars *st) { if (threadIdx.x == 0 && blockIdx.x == 0) st->beta = st->bnew; }
// final normalisation (smc.rs:565-575): lw <- lw - lse(lw), w <- exp(lw); uniform if lse is not finite
__global__ void k_smc_normalize(double *lw, double *w, long long n, const FgSmcScalars *st) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (isfinite(st->lse1)) { const double nz = lw[i] - st->lse1; lw[i] = nz; w[i] = exp(nz); }
    else { lw[i] = -log((double)n); w[i] = 1.0 / (double)n; }
}
__global__ void k_smc_split_acc(const double *acc, double *lprior, double *ll, long long n) {   // particle_log_likelihood  smc.rs:381-383
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    lprior[i] = acc[i];
    ll[i] = acc[n + i] + acc[2 * n + i];
}
__global__ void k_smc_is_weights(double *lw, const double *ll, long long n) {    // pure importance sampling  smc.rs:490
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lw[i] = -log((double)n) + ll[i];
}

// ---------------------------------------------------------------------------------------
// resampling
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_chunk_sums(const double *w, long long n, double *chunk_sum) {
    __shared__ double sh[SCAN_THREADS / 64];
    const long long base = (long long)blockIdx.x * SCAN_CHUNK + (long long)threadIdx.x * SCAN_ITEMS;
    double s = 0.0;
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) s += w[base + k];
    s = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = s;
}
__global__ void k_scan_chunk_offsets(double *chunk_sum, int n_chunks) {   // exclusive scan, sequential (n_chunks is small)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double run = 0.0;
    for (int k = 0; k < n_chunks; ++k) { const double t = chunk_sum[k]; chunk_sum[k] = run; run += t; }
}
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_cumsum(const double *w, long long n, const double *chunk_off, double *cum) {
    __shared__ double sh[SCAN_THREADS];
    const long long base = (long long)blockIdx.x * SCAN_CHUNK + (long long)threadIdx.x * SCAN_ITEMS;
    double loc[SCAN_ITEMS];
    double s = 0.0;
    for (int k = 0; k < SCAN_ITEMS; ++k) { s += (base + k < n) ? w[base + k] : 0.0; loc[k] = s; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < SCAN_THREADS; o <<= 1) {             // Hillis-Steele inclusive scan of the thread totals
        const double t = (threadIdx.x >= (unsigned)o) ? sh[threadIdx.x - o] : 0.0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    const double off = chunk_off[blockIdx.x] + (threadIdx.x ? sh[threadIdx.x - 1] : 0.0);
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) cum[base + k] = off + loc[k];
}
// idx_j = first k with cum[k] >= thr_j, else n-1: where `while cum < thr && i < n` stops (smc.rs:263-270)
__global__ void k_resample_search(const double *cum, long long n, int method, double U, const double *u_arr,
                                  unsigned long long seed, uint32_t step, long long *idx) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double thr;
    if (method == 1) thr = U / (double)n + (double)j / (double)n;                        // systematic  smc.rs:258,264
    else {
        double u;
        if (u_arr) u = u_arr[j];
        else { FgStream s = fg_stream(seed, (uint32_t)j, step, FG_RNG_SMC_RESAMPLE); u = fg_rng_u01(s); }
        thr = (method == 2) ? ((double)j + u) / (double)n : u;                             // stratified :284 / multinomial :300
    }
    long long lo = 0, hi = n;                                  // first k in [0,n) with cum[k] >= thr
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (cum[mid] >= thr) hi = mid; else lo = mid + 1; }
    idx[j] = lo < n ? lo : n - 1;
}
__global__ void k_smc_gather(const long long *src, long long *dst, const double *ll_src, double *ll_dst, const double *lp_src,
                             double *lp_dst, const long long *idx, int S, long long n) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long long a = idx[j];
    for (int s = 0; s < S; ++s) dst[(long long)s * n + j] = src[(long long)s * n + a];   // particles[i].clone()  smc.rs:537
    ll_dst[j] = ll_src[a];
    lp_dst[j] = lp_src[a];
}

// ---------------------------------------------------------------------------------------
// rejuvenation: tempered_single_site_mh (smc.rs:631-688), one move per particle
// ---------------------------------------------------------------------------------------
struct FgSmcDev {
    double *ll, *lprior;              // [N]
    double *scale, *log_scale;        // [S] shared DiminishingAdaptation (smc.rs:482)
    long long *acc, *tot;             // [S]
    unsigned int *sw_n, *sw_a;        // [S] per-sweep proposal / accept counts
};
__global__ __launch_bounds__(FG_WAVE, FG_MIN_WAVES) void k_smc_rejuv(FgProgramDev P, FgChainCtx X, FgSmcDev M, const FgSmcScalars *st,
                                                                      uint32_t move_id) {
    extern __shared__ double lds[];
    constexpr int tw = FG_WAVE;
    const long long chain = (long long)blockIdx.x * tw + threadIdx.x;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    double *slots = lds + threadIdx.x;
    fg_load_values(P, X, c, slots, tw);
    const double beta = st->beta;
    FgStream rng = fg_stream(X.seed, X.chain0 + (uint32_t)c, move_id, FG_RNG_SMC_REJUV);
    unsigned long long ra, rb;
    fg_rng_block(rng, ra, rb);
    const int k = (int)fg_pick(ra, (uint32_t)P.d);            // f64_sites[rng.gen_range(0..len)]  smc.rs:650
    const int site = P.f64_site[k];                           // sorted site index (adaptation / values row)
    const double scale = M.scale[site];                       // get_scale  smc.rs:651
    const double z = fg_rng_normal(rng);                      // Normal(0,1).sample  smc.rs:655
    const double cur = slots[k * tw];                         // LDS slot of coordinate k is k
    const double prop = cur + scale * z;
    double pri[2], lik[2];
    for (int pass = 0; pass < 2; ++pass) {                    // score current, then proposed: two model runs  smc.rs:662-675
        slots[k * tw] = pass ? prop : cur;
        FgAcc3 A = {0.0, 0.0, 0.0};
        if (P.sstream && P.sstream_kinds == 0) fg_score_stream<0>(P.sstream, P.n_sstream, P.pool, slots, tw, A);
        else if (P.sstream) fg_score_stream<2>(P.sstream, P.n_sstream, P.pool, slots, tw, A);
        else fg_exec<FG_MODE_SCORE, false>(P.ins_fast, P.n_ins, P.pool, slots, tw, A, nullptr, nullptr, 0, false);
        pri[pass] = A.prior; lik[pass] = A.lik + A.fac;
    }
    const double log_alpha = (pri[1] - pri[0]) + beta * (lik[1] - lik[0]);                   // smc.rs:678-679
    const double u = fg_rng_u01(rng);
    const bool accept = (log_alpha >= 0.0) || (u < exp(log_alpha));                          // smc.rs:680
    if (live) {
        if (accept) X.values[(long long)site * X.C + c] = fg_as_i64(prop);
        M.lprior[c] = accept ? pri[1] : pri[0];               // the freshly scored trace is returned either way
        M.ll[c] = accept ? lik[1] : lik[0];
    }
    // per-sweep proposal / accept counts, one pair of atomics per distinct site in the wave (a one-site model would
    // otherwise send a million atomics to one address)
    unsigned long long todo = __ballot(live);
    const unsigned long long acc_mask = __ballot(live && accept);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int s_lead = __builtin_amdgcn_readlane(site, leader);
        const unsigned long long same = __ballot(live && site == s_lead);
        if ((int)threadIdx.x == leader) {
            atomicAdd(&M.sw_n[s_lead], (unsigned int)__popcll(same));
            const unsigned int na = (unsigned int)__popcll(same & acc_mask);
            if (na) atomicAdd(&M.sw_a[s_lead], na);
        }
        todo &= ~same;
    }
}
// per-sweep batched DiminishingAdaptation update (see file header; oracle: adapt_update_batched)
__global__ void k_smc_adapt(FgSmcDev M, int S) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= S) return;
    const long long n = M.sw_n[j], a = M.sw_a[j];
    M.sw_n[j] = 0; M.sw_a[j] = 0;
    if (n <= 0) return;
    const long long T0 = M.tot[j];
    const long long tot = T0 + n, acc = M.acc[j] + a;
    M.tot[j] = tot; M.acc[j] = acc;
    if (tot < 10) return;
    const double rate = (double)acc / (double)tot, gamma = 0.7;
    const long long lo = T0 + 1 < 10 ? 10 : T0 + 1;
    double step = 0.0;
    if (tot - lo + 1 <= 64) { for (long long t = lo; t <= tot; ++t) step += 1.0 / pow((double)t, gamma); }
    else { const double hi = (double)tot, l = (double)lo, e = 1.0 - gamma;
           step = (pow(hi, e) - pow(l, e)) / e + 0.5 * (pow(l, -gamma) + pow(hi, -gamma)); }
    double ls = M.log_scale[j] + step * (rate - 0.44);
    const double ns = exp(ls);
    const double sc = (isfinite(ns) && ns > 0.0) ? fmin(fmax(ns, 0.001), 100.0) : 1.0;
    M.scale[j] = sc;
    M.log_scale[j] = (sc == 1.0) ? 0.0 : log(sc);
}

// ======================================================================================
// host side
// ======================================================================================
namespace {

struct Reducer {     // scratch for the two-pass reductions
    double *part_max = nullptr, *part_sum = nullptr;
    int init() {
        if (dev_alloc(&part_max, RED_BLOCKS) || dev_alloc(&part_sum, 2 * RED_BLOCKS)) return FG_E_HIP;
        return FG_OK;
    }
    void free_all() { if (part_max) (void)hipFree(part_max); if (part_sum) (void)hipFree(part_sum); part_max = part_sum = nullptr; }
    // lse / ESS of v = lw + (b - beta) ll, then k_smc_finish(phase)
    int run(hipStream_t s, const double *lw, const double *ll, long long n, FgSmcScalars *st, const double *b_ptr, int phase) {
        hipLaunchKernelGGL(k_smc_red_max, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, s, lw, ll, n, b_ptr, (const double *)&st->beta, part_max);
        hipLaunchKernelGGL(k_smc_red_sum, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, s, lw, ll, n, b_ptr, (const double *)&st->beta,
                           (const double *)part_max, part_sum);
        hipLaunchKernelGGL(k_smc_finish, dim3(1), dim3(RED_THREADS), 0, s, st, (const double *)part_max, (const double *)part_sum, RED_BLOCKS, n, phase);
        HIPCHK(hipGetLastError());
        return FG_OK;
    }
};

struct Scanner {     // scratch for the prefix sum
    double *chunk = nullptr, *cum = nullptr; long long cap = 0;
    int ensure(long long n) {
        if (n <= cap) return FG_OK;
        free_all();
        const long long nc = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
        if (dev_alloc(&chunk, (size_t)nc) || dev_alloc(&cum, (size_t)n)) return FG_E_HIP;
        cap = n;
        return FG_OK;
    }
    void free_all() { if (chunk) (void)hipFree(chunk); if (cum) (void)hipFree(cum); chunk = cum = nullptr; cap = 0; }
    int indices(hipStream_t s, const double *w, long long n, int method, double U, const double *d_u, unsigned long long seed,
                uint32_t step, long long *d_idx) {
        int rc = ensure(n);
        if (rc) return rc;
        const int nc = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
        hipLaunchKernelGGL(k_scan_chunk_sums, dim3(nc), dim3(SCAN_THREADS), 0, s, w, n, chunk);
        hipLaunchKernelGGL(k_scan_chunk_offsets, dim3(1), dim3(1), 0, s, chunk, nc);
        hipLaunchKernelGGL(k_scan_cumsum, dim3(nc), dim3(SCAN_THREADS), 0, s, w, n, (const double *)chunk, cum);
        hipLaunchKernelGGL(k_resample_search, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const double *)cum, n, method, U, d_u, seed,
                           step, d_idx);
        HIPCHK(hipGetLastError());
        return FG_OK;
    }
};

int set_device_or_fail(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fg_set_error("no HIP device available: no CPU fallback (FG_E_NO_DEVICE)"); return FG_E_NO_DEVICE; }
    if (device < 0 || device >= ndev) { fg_set_error("bad device ordinal"); return FG_E_BAD_ARG; }
    HIPCHK(hipSetDevice(device));
    return FG_OK;
}

}  // namespace

extern "C" {

void fg_smc_config_default(fg_smc_config *c) {      // SMCConfig::default, smc.rs:181-189
    if (!c) return;
    c->resampling_method = FG_RESAMPLE_SYSTEMATIC; c->ess_threshold = 0.5; c->rejuvenation_steps = 0;
}

// ---- standalone device primitives (no program needed) ----
int fg_device_log_sum_exp(int device, const double *h_x, int64_t n, double *out) {
    int rc = set_device_or_fail(device);
    if (rc) return rc;
    if (!out || n < 0) return FG_E_BAD_ARG;
    if (n == 0) { *out = -INFINITY; return FG_OK; }       // numerical.rs:16-18
    double *d_x = nullptr; FgSmcScalars *st = nullptr; Reducer R;
    if (dev_alloc(&d_x, (size_t)n) || dev_alloc(&st, 1) || R.init()) return FG_E_HIP;
    HIPCHK(hipMemcpy(d_x, h_x, (size_t)n * 8, hipMemcpyHostToDevice));
    rc = R.run(nullptr, d_x, nullptr, n, st, (const double *)&st->one, 4);
    FgSmcScalars h;
    if (!rc) { hipError_t e_ = hipMemcpy(&h, st, sizeof(h), hipMemcpyDeviceToHost); if (e_ != hipSuccess) rc = FG_E_HIP; else *out = h.lse1; }
    (void)hipFree(d_x); (void)hipFree(st); R.free_all();
    return rc;
}

int fg_device_next_beta(int device, double beta, const double *h_log_w, const double *h_ll, int64_t n, double target_ess,
                        double *out_beta) {
    int rc = set_device_or_fail(device);
    if (rc) return rc;
    if (!out_beta || n <= 0) return FG_E_BAD_ARG;
    double *d_lw = nullptr, *d_ll = nullptr; FgSmcScalars *st = nullptr; Reducer R;
    if (dev_alloc(&d_lw, (size_t)n) || dev_alloc(&d_ll, (size_t)n) || dev_alloc(&st, 1) || R.init()) return FG_E_HIP;
    FgSmcScalars h; std::memset(&h, 0, sizeof(h));
    h.beta = beta; h.one = 1.0; h.target_ess = target_ess;
    HIPCHK(hipMemcpy(st, &h, sizeof(h), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_lw, h_log_w, (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_ll, h_ll, (size_t)n * 8, hipMemcpyHostToDevice));
    rc = R.run(nullptr, d_lw, d_ll, n, st, (const double *)&st->one, 0);
    for (int it = 0; it < 64 && !rc; ++it) rc = R.run(nullptr, d_lw, d_ll, n, st, (const double *)&st->mid, 1);
    if (!rc) hipLaunchKernelGGL(k_smc_finish, dim3(1), dim3(RED_THREADS), 0, nullptr, st, (const double *)R.part_max, (const double *)R.part_sum, RED_BLOCKS, (long long)n, 2);
    if (!rc) { hipError_t e_ = hipMemcpy(&h, st, sizeof(h), hipMemcpyDeviceToHost); if (e_ != hipSuccess) rc = FG_E_HIP; else *out_beta = h.bnew; }
    (void)hipFree(d_lw); (void)hipFree(d_ll); (void)hipFree(st); R.free_all();
    return rc;
}

int fg_device_resample_indices(int device, int method, const double *h_weights, int64_t n, const double *h_u, int64_t *h_idx) {
    int rc = set_device_or_fail(device);
    if (rc) return rc;
    if (!h_weights || !h_u || !h_idx || n <= 0 || method < 0 || method > 2) return FG_E_BAD_ARG;
    double *d_w = nullptr, *d_u = nullptr; long long *d_idx = nullptr; Scanner S;
    const size_t nu = (method == FG_RESAMPLE_SYSTEMATIC) ? 1 : (size_t)n;
    if (dev_alloc(&d_w, (size_t)n) || dev_alloc(&d_u, nu) || dev_alloc(&d_idx, (size_t)n)) return FG_E_HIP;
    HIPCHK(hipMemcpy(d_w, h_weights, (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_u, h_u, nu * 8, hipMemcpyHostToDevice));
    rc = S.indices(nullptr, d_w, n, method, h_u[0], method == FG_RESAMPLE_SYSTEMATIC ? nullptr : d_u, 0, 0, d_idx);
    if (!rc) { hipError_t e_ = hipMemcpy(h_idx, d_idx, (size_t)n * 8, hipMemcpyDeviceToHost); if (e_ != hipSuccess) rc = FG_E_HIP; }
    (void)hipFree(d_w); (void)hipFree(d_u); (void)hipFree(d_idx); S.free_all();
    return rc;
}

// ---- adaptive_smc (smc.rs:455-581) ----
int fg_smc_run(fg_engine *e, const fg_smc_config *cfg, double *h_log_w, double *h_weights, fg_smc_result *res, double *h_betas,
               int max_betas) {
    NEED_ENGINE(e);
    if (!cfg || !res) return FG_E_BAD_ARG;
    if (cfg->resampling_method < 0 || cfg->resampling_method > 2 || cfg->rejuvenation_steps < 0) { fg_set_error("bad SMC config"); return FG_E_BAD_ARG; }
    const long long N = e->C;
    const int S = e->S, TB = 256, NB = (int)((N + TB - 1) / TB);
    hipStream_t s = e->stream;
    std::vector<void *> allocs;
    auto A = [&](auto **p, size_t n) { int rc = dev_alloc(p, n); if (!rc) allocs.push_back((void *)*p); return rc; };
    auto cleanup = [&]() { (void)hipStreamSynchronize(s); for (void *q : allocs) (void)hipFree(q); };
    FgSmcDev M{}; FgSmcScalars *st = nullptr; Reducer R; Scanner SC;
    double *d_lw = nullptr, *d_w = nullptr, *d_ll2 = nullptr, *d_lp2 = nullptr;
    long long *d_vals2 = nullptr, *d_idx = nullptr;
    const size_t Sn = (size_t)std::max(1, S);
    if (A(&M.ll, N) || A(&M.lprior, N) || A(&M.scale, Sn) || A(&M.log_scale, Sn) || A(&M.acc, Sn) || A(&M.tot, Sn) || A(&M.sw_n, Sn) ||
        A(&M.sw_a, Sn) || A(&st, 1) || A(&d_lw, N) || A(&d_w, N) || A(&d_ll2, N) || A(&d_lp2, N) || A(&d_vals2, Sn * N) || A(&d_idx, N) ||
        R.init()) { cleanup(); return FG_E_HIP; }
    allocs.push_back(R.part_max); allocs.push_back(R.part_sum);
    int rc = FG_OK;
#define SMC_TRY(x) do { rc = (x); if (rc) { cleanup(); SC.free_all(); return rc; } } while (0)
#define SMC_HIP(x) do { if ((x) != hipSuccess) { fg_set_error(#x); cleanup(); SC.free_all(); return FG_E_HIP; } } while (0)
    FgSmcScalars h; std::memset(&h, 0, sizeof(h));
    h.one = 1.0; h.beta = 0.0;
    h.target_ess = std::min(std::max(cfg->ess_threshold * (double)N, 1.0), (double)N);     // smc.rs:481
    SMC_HIP(hipMemcpyAsync(st, &h, sizeof(h), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((Sn + TB - 1) / TB)), dim3(TB), 0, s, M.scale, (long long)Sn, 1.0);
    // smc_prior_particles (smc.rs:764-790)
    SMC_TRY(fg_launch_prior(e, 0, FG_RNG_SMC_PRIOR, e->d_acc, nullptr));
    hipLaunchKernelGGL(k_smc_split_acc, dim3(NB), dim3(TB), 0, s, (const double *)e->d_acc, M.lprior, M.ll, N);
    hipLaunchKernelGGL(k_fill, dim3(NB), dim3(TB), 0, s, d_lw, N, -std::log((double)N));                  // smc.rs:476
    long long n_runs = N;
    int n_steps = 0;
    std::vector<double> betas;
    if (cfg->rejuvenation_steps == 0) {                      // single importance-sampling reweight: smc.rs:484-493
        hipLaunchKernelGGL(k_smc_is_weights, dim3(NB), dim3(TB), 0, s, d_lw, (const double *)M.ll, N);
        SMC_TRY(R.run(s, d_lw, nullptr, N, st, (const double *)&st->one, 3));      // log_evidence = lse(combined)
        betas.push_back(1.0); n_steps = 1;
    } else {
        double beta = 0.0;
        int steps = 0;
        while (beta < 1.0) {                                 // smc.rs:501-560
            steps += 1;
            // next_beta: ESS at b = 1, then 64 bisections on the device (smc.rs:588-622)
            SMC_TRY(R.run(s, d_lw, M.ll, N, st, (const double *)&st->one, 0));
            for (int it = 0; it < 64; ++it) SMC_TRY(R.run(s, d_lw, M.ll, N, st, (const double *)&st->mid, 1));
            if (steps >= 10000) { int one = 1; SMC_HIP(hipMemcpyAsync(&st->force_one, &one, sizeof(int), hipMemcpyHostToDevice, s)); }
            hipLaunchKernelGGL(k_smc_finish, dim3(1), dim3(RED_THREADS), 0, s, st, (const double *)R.part_max, (const double *)R.part_sum, RED_BLOCKS, N, 2);
            // reweight + evidence (smc.rs:512-529)
            SMC_TRY(R.run(s, d_lw, M.ll, N, st, (const double *)&st->bnew, 3));
            hipLaunchKernelGGL(k_smc_apply, dim3(NB), dim3(TB), 0, s, d_lw, (const double *)M.ll, d_w, N, (const FgSmcScalars *)st);
            hipLaunchKernelGGL(k_smc_set_beta, dim3(1), dim3(1), 0, s, st);
            SMC_HIP(hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, s));
            SMC_HIP(hipStreamSynchronize(s));
            beta = h.beta;
            betas.push_back(beta); n_steps++;
            if (beta < 1.0) {                                // resample + rejuvenate (smc.rs:534-559)
                double U = 0.0;
                if (cfg->resampling_method == FG_RESAMPLE_SYSTEMATIC) { FgStream rs = fg_stream(e->seed, 0, (uint32_t)steps, FG_RNG_SMC_RESAMPLE); U = fg_rng_u01(rs); }
                SMC_TRY(SC.indices(s, d_w, N, cfg->resampling_method, U, nullptr, e->seed, (uint32_t)steps, d_idx));
                hipLaunchKernelGGL(k_smc_gather, dim3(NB), dim3(TB), 0, s, (const long long *)e->d_values, d_vals2, (const double *)M.ll, d_ll2,
                                   (const double *)M.lprior, d_lp2, (const long long *)d_idx, S, N);
                SMC_HIP(hipMemcpyAsync(e->d_values, d_vals2, (size_t)S * N * 8, hipMemcpyDeviceToDevice, s));
                std::swap(M.ll, d_ll2); std::swap(M.lprior, d_lp2);
                hipLaunchKernelGGL(k_fill, dim3(NB), dim3(TB), 0, s, d_lw, N, -std::log((double)N));
                if (e->d > 0) {
                    SMC_TRY(set_lds(k_smc_rejuv, e->lds_score));
                    for (int r = 0; r < cfg->rejuvenation_steps; ++r) {
                        hipLaunchKernelGGL(k_smc_rejuv, dim3((unsigned)((N + e->tw - 1) / e->tw)), dim3(e->tw), e->lds_score, s, e->P, e->X, M,
                                           (const FgSmcScalars *)st, (uint32_t)((steps - 1) * cfg->rejuvenation_steps + r));
                        hipLaunchKernelGGL(k_smc_adapt, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, s, M, S);
                        n_runs += 2 * N;
                    }
                }
                SMC_HIP(hipGetLastError());
            }
        }
    }
    // attach the final normalised weights (smc.rs:565-575)
    SMC_TRY(R.run(s, d_lw, nullptr, N, st, (const double *)&st->one, 4));
    hipLaunchKernelGGL(k_smc_normalize, dim3(NB), dim3(TB), 0, s, d_lw, d_w, N, (const FgSmcScalars *)st);
    SMC_HIP(hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, s));
    if (h_log_w) SMC_HIP(hipMemcpyAsync(h_log_w, d_lw, (size_t)N * 8, hipMemcpyDeviceToHost, s));
    if (h_weights) SMC_HIP(hipMemcpyAsync(h_weights, d_w, (size_t)N * 8, hipMemcpyDeviceToHost, s));
    SMC_HIP(hipStreamSynchronize(s));
    res->log_evidence = h.log_evidence; res->n_steps = n_steps; res->n_model_runs = n_runs;
    if (h_betas) for (int i = 0; i < (int)betas.size() && i < max_betas; ++i) h_betas[i] = betas[i];
    cleanup();
    SC.free_all();
    return FG_OK;
#undef SMC_TRY
#undef SMC_HIP
}

}  // extern "C"
