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
#include "fg_cold.h"

#define RED_BLOCKS 512
#define RED_THREADS 256
#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_CHUNK (SCAN_THREADS * SCAN_ITEMS)

// (FgSmcScalars, FgSmcDev: fg_dev_types.h -- the rejuvenation kernel compiled at run time takes them too)
// few, large blocks: a pass is dominated by what follows the sums -- one ticket atomic per block, the last block's sweep over the
// blocks' partials -- not by the two exps per particle (512 x 256: 1.13 ms per run, 128 x 512: 1.00 ms, 2048 x 256: 2.3 ms)
#ifndef ESS_BLOCKS
#define ESS_BLOCKS 128
#endif
#ifndef ESS_THREADS
#define ESS_THREADS 512
#endif
#define ESS_MAXC 8

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
    if (phase == 3) {                                              // the reweight's log-normaliser; beta advances here (nothing reads the old one after the sums)
        st->log_norm = lse1; st->log_evidence += lse1;
        st->dbeta = st->bnew - st->beta; st->beta = st->bnew;
        return;
    }
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

// ---------------------------------------------------------------------------------------
// next_beta (smc.rs:588-622) without a launch per reduction: ONE kernel per pass evaluates ESS(b) for every candidate of
// the next THREE bisection levels (the 7 midpoints of the depth-3 decision tree over [lo, hi]; the first pass also
// b = 1), the last block to arrive walks the tree -- the same comparisons `ess_at(mid) < target` on the same midpoints
// 0.5 * (lo + hi) as 3 consecutive iterations of the reference's loop -- and writes the next pass's candidates.  64
// bisections = 22 passes instead of 195 launches.
// Each ESS(b) = exp(2 lse(v) - lse(2 v)), v_i = lw_i + (b - beta) ll_i, is accumulated as (max, sum exp(v - max),
// sum exp(v - max)^2) per thread, then combined by a fixed tree (lanes, waves, blocks in index order) with the usual
// rescaling, so the result does not depend on scheduling.  It differs from the reference's two-pass log_sum_exp in the last
// bits only (a partial maximum instead of the global one); these values DECIDE comparisons, they are never stored -- the
// reweight that follows uses the exact two-pass form (k_smc_red_max / k_smc_red_sum).
// The last block's single-thread epilogue of a pass: walk the decision tree with the candidates' ESS values and set up the
// next pass.
__device__ __forceinline__ void fg_ess_decide(FgSmcScalars *st, const double *ess_c, int nc) {
    int c0 = 0;
    if (st->first) {                                               // candidate 0 of the first pass is b = 1: smc.rs:604-607
        st->first = 0; c0 = 1;
        st->lo = st->beta; st->hi = 1.0;
        if (ess_c[0] >= st->target_ess) { st->done = 1; st->bnew = 1.0; st->ticket = 0u; return; }
    }
    // the remaining candidates are the midpoint tree of [lo, hi] in heap order: node j has children 2j + 1, 2j + 2
    const int ntree = nc - c0;
    int node = 0, depth = 0;
    double lo = st->lo, hi = st->hi;
    while (node < ntree && st->iters + depth < 64) {               // smc.rs:612-619, one level = one iteration
        const double mid = st->cand[c0 + node];
        if (ess_c[c0 + node] < st->target_ess) { hi = mid; node = 2 * node + 1; } else { lo = mid; node = 2 * node + 2; }
        ++depth;
    }
    st->lo = lo; st->hi = hi; st->iters += depth;
    if (st->iters >= 64) {                                          // smc.rs:620-621
        st->bnew = fmin(fmax(hi, st->beta + 1e-9), 1.0);
        st->done = 1;
    } else {                                                        // next pass: the midpoint tree of the new bracket
        const int left = 64 - st->iters, lv = left < 3 ? left : 3;
        double blo[7], bhi[7];
        blo[0] = lo; bhi[0] = hi;
        const int nn = (1 << lv) - 1;
        for (int j = 0; j < nn; ++j) {
            const double mid = 0.5 * (blo[j] + bhi[j]);             // smc.rs:613
            st->cand[j] = mid;
            if (2 * j + 2 < 7) { blo[2 * j + 1] = blo[j]; bhi[2 * j + 1] = mid; blo[2 * j + 2] = mid; bhi[2 * j + 2] = bhi[j]; }
        }
        st->n_cand = nn;
    }
    st->ticket = 0u;
}
struct EssAcc { double m, s1, s2; };
__device__ __forceinline__ EssAcc ess_combine(const EssAcc &a, const EssAcc &b) {
    EssAcc r;
    r.m = fmax(a.m, b.m);
    if (isinf(r.m) && r.m < 0.0) { r.s1 = 0.0; r.s2 = 0.0; return r; }
    const double ea = exp(a.m - r.m), eb = exp(b.m - r.m);           // one of them is exp(0) = 1
    r.s1 = a.s1 * ea + b.s1 * eb;
    r.s2 = a.s2 * (ea * ea) + b.s2 * (eb * eb);
    return r;
}
__device__ __forceinline__ EssAcc ess_shfl(const EssAcc &a, int o) {
    EssAcc r; r.m = __shfl_down(a.m, o, 64); r.s1 = __shfl_down(a.s1, o, 64); r.s2 = __shfl_down(a.s2, o, 64); return r;
}
__global__ __launch_bounds__(ESS_THREADS) void k_smc_ess_pass(const double *lw, const double *ll, long long n, FgSmcScalars *st, double *part /*[gridDim.x][ESS_MAXC][3]*/) {
    __shared__ double sh[ESS_THREADS / 64][ESS_MAXC][3];
    __shared__ int is_last;
    if (st->done) return;
    const int nc = st->n_cand;
    const double beta = st->beta;
    double bc[ESS_MAXC];
#pragma unroll
    for (int q = 0; q < ESS_MAXC; ++q) bc[q] = st->cand[q < nc ? q : 0] - beta;
    EssAcc A[ESS_MAXC];
#pragma unroll
    for (int q = 0; q < ESS_MAXC; ++q) { A[q].m = -INFINITY; A[q].s1 = 0.0; A[q].s2 = 0.0; }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double w = lw[i], l = ll[i];
#pragma unroll
        for (int q = 0; q < ESS_MAXC; ++q) {
            if (q < nc) {
                const double v = w + bc[q] * l;                    // smc.rs:593
                if (v > A[q].m) {                                   // a new running maximum: rescale what has been summed
                    const double e = exp(A[q].m - v);               // exp(-inf) = 0 on the first element
                    A[q].s1 = A[q].s1 * e + 1.0; A[q].s2 = A[q].s2 * (e * e) + 1.0; A[q].m = v;
                } else {                                            // NaN propagates like the reference's sum; -inf adds nothing
                    const double e = (isinf(v) && v < 0.0) ? 0.0 : exp(v - A[q].m);
                    A[q].s1 += e; A[q].s2 += e * e;
                }
            }
        }
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < ESS_MAXC; ++q) {
        if (q < nc) {
            for (int o = 32; o > 0; o >>= 1) A[q] = ess_combine(A[q], ess_shfl(A[q], o));
            if (lane == 0) { sh[wv][q][0] = A[q].m; sh[wv][q][1] = A[q].s1; sh[wv][q][2] = A[q].s2; }
        }
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)nc) {
        const int q = threadIdx.x;
        EssAcc r = { sh[0][q][0], sh[0][q][1], sh[0][q][2] };
        for (int k = 1; k < ESS_THREADS / 64; ++k) { const EssAcc o = { sh[k][q][0], sh[k][q][1], sh[k][q][2] }; r = ess_combine(r, o); }
        double *p = part + ((long long)blockIdx.x * ESS_MAXC + q) * 3;
        p[0] = r.m; p[1] = r.s1; p[2] = r.s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                            // release the partials ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int tk = atomicAdd(&st->ticket, 1u);         // ... before the arrival ticket
        is_last = tk == gridDim.x - 1;
        if (is_last) __threadfence();                               // acquire: every block's partials are visible
    }
    __syncthreads();
    if (!is_last) return;
    // ---- last block: combine the blocks' partials in index order (fixed tree), then walk the decision tree
    __shared__ double ess_c[ESS_MAXC];
    for (int q = 0; q < nc; ++q) {
        EssAcc r = { -INFINITY, 0.0, 0.0 };
        for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x) {
            const double *p = part + ((long long)b * ESS_MAXC + q) * 3;
            const EssAcc o = { __builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2) };
            r = ess_combine(r, o);
        }
        for (int o = 32; o > 0; o >>= 1) r = ess_combine(r, ess_shfl(r, o));
        __syncthreads();
        if (lane == 0) { sh[wv][0][0] = r.m; sh[wv][0][1] = r.s1; sh[wv][0][2] = r.s2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            EssAcc t = { sh[0][0][0], sh[0][0][1], sh[0][0][2] };
            for (int k = 1; k < ESS_THREADS / 64; ++k) { const EssAcc o = { sh[k][0][0], sh[k][0][1], sh[k][0][2] }; t = ess_combine(t, o); }
            const bool empty = isinf(t.m) && t.m < 0.0;
            const double lse1 = (empty || t.s1 == 0.0) ? -INFINITY : t.m + log(t.s1);          // numerical.rs:33-37
            const double lse2 = (empty || t.s2 == 0.0) ? -INFINITY : 2.0 * t.m + log(t.s2);
            ess_c[q] = (!isfinite(lse1) || !isfinite(lse2)) ? (double)n : exp(2.0 * lse1 - lse2);   // smc.rs:598-601
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) fg_ess_decide(st, ess_c, nc);
}

// The same pass when the incoming log-weights are UNIFORM (every next_beta call of adaptive_smc: log_w = -ln N after the
// previous step's resample, smc.rs:476,538-540): ESS(b) = (sum t_i)^2 / sum t_i^2 with t_i = exp((b - beta)(ll_i - L)),
// L = max ll (the common factor of the weights cancels).  The 7 midpoints of the depth-3 tree are equally spaced,
// b_k = lo + k delta, so t_ik = E_i R_i^k with E_i = exp((lo - beta)(ll_i - L)), R_i = exp(delta (ll_i - L)): two exps
// and seven multiplications per particle instead of seven exps.  The particle with ll_i = L contributes exactly 1 to every
// sum (no underflow of the whole sum); like k_smc_ess_pass these values only decide comparisons.
__global__ __launch_bounds__(ESS_THREADS) void k_smc_ess_pass_uniform(const double *ll, long long n, FgSmcScalars *st, const double *ll_max, double *part /*[gridDim.x][ESS_MAXC][3]*/) {
    __shared__ double sh[ESS_THREADS / 64][ESS_MAXC][2];
    __shared__ int is_last;
    if (st->done) return;
    const int nc = st->n_cand, first = st->first;
    const double beta = st->beta, L = *ll_max;
    const int nt = nc - first;                                       // tree candidates: 1, 3 or 7, heap order over [lo, hi]
    const int lv = nt >= 7 ? 3 : (nt >= 3 ? 2 : 1);
    const double lo = first ? beta : st->lo, hi = first ? 1.0 : st->hi;
    // heap node j of the tree sits at grid position k(j) of lo + k (hi - lo) / 2^lv:  lv = 3: 4, 2, 6, 1, 3, 5, 7;  lv = 2: 2, 1, 3;
    // lv = 1: 1.  (The candidates the decision walk RECORDS are the reference's 0.5 * (lo + hi) chains in st->cand; the grid
    // points used here agree with them to the last bit or two.)
    const double d0 = lo - beta, d1 = 1.0 - beta, dl = (hi - lo) / (double)(1 << lv);
    double s1[ESS_MAXC], s2[ESS_MAXC];
#pragma unroll
    for (int q = 0; q < ESS_MAXC; ++q) { s1[q] = 0.0; s2[q] = 0.0; }
    const bool allneg = isinf(L) && L < 0.0;
#define ESS_ADD(q, tv) { const double tv_ = (tv); s1[q] += tv_; s2[q] += tv_ * tv_; }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n && !allneg; i += (long long)gridDim.x * blockDim.x) {
        const double x = ll[i] - L;                                  // <= 0; -inf: the particle has no weight at any b > beta
        if (isinf(x) && x < 0.0) continue;
        const double E = exp(d0 * x), R = exp(dl * x);
        const double p1 = E * R, p2 = p1 * R, p3 = p2 * R;
        if (first) {                                                 // candidate 0 = b = 1, then the depth-3 tree over [beta, 1]
            ESS_ADD(0, exp(d1 * x))
            const double p4 = p3 * R, p5 = p4 * R, p6 = p5 * R, p7 = p6 * R;
            ESS_ADD(1, p4) ESS_ADD(2, p2) ESS_ADD(3, p6) ESS_ADD(4, p1) ESS_ADD(5, p3) ESS_ADD(6, p5) ESS_ADD(7, p7)
        } else if (lv == 3) {
            const double p4 = p3 * R, p5 = p4 * R, p6 = p5 * R, p7 = p6 * R;
            ESS_ADD(0, p4) ESS_ADD(1, p2) ESS_ADD(2, p6) ESS_ADD(3, p1) ESS_ADD(4, p3) ESS_ADD(5, p5) ESS_ADD(6, p7)
        } else if (lv == 2) { ESS_ADD(0, p2) ESS_ADD(1, p1) ESS_ADD(2, p3) }
        else ESS_ADD(0, p1)
    }
#undef ESS_ADD
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < ESS_MAXC; ++q) {
        if (q < nc) {
            for (int o = 32; o > 0; o >>= 1) { s1[q] += __shfl_down(s1[q], o, 64); s2[q] += __shfl_down(s2[q], o, 64); }
            if (lane == 0) { sh[wv][q][0] = s1[q]; sh[wv][q][1] = s2[q]; }
        }
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)nc) {
        const int q = threadIdx.x;
        double a = sh[0][q][0], b = sh[0][q][1];
        for (int k = 1; k < ESS_THREADS / 64; ++k) { a += sh[k][q][0]; b += sh[k][q][1]; }
        double *p = part + ((long long)blockIdx.x * ESS_MAXC + q) * 3;
        p[0] = 0.0; p[1] = a; p[2] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int tk = atomicAdd(&st->ticket, 1u);
        is_last = tk == gridDim.x - 1;
        if (is_last) __threadfence();
    }
    __syncthreads();
    if (!is_last) return;
    // last block: every candidate's sums over the blocks in ONE sweep (thread t takes blocks t, t + 256, ...; lanes, then waves,
    // in index order: a fixed tree)
    __shared__ double ess_c[ESS_MAXC];
    __shared__ double shl[ESS_THREADS / 64][ESS_MAXC][2];
    double a[ESS_MAXC], b[ESS_MAXC];
#pragma unroll
    for (int q = 0; q < ESS_MAXC; ++q) { a[q] = 0.0; b[q] = 0.0; }
    for (int bk = threadIdx.x; bk < (int)gridDim.x; bk += blockDim.x) {
        const double *p = part + (long long)bk * ESS_MAXC * 3;
#pragma unroll
        for (int q = 0; q < ESS_MAXC; ++q) if (q < nc) { a[q] += __builtin_nontemporal_load(p + 3 * q + 1); b[q] += __builtin_nontemporal_load(p + 3 * q + 2); }
    }
#pragma unroll
    for (int q = 0; q < ESS_MAXC; ++q) {
        if (q < nc) {
            for (int o = 32; o > 0; o >>= 1) { a[q] += __shfl_down(a[q], o, 64); b[q] += __shfl_down(b[q], o, 64); }
            if (lane == 0) { shl[wv][q][0] = a[q]; shl[wv][q][1] = b[q]; }
        }
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)nc) {
        const int q = threadIdx.x;
        double ta = shl[0][q][0], tb = shl[0][q][1];
        for (int k = 1; k < ESS_THREADS / 64; ++k) { ta += shl[k][q][0]; tb += shl[k][q][1]; }
        const double e = ta * ta / tb;
        ess_c[q] = (allneg || !(ta > 0.0) || !isfinite(e)) ? (double)n : e;                       // smc.rs:598-601
    }
    __syncthreads();
    if (threadIdx.x == 0) fg_ess_decide(st, ess_c, nc);
}

// ---- next_beta for UNIFORM incoming weights WITHOUT a cross-block epilogue --------------------------------------------------
// k_smc_ess_pass_uniform ends every pass with a ticket atomic per block, two fences and the last block's sweep + single-thread
// decision: a chain of device-scope round trips.  Here the kernel boundary IS the grid barrier: pass p + 1 starts with EVERY block
// combining the block partials of pass p (a fixed tree: the same numbers in every block) and walking the decision tree itself -- a
// few hundred redundant additions per block instead of the epilogue -- then forms its candidates' sums and leaves its own
// partials.  Bracket state and partials are double-buffered by pass parity; block 0 records the new bracket.  The decisions are
// fg_ess_decide's: the same comparisons `ESS(mid) < target` on the reference's midpoints 0.5 (lo + hi) (smc.rs:612-619), three
// levels per pass; pass 0 also evaluates b = 1 (smc.rs:604-607) and reduces the block maxima of ll itself.
struct FgEssBracket { double lo, hi, bnew; int iters, done, first, extra;     // extra: candidate 0 of the next pass is b = xb (first: b = 1)
                      double s1_hi, s1_one;      // sum_i exp((b - beta)(ll_i - max ll)) at b = hi and at b = 1: the reweight's log-sum-exp needs no pass of its own (k_smc_ess2_apply)
                      // the zoom passes (see fg_ess_step): ESS(za) >= target > ESS(zb) with the two values, the window of the next pass
                      // (candidates wl + j wd, j = 1 .. 7), the sections per pass, passes taken
                      double za, zb, fa, fb, wl, wd, xb;
                      int zmode, zk, zpass, pad; };
// Sum over the 64 lanes of a wave by DPP moves (row_shr 1, 2, 4, 8 inside each row of 16 lanes, then row_bcast 15 / 31 across the
// rows): the total lands in lane 63.  No LDS round trip per step -- __shfl_down on a double is two ds_bpermute_b32 and a wait, and
// the sixteen sums of a pass spent 4-5 us in them (tools/prof_smc_phases.sh) -- and a fixed tree: the same bits wherever it runs.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double fg_dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi, lo);                            // lanes without a source add +0.0
}
__device__ __forceinline__ double fg_wave_sum63(double v) {
    v = fg_dpp_add<0x111, 0xf>(v); v = fg_dpp_add<0x112, 0xf>(v); v = fg_dpp_add<0x114, 0xf>(v); v = fg_dpp_add<0x118, 0xf>(v);
    v = fg_dpp_add<0x142, 0xa>(v); v = fg_dpp_add<0x143, 0xc>(v);
    return v;
}
#ifndef ESS2_BLOCKS
#define ESS2_BLOCKS 256
#endif
#ifndef ESS2_THREADS
#define ESS2_THREADS 512
#endif
#ifndef ESS2_UNROLL
#define ESS2_UNROLL 4
#endif
static_assert(ESS2_BLOCKS <= ESS2_THREADS && ESS2_THREADS % 64 == 0, "fg_ess_collect: one thread per block partial");
// candidates of a pass over bracket B in heap order (node j has children 2 j + 1, 2 j + 2); returns their number
__device__ __forceinline__ int fg_ess_candidates(const FgEssBracket &B, double *cand) {
    int c0 = 0;
    if (B.first) { cand[0] = 1.0; c0 = 1; }
    else if (B.extra) { cand[0] = B.xb; c0 = 1; }
    if (B.zmode) {                                                  // eight sections of the window: its seven inner points, in the order the sums are kept
        const int ord[7] = {4, 2, 6, 1, 3, 5, 7};
        for (int q = 0; q < 7; ++q) cand[c0 + q] = B.wl + (double)ord[q] * B.wd;
        return c0 + 7;
    }
    const int left = 64 - B.iters, lv = B.first ? 3 : (left < 3 ? left : 3);
    const int nn = (1 << lv) - 1;
    double blo[7], bhi[7];
    blo[0] = B.lo; bhi[0] = B.hi;
    for (int j = 0; j < nn; ++j) {
        const double mid = 0.5 * (blo[j] + bhi[j]);                 // smc.rs:613
        cand[c0 + j] = mid;
        if (2 * j + 2 < 7) { blo[2 * j + 1] = blo[j]; bhi[2 * j + 1] = mid; blo[2 * j + 2] = mid; bhi[2 * j + 2] = bhi[j]; }
    }
    return c0 + nn;
}
// the window of the next zoom pass: 8 sections of width (zb - za) / zk around the point where the chord through (za, fa), (zb, fb)
// meets the target, kept inside [za, zb] (zk = 8: the whole bracket)
__device__ __forceinline__ void fg_ess_window(FgEssBracket &B, double target) {
    const double w = B.zb - B.za;
    double c = B.za + w * ((B.fa - target) / (B.fa - B.fb));
    if (!isfinite(c)) c = B.za + 0.5 * w;
    B.wd = w / (double)B.zk;
    double wl = c - 4.0 * B.wd;
    if (wl + 8.0 * B.wd > B.zb) wl = B.zb - 8.0 * B.wd;
    if (wl < B.za || B.zk == 8) wl = B.za;
    B.wl = wl;
}
// The bracket after a pass whose candidates had effective sample sizes ess_c (fg_ess_decide as a pure function).
//
// next_beta's answer (smc.rs:588-622) is hi after 64 halvings of [beta, 1], each decided by `ESS(mid) < target`.  ESS(b) does not
// increase with b (d ln ESS / d b = 2 (E_b[ll] - E_2b[ll]) <= 0 under uniform incoming weights), so every one of those decisions
// follows from WHERE mid lies relative to any pair za < zb with ESS(za) >= target > ESS(zb).  The zoom passes find such a pair a few
// units in the last place apart in ~6 passes instead of 18: each evaluates the seven inner points of eight equal sections of a window
// around the chord's root, 1/zk of the bracket wide per section (zk = 64, then x 8 per hit up to 4 096; a miss -- the sign change left
// or right of the window -- still moves that end and falls back to zk = 8, plain 8-section).  Then the 64 halvings are REPLAYED from
// [beta, 1]: mid <= za goes right, mid >= zb goes left; a midpoint strictly inside (za, zb), if one comes up, and the last levels are
// decided by evaluation again, by the plain passes below, whose first candidate is b = hi itself (its sum is the reweight's
// log-normaliser).  Same beta' as the loop of smc.rs:612-619 wherever the evaluated ESS is monotone; inside the few-ulp band where
// rounding makes it wiggle either is a root to working precision (the tests hold the ladder to 1e-9).  FG_SMC_ZOOM=0: plain passes only.
__device__ __forceinline__ FgEssBracket fg_ess_step(FgEssBracket B, const double *ess_c, const double *s1_c, double beta, double target, double n_particles) {
    double cand[ESS_MAXC];
    const int nc = fg_ess_candidates(B, cand);
    int c0 = 0;
    const bool was_first = B.first != 0;
    if (B.first) {                                                 // candidate 0 of the first pass is b = 1: smc.rs:604-607
        B.first = 0; c0 = 1;
        B.s1_one = s1_c[0]; B.s1_hi = s1_c[0];                     // hi = 1 until a midpoint replaces it
        if (ess_c[0] >= target) { B.done = 1; B.bnew = 1.0; return B; }
        B.za = beta; B.fa = n_particles; B.zb = 1.0; B.fb = ess_c[0];   // ESS(beta) = N: the incoming weights are uniform
    } else if (B.extra) { B.extra = 0; c0 = 1; B.s1_hi = s1_c[0]; }    // b = xb = hi
    if (B.zmode) {
        const int inv[8] = {0, 3, 1, 4, 0, 5, 2, 6};                // heap slot of the j-th inner point
        const double w_old = B.zb - B.za;
        int j = 1;
        while (j <= 7 && !(ess_c[c0 + inv[j]] < target)) ++j;       // the first point below the target
        if (j <= 7) {
            B.zb = B.wl + (double)j * B.wd; B.fb = ess_c[c0 + inv[j]];
            if (j >= 2) { B.za = B.wl + (double)(j - 1) * B.wd; B.fa = ess_c[c0 + inv[j - 1]]; }
        } else { B.za = B.wl + 7.0 * B.wd; B.fa = ess_c[c0 + inv[7]]; }
        const double w = B.zb - B.za;
        B.zk = (w <= 1.5 * B.wd) ? (B.zk >= 512 ? 4096 : B.zk * 8) : 8;
        B.zpass += 1;
        if (w < w_old && B.zpass < 12 && w > fabs(B.zb) * 0x1p-48) { fg_ess_window(B, target); return B; }
        // leave: replay the halvings that the pair decides
        B.zmode = 0;
        double lo = B.lo, hi = B.hi;
        int it = B.iters;
        while (it < 64) {
            const double mid = 0.5 * (lo + hi);                     // smc.rs:613
            if (mid == lo || mid == hi) { it = 64; break; }         // the fixed point of the loop: no later iteration changes lo or hi
            if (mid <= B.za) lo = mid; else if (mid >= B.zb) hi = mid; else break;
            ++it;
        }
        B.lo = lo; B.hi = hi; B.iters = it;
        B.extra = 1; B.xb = hi;                                     // the next pass evaluates b = hi (and up to three more levels)
        return B;
    }
    const int ntree = nc - c0;
    int node = 0, depth = 0;
    double lo = B.lo, hi = B.hi;
    while (node < ntree && B.iters + depth < 64) {                 // smc.rs:612-619, one level = one iteration
        const double mid = cand[c0 + node];
        if (ess_c[c0 + node] < target) { hi = mid; B.s1_hi = s1_c[c0 + node]; if (was_first) { B.zb = mid; B.fb = ess_c[c0 + node]; } node = 2 * node + 1; }
        else { lo = mid; if (was_first) { B.za = mid; B.fa = ess_c[c0 + node]; } node = 2 * node + 2; }
        ++depth;
    }
    // a pass that leaves the bracket where it found it has reached the fixed point of the bisection (lo and hi are adjacent doubles or
    // equal: every later midpoint, hence every later decision, repeats) -- the remaining iterations of smc.rs:612-619 change nothing
    const bool fixed = depth > 0 && lo == B.lo && hi == B.hi;
    B.lo = lo; B.hi = hi; B.iters += depth;
    if (fixed) B.iters = 64;
    if (B.iters >= 64) { B.bnew = fmin(fmax(hi, beta + 1e-9), 1.0); B.done = 1; }   // smc.rs:620-621
    else if (was_first && B.zk > 0 && B.zb > B.za) { B.zmode = 1; B.zpass = 0; fg_ess_window(B, target); }   // (zk = 0: plain passes only)
    return B;
}
// every candidate's ESS from the block partials of the previous pass: thread t takes blocks t, t + T, ...; lanes, then waves, in
// index order -- the same tree, hence the same bits, in every block
__device__ __forceinline__ void fg_ess_collect(const double *part /*[nb][ESS_MAXC][2]*/, int nb, int nc, long long n, bool allneg, double (*shl)[ESS_MAXC][2], double *ess_c, double *s1_c) {
    double a[ESS_MAXC], b[ESS_MAXC];
#pragma unroll
    for (int q = 0; q < ESS_MAXC; ++q) { a[q] = 0.0; b[q] = 0.0; }
    for (int bk = threadIdx.x; bk < nb; bk += blockDim.x) {
        const double *p = part + (long long)bk * ESS_MAXC * 2;
#pragma unroll
        for (int q = 0; q < ESS_MAXC; ++q) if (q < nc) { a[q] += p[2 * q]; b[q] += p[2 * q + 1]; }
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nwv = (nb + 63) >> 6;                                // waves that hold partials (the others would add zeros)
    if (wv < nwv) {
#pragma unroll
        for (int q = 0; q < ESS_MAXC; ++q) {                        // all candidates, no test per candidate: sixteen independent chains
            a[q] = fg_wave_sum63(a[q]); b[q] = fg_wave_sum63(b[q]);
            if (lane == 63) { shl[wv][q][0] = a[q]; shl[wv][q][1] = b[q]; }
        }
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)nc) {
        const int q = threadIdx.x;
        double ta = shl[0][q][0], tb = shl[0][q][1];
        for (int k = 1; k < nwv; ++k) { ta += shl[k][q][0]; tb += shl[k][q][1]; }
        const double e = ta * ta / tb;
        ess_c[q] = (allneg || !(ta > 0.0) || !isfinite(e)) ? (double)n : e;                       // smc.rs:598-601
        s1_c[q] = ta;
    }
    __syncthreads();
}
// pass `pass` of next_beta; beta_ptr: the current beta.  brk[2]: the bracket pass `pass` starts from is brk[pass & 1] once the
// previous pass's decision is folded in (pass 0 builds it from beta); part[2][ESS2_BLOCKS][ESS_MAXC][2]; lmax[2]: {max ll, 0}.
// FG_SMC_PROF (experiment builds, tools/prof_smc_phases.sh): 100 MHz real-time stamps of the phases of a pass, block 0 and the last block
#ifdef FG_SMC_PROF
__device__ long long fg_smc_prof[64][2][8];
#define FG_SMC_T(i) { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && pass < 64) fg_smc_prof[pass][blockIdx.x != 0][i] = wall_clock64(); }
#else
#define FG_SMC_T(i)
#endif
__global__ __launch_bounds__(ESS2_THREADS) void k_smc_ess2_pass(const double *ll, long long n, int pass, const double *beta_ptr, double target, const double *part_max, int n_pmax,
                                                                FgEssBracket *brk, double *part, double *lmax, int *host_flag, int flag_base, int zoom) {
    __shared__ double shl[ESS2_THREADS / 64][ESS_MAXC][2];
    __shared__ double ess_c[ESS_MAXC], s1_c[ESS_MAXC];
    __shared__ FgEssBracket shB;
    __shared__ double shm[ESS2_THREADS / 64];
    FG_SMC_T(0)
    const double beta = *beta_ptr;
    const int nb = (int)gridDim.x;
    double L;
    FgEssBracket B;
    if (pass == 0) {                                               // max ll from the block maxima (max is exact: any order)
        double m = -INFINITY;
        for (int k = threadIdx.x; k < n_pmax; k += blockDim.x) m = fmax(m, part_max[k]);
        L = block_reduce_max(m, shm);
        if (blockIdx.x == 0 && threadIdx.x == 0) lmax[0] = L;
        B.lo = beta; B.hi = 1.0; B.bnew = 1.0; B.iters = 0; B.done = 0; B.first = 1; B.extra = 0; B.s1_hi = 0.0; B.s1_one = 0.0;
        B.za = beta; B.zb = 1.0; B.fa = (double)n; B.fb = 0.0; B.wl = beta; B.wd = 0.0; B.xb = 1.0; B.zmode = 0; B.zk = zoom ? 64 : 0; B.zpass = 0; B.pad = 0;
    } else {
        L = lmax[0];
        const FgEssBracket P = brk[(pass - 1) & 1];                 // what pass - 1 started from
        if (P.done) B = P;
        else {
            double cand[ESS_MAXC];
            const int ncp = fg_ess_candidates(P, cand);
            FG_SMC_T(1)
            fg_ess_collect(part + (size_t)((pass - 1) & 1) * ESS2_BLOCKS * ESS_MAXC * 2, nb, ncp, n, isinf(L) && L < 0.0, shl, ess_c, s1_c);
            FG_SMC_T(2)
            if (threadIdx.x == 0) shB = fg_ess_step(P, ess_c, s1_c, beta, target, (double)n);
            __syncthreads();
            B = shB;
        }
    }
    FG_SMC_T(3)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        brk[pass & 1] = B;
        if (host_flag) { *host_flag = flag_base | (1 + (B.done ? 1 : 0)); __threadfence_system(); }   // pinned host memory: the host launches pass p + 1 when it sees pass p's flag
    }
    if (B.done) return;
    // the candidates' sums over this block's particles (the two-exp form of k_smc_ess_pass_uniform)
    const int first = B.first, x0 = (first || B.extra) ? 1 : 0;     // candidate 0: b = 1 (first pass) or b = xb
    const int left = 64 - B.iters, lv = B.zmode ? 3 : (first ? 3 : (left < 3 ? left : 3));
    const int nc = x0 + (1 << lv) - 1;
    const double lo = B.zmode ? B.wl : B.lo;
    const double d0 = lo - beta, d1 = (first ? 1.0 : B.xb) - beta, dl = B.zmode ? B.wd : (B.hi - B.lo) / (double)(1 << lv);
    double s1[ESS_MAXC], s2[ESS_MAXC];
#pragma unroll
    for (int q = 0; q < ESS_MAXC; ++q) { s1[q] = 0.0; s2[q] = 0.0; }
    const bool allneg = isinf(L) && L < 0.0;
#define ESS_ADD(q, tv) { const double tv_ = (tv); s1[q] += tv_; s2[q] += tv_ * tv_; }
    // ESS2_UNROLL particles per thread and trip: their loads go out together, then the arithmetic runs per particle in index order --
    // the sums are the same as one at a time
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n && !allneg; i0 += ESS2_UNROLL * stride) {
        double xs[ESS2_UNROLL];
#pragma unroll
        for (int u = 0; u < ESS2_UNROLL; ++u) { const long long i = i0 + u * stride; xs[u] = i < n ? ll[i] - L : -INFINITY; }
#pragma unroll
        for (int u = 0; u < ESS2_UNROLL; ++u) {
            const double x = xs[u];                                  // <= 0; -inf: the particle has no weight at any b > beta (or lies past the end)
            if (isinf(x) && x < 0.0) continue;
            if (x0) {
                ESS_ADD(0, exp(d1 * x))
                if (lv == 0) continue;
            }
            const double E = exp(d0 * x), R = exp(dl * x);
            const double p1 = E * R, p2 = p1 * R, p3 = p2 * R;
            if (lv == 3) {                                           // the depth-3 tree (or the window's seven inner points), heap order
                const double p4 = p3 * R, p5 = p4 * R, p6 = p5 * R, p7 = p6 * R;
                if (x0) { ESS_ADD(1, p4) ESS_ADD(2, p2) ESS_ADD(3, p6) ESS_ADD(4, p1) ESS_ADD(5, p3) ESS_ADD(6, p5) ESS_ADD(7, p7) }
                else { ESS_ADD(0, p4) ESS_ADD(1, p2) ESS_ADD(2, p6) ESS_ADD(3, p1) ESS_ADD(4, p3) ESS_ADD(5, p5) ESS_ADD(6, p7) }
            } else if (lv == 2) { if (x0) { ESS_ADD(1, p2) ESS_ADD(2, p1) ESS_ADD(3, p3) } else { ESS_ADD(0, p2) ESS_ADD(1, p1) ESS_ADD(2, p3) } }
            else { if (x0) ESS_ADD(1, p1) else ESS_ADD(0, p1) }
        }
    }
#undef ESS_ADD
    FG_SMC_T(4)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < ESS_MAXC; ++q) {                            // all candidates, no test per candidate (the unused ones are zeros)
        s1[q] = fg_wave_sum63(s1[q]); s2[q] = fg_wave_sum63(s2[q]);
        if (lane == 63) { shl[wv][q][0] = s1[q]; shl[wv][q][1] = s2[q]; }
    }
    __syncthreads();
    FG_SMC_T(5)
    if (threadIdx.x < (unsigned)nc) {
        const int q = threadIdx.x;
        double a = shl[0][q][0], b = shl[0][q][1];
        for (int k = 1; k < ESS2_THREADS / 64; ++k) { a += shl[k][q][0]; b += shl[k][q][1]; }
        double *p = part + ((size_t)(pass & 1) * ESS2_BLOCKS + blockIdx.x) * ESS_MAXC * 2 + 2 * q;
        p[0] = a; p[1] = b;
    }
    FG_SMC_T(6)
}
#ifdef FG_SMC_PROF
extern "C" int fg_debug_smc_prof(long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(fg_smc_prof), sizeof(long long) * 64 * 2 * 8) == hipSuccess ? 0 : -1; }
#endif
// after the last pass: fold its decision in and publish beta' (st->bnew)
// ... and the maximum of the reweight that follows: v_i = lw0 + (beta' - beta) ll_i is non-decreasing in ll_i (rounding is monotone),
// so max v = lw0 + (beta' - beta) max ll, formed by the same two operations -- the reduction pass over v is not needed (part_max
// [n_pmax] is what k_smc_red_sum reads the maximum from).
__global__ __launch_bounds__(ESS2_THREADS) void k_smc_ess2_final(long long n, int last_pass, int nb, const double *beta_ptr, double target, FgEssBracket *brk, const double *part,
                                                                 const double *lmax, FgSmcScalars *st, double lw0, double *part_max, int n_pmax) {
    __shared__ double sh_bnew;
    __shared__ double shl[ESS2_THREADS / 64][ESS_MAXC][2];
    __shared__ double ess_c[ESS_MAXC], s1_c[ESS_MAXC];
    const double beta = *beta_ptr, L = lmax[0];
    FgEssBracket P = brk[last_pass & 1];
    if (!P.done) {
        double cand[ESS_MAXC];
        const int ncp = fg_ess_candidates(P, cand);
        fg_ess_collect(part + (size_t)(last_pass & 1) * ESS2_BLOCKS * ESS_MAXC * 2, nb, ncp, n, isinf(L) && L < 0.0, shl, ess_c, s1_c);
        if (threadIdx.x == 0) P = fg_ess_step(P, ess_c, s1_c, beta, target, (double)n);
    }
    if (threadIdx.x == 0) {
        if (!P.done) P.bnew = fmin(fmax(P.hi, beta + 1e-9), 1.0);    // (not reached: 22 passes cover 64 iterations)
        st->bnew = st->force_one ? 1.0 : P.bnew;                     // smc.rs:504-506: the step cap forces beta = 1
        sh_bnew = st->bnew;
        st->done = 1; st->lo = P.lo; st->hi = P.hi; st->iters = P.iters;
        brk[last_pass & 1] = P;
    }
    __syncthreads();
    const double vmax = lw0 + (sh_bnew - beta) * L;                  // smc_v at the particle with the largest ll
    for (int k = threadIdx.x; k < n_pmax; k += blockDim.x) part_max[k] = vmax;
}
// What follows the last pass in ONE launch (adaptive_smc: the log-weights are uniform, lw0 = -ln N, at every step's start):
// every block folds the last pass's decision in (fg_ess_collect / fg_ess_step: the same tree, the same bits in every block) -> beta';
// the reweight's log-normaliser log_sum_exp(lw0 + (beta' - beta) ll) (smc.rs:512-518) is max + ln(sum) with max = lw0 + (beta' - beta)
// max ll (rounding is monotone) and sum = sum_i exp((beta' - beta)(ll_i - max ll)) -- the sum the pass that evaluated b = beta' already
// formed for ESS(b) and the bracket carries (s1_hi, s1_one); then lw <- v - log_norm, w <- exp(lw) (smc.rs:520-528) over the scan's
// chunks with the chunk totals of the resampling prefix sum (k_scan_chunk_sums' additions), and on the ladder's last step the sums
// of the final normalisation, exp(lw - max lw) (smc.rs:565-575).  Block 0 publishes the scalars -- also to pinned host memory: the host's
// look at beta needs no copy.  (A ticket that lets the LAST block publish costs a device-scope release per block: 89 us for 512 blocks.)  beta' = beta + 1e-9 (a bracket narrower
// than the guaranteed progress, smc.rs:621) is no candidate of any pass: need_sum is set and the host runs the separate kernels.
struct FgSmcHostScalars { double beta, log_evidence; int need_sum, pass1_done; int flag[64]; };   // flag[p]: (epoch << 2) | (1 + "the bracket pass p started from is final")
__global__ __launch_bounds__(SCAN_THREADS) void k_smc_ess2_apply(const double *ll, long long n, int last_pass, int nb, double target, const FgEssBracket *brk, const double *part,
                                                                  const double *lmax, FgSmcScalars *st, const double *beta_in, double *beta_out, double lw0, double *lw, double *w,
                                                                  double *chunk_sum, double *chunk_sum2, FgSmcHostScalars *hs, int force_sum) {
    __shared__ double shl[ESS2_THREADS / 64][ESS_MAXC][2];
    __shared__ double ess_c[ESS_MAXC], s1_c[ESS_MAXC];
    __shared__ double sh_sc[6];
    __shared__ double shr[SCAN_THREADS / 64];
    const double beta = *beta_in, L = lmax[0];
    const bool allneg = isinf(L) && L < 0.0;
    FgEssBracket P = brk[last_pass & 1];
    if (!P.done) {
        double cand[ESS_MAXC];
        const int ncp = fg_ess_candidates(P, cand);
        fg_ess_collect(part + (size_t)(last_pass & 1) * ESS2_BLOCKS * ESS_MAXC * 2, nb, ncp, n, allneg, shl, ess_c, s1_c);
        if (threadIdx.x == 0) P = fg_ess_step(P, ess_c, s1_c, beta, target, (double)n);
    }
    if (threadIdx.x == 0) {
        if (!P.done) P.bnew = fmin(fmax(P.hi, beta + 1e-9), 1.0);    // (not reached: 22 passes cover 64 iterations)
        const double bnew = st->force_one ? 1.0 : P.bnew;            // smc.rs:504-506: the step cap forces beta = 1
        double s1 = 0.0; int need = 0;
        if (bnew == 1.0) s1 = P.s1_one; else if (bnew == P.hi) s1 = P.s1_hi; else need = 1;
        if (force_sum) need = 1;                                     // (FG_SMC_FORCE_SUM=1: tests of the separate-kernels path)
        const double dbeta = bnew - beta;
        const double vmax = lw0 + dbeta * L;                         // smc_v at the particle with the largest ll
        const double lse1 = (allneg || s1 == 0.0) ? -INFINITY : vmax + log(s1);       // numerical.rs:33-37
        sh_sc[0] = bnew; sh_sc[1] = dbeta; sh_sc[2] = lse1; sh_sc[3] = need ? 1.0 : 0.0;
        sh_sc[4] = isfinite(lse1) ? vmax - lse1 : -log((double)n);   // max of the new log-weights (the same two operations as the particle's own)
    }
    __syncthreads();
    const double bnew = sh_sc[0], dbeta = sh_sc[1], log_norm = sh_sc[2], mfin = sh_sc[4];
    const bool need_sum = sh_sc[3] != 0.0, fin = isfinite(log_norm), last_step = bnew >= 1.0;
    const long long n_chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    // Per particle the arithmetic is elementwise: global loads and stores run over consecutive addresses (thread t takes elements t,
    // t + 256, ... of the chunk).  The chunk total keeps k_scan_chunk_sums' order -- thread t adds ITS eight consecutive weights, then the
    // block tree -- through a copy of the weights in LDS (the prefix sum that follows forms the same partial sums).
    __shared__ double shw[SCAN_CHUNK + SCAN_CHUNK / 8];            // (padded: row t of eight starts at 9 t)
    if (!need_sum)
        for (long long ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
            const long long base = ch * SCAN_CHUNK;
            double s2 = 0.0;
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; ++k) {
                const int o = k * SCAN_THREADS + (int)threadIdx.x;
                const long long i = base + o;
                double wi = 0.0;
                if (i < n) {
                    const double v = lw0 + dbeta * ll[i];
                    const double nl = fin ? v - log_norm : -log((double)n);
                    wi = exp(nl);
                    lw[i] = nl; w[i] = wi;
                    if (last_step) s2 += exp(nl - mfin);
                }
                shw[o + (o >> 3)] = wi;
            }
            __syncthreads();
            double sw = 0.0;
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; ++k) if (base + (long long)threadIdx.x * SCAN_ITEMS + k < n) sw += shw[9 * (int)threadIdx.x + k];
            sw = block_reduce_sum(sw, shr);
            if (last_step) s2 = block_reduce_sum(s2, shr);
            if (threadIdx.x == 0) { chunk_sum[ch] = sw; if (last_step) chunk_sum2[ch] = s2; }
        }
    // block 0 publishes: no block of this launch reads what it writes (beta comes in through beta_in, the new one goes to beta_out --
    // the host swaps the two slots from step to step -- and to st->beta for the rejuvenation sweeps)
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    st->bnew = bnew; st->done = 1; st->lo = P.lo; st->hi = P.hi; st->iters = P.iters;
    st->need_sum = need_sum ? 1 : 0;
    if (!need_sum) {
        st->lse1 = log_norm; st->log_norm = log_norm; st->log_evidence += log_norm;     // smc.rs:517-518, :529
        st->dbeta = dbeta; st->beta = bnew; *beta_out = bnew; st->fin_max = mfin;
    }
    if (hs) { hs->beta = need_sum ? beta : bnew; hs->log_evidence = st->log_evidence; hs->need_sum = need_sum ? 1 : 0; __threadfence_system(); }
}
// the final normalisation (smc.rs:565-575) behind k_smc_ess2_apply's last step: lse = max + ln(sum exp(lw - max)) from the chunk sums
// (a fixed tree: the same bits in every block), lw <- lw - lse, w <- exp(lw); uniform if lse is not finite
__global__ __launch_bounds__(SCAN_THREADS) void k_smc_final_norm(double *lw, double *w, long long n, const double *chunk_sum2, FgSmcScalars *st) {
    __shared__ double shr[SCAN_THREADS / 64];
    const int n_chunks = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
    double s = 0.0;
    for (int k = threadIdx.x; k < n_chunks; k += blockDim.x) s += chunk_sum2[k];
    s = block_reduce_sum(s, shr);
    const double m = st->fin_max;
    const bool empty = isinf(m) && m < 0.0;
    const double lse = (empty || s == 0.0) ? -INFINITY : m + log(s);
    if (blockIdx.x == 0 && threadIdx.x == 0) st->lse1 = lse;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (isfinite(lse)) { const double nz = lw[i] - lse; lw[i] = nz; w[i] = exp(nz); }
        else { lw[i] = -log((double)n); w[i] = 1.0 / (double)n; }
    }
}
__global__ __launch_bounds__(RED_THREADS) void k_smc_max_finish(const double *part_max, int nb, double *out) {   // max of the block maxima
    __shared__ double sh[RED_THREADS / 64];
    double m = -INFINITY;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) m = fmax(m, part_max[k]);
    m = block_reduce_max(m, sh);
    if (threadIdx.x == 0) *out = m;
}
// arms the lookahead search for the current beta: first pass = {b = 1} + the midpoint tree of [beta, 1]
__global__ void k_smc_ess_begin(FgSmcScalars *st) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st->done = 0; st->iters = 0; st->first = 1; st->ticket = 0u;
    const double lo = st->beta, hi = 1.0;
    double blo[7], bhi[7];
    blo[0] = lo; bhi[0] = hi;
    st->cand[0] = 1.0;
    for (int j = 0; j < 7; ++j) {
        const double mid = 0.5 * (blo[j] + bhi[j]);
        st->cand[1 + j] = mid;
        if (2 * j + 2 < 7) { blo[2 * j + 1] = blo[j]; bhi[2 * j + 1] = mid; blo[2 * j + 2] = mid; bhi[2 * j + 2] = bhi[j]; }
    }
    st->n_cand = 8;
}
__global__ void k_smc_ess_end(FgSmcScalars *st) {                   // smc.rs:504-506: the step cap forces beta = 1
    if (threadIdx.x == 0 && blockIdx.x == 0 && st->force_one) st->bnew = 1.0;
}

// lw <- combined - log_norm (or uniform), w <- exp(lw)      smc.rs:520-528,535
__global__ void k_smc_apply(double *lw, const double *ll, double *w, long long n, const FgSmcScalars *st) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = lw[i] + st->dbeta * ll[i];
    const double nl = isfinite(st->log_norm) ? v - st->log_norm : -log((double)n);
    lw[i] = nl;
    if (w) w[i] = exp(nl);
}
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
// the same, and the block maxima of ll for next_beta's first pass (k_smc_red_max's fold: fmax from -inf)
__global__ __launch_bounds__(RED_THREADS) void k_smc_split_acc_max(const double *acc, double *lprior, double *ll, long long n, double *part_max) {
    __shared__ double sh[RED_THREADS / 64];
    double m = -INFINITY;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        lprior[i] = acc[i];
        const double v = acc[n + i] + acc[2 * n + i];
        ll[i] = v;
        m = fmax(m, v);
    }
    m = block_reduce_max(m, sh);
    if (threadIdx.x == 0) part_max[blockIdx.x] = m;
}
// the run's scalars and DiminishingAdaptation::new for every site (scale 1, log-scale 0, no counts): mcmc_utils.rs:60-75
__global__ void k_smc_init(FgSmcScalars *st, FgSmcScalars h, FgSmcDev M, long long Sn) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) *st = h;
    if (j < Sn) { M.scale[j] = 1.0; M.log_scale[j] = 0.0; M.acc[j] = 0; M.tot[j] = 0; }
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
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_chunk_offsets(double *chunk_sum, int n_chunks) {   // exclusive scan of the chunk totals, one block
    __shared__ double sh[SCAN_THREADS];
    if (blockIdx.x != 0) return;
    const int per = (n_chunks + SCAN_THREADS - 1) / SCAN_THREADS;
    const int k0 = threadIdx.x * per, k1 = k0 + per < n_chunks ? k0 + per : n_chunks;
    double tot = 0.0;
    for (int k = k0; k < k1; ++k) tot += chunk_sum[k];
    sh[threadIdx.x] = tot;
    __syncthreads();
    for (int o = 1; o < SCAN_THREADS; o <<= 1) {             // Hillis-Steele inclusive scan of the thread totals (fixed order)
        const double t = (threadIdx.x >= (unsigned)o) ? sh[threadIdx.x - o] : 0.0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    double run = threadIdx.x ? sh[threadIdx.x - 1] : 0.0;
    for (int k = k0; k < k1; ++k) { const double t = chunk_sum[k]; chunk_sum[k] = run; run += t; }
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
// k_scan_chunk_offsets and k_scan_cumsum in one launch: every block repeats the scan of the chunk totals (the same additions in the same
// order: a few hundred per block) and keeps its own chunk's offset
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_cumsum_off(const double *w, long long n, const double *chunk_sum, int n_chunks, double *cum) {
    __shared__ double sh[SCAN_THREADS];
    __shared__ double sh_off;
    const int per = (n_chunks + SCAN_THREADS - 1) / SCAN_THREADS;
    const int k0 = threadIdx.x * per, k1 = k0 + per < n_chunks ? k0 + per : n_chunks;
    double tot = 0.0;
    for (int k = k0; k < k1; ++k) tot += chunk_sum[k];
    sh[threadIdx.x] = tot;
    __syncthreads();
    for (int o = 1; o < SCAN_THREADS; o <<= 1) {             // Hillis-Steele inclusive scan of the thread totals (fixed order)
        const double t = (threadIdx.x >= (unsigned)o) ? sh[threadIdx.x - o] : 0.0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    {
        double run = threadIdx.x ? sh[threadIdx.x - 1] : 0.0;
        for (int k = k0; k < k1; ++k) { if (k == (int)blockIdx.x) sh_off = run; run += chunk_sum[k]; }
    }
    __syncthreads();
    const double chunk_off = sh_off;
    __syncthreads();
    // the chunk's weights through LDS: global loads and stores over consecutive addresses, thread t's eight consecutive items from row t
    __shared__ double shw[SCAN_CHUNK + SCAN_CHUNK / 8];            // (padded: row t of eight starts at 9 t)
    const long long cbase = (long long)blockIdx.x * SCAN_CHUNK;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { const int o = k * SCAN_THREADS + (int)threadIdx.x; shw[o + (o >> 3)] = (cbase + o < n) ? w[cbase + o] : 0.0; }
    __syncthreads();
    double loc[SCAN_ITEMS];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { s += shw[9 * (int)threadIdx.x + k]; loc[k] = s; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < SCAN_THREADS; o <<= 1) {             // Hillis-Steele inclusive scan of the thread totals
        const double t = (threadIdx.x >= (unsigned)o) ? sh[threadIdx.x - o] : 0.0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    const double off = chunk_off + (threadIdx.x ? sh[threadIdx.x - 1] : 0.0);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) shw[9 * (int)threadIdx.x + k] = off + loc[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { const int o = k * SCAN_THREADS + (int)threadIdx.x; if (cbase + o < n) cum[cbase + o] = shw[o + (o >> 3)]; }
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
// The same index in two levels: the first chunk of the prefix sum whose LAST element reaches the threshold (the chunk ends in LDS: nine
// steps at LDS latency instead of nine round trips to L2), then the first such element inside that chunk (cum is non-decreasing: the
// same k).  20.7 -> 19.2 us per 1 048 576 outputs.  (Tried on top: the block's stretch of the prefix sum -- its 256 rising thresholds land in
// ~256 consecutive entries -- staged in LDS behind the searches of its first and last output, every thread searching there: 22.6 - 29.6 us,
// the two searches ahead of the staging are a serial prologue per block and the LDS for the stretch costs occupancy.)
__global__ __launch_bounds__(256) void k_resample_search2(const double *cum, long long n, int n_chunks, int method, double U, const double *u_arr,
                                                          unsigned long long seed, uint32_t step, long long *idx) {
    extern __shared__ double ends[];
    for (int c = threadIdx.x; c < n_chunks; c += blockDim.x) { const long long e = (long long)(c + 1) * SCAN_CHUNK; ends[c] = cum[(e < n ? e : n) - 1]; }
    __syncthreads();
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
    int cl = 0, ch = n_chunks;                                 // first chunk whose end >= thr
    while (cl < ch) { const int mid = (cl + ch) >> 1; if (ends[mid] >= thr) ch = mid; else cl = mid + 1; }
    if (cl >= n_chunks) { idx[j] = n - 1; return; }
    long long lo = (long long)cl * SCAN_CHUNK, hi = lo + SCAN_CHUNK < n ? lo + SCAN_CHUNK : n;
    hi -= 1;                                                   // cum[hi] >= thr is known: the answer lies in [lo, hi]
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (cum[mid] >= thr) hi = mid; else lo = mid + 1; }
    idx[j] = lo;
}
__global__ void k_smc_gather(const long long *src, long long *dst, const double *ll_src, double *ll_dst, const double *lp_src,
                             double *lp_dst, const long long *idx, int S, long long n) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long long a = idx[j];
    for (int s = 0; s < S; ++s) dst[(long long)s * n + j] = src[(long long)s * n + a];   // particles[i].clone()  smc.rs:537
    if (ll_dst) { ll_dst[j] = ll_src[a]; lp_dst[j] = lp_src[a]; }   // (null: a rejuvenation sweep follows, which scores every particle again)
}

// ---------------------------------------------------------------------------------------
// rejuvenation: tempered_single_site_mh (smc.rs:631-688), one move per particle
// ---------------------------------------------------------------------------------------
#define FG_SMC_WPB(SCORE) ((SCORE) < 0 ? 1 : ((SCORE) == 2 ? 4 : 16))   /* tiles (waves) per block of k_smc_rejuv<SCORE> */
// SCORE: 0 = score stream of fast Normals, 3 = + linear predictors / option selects / Categorical tables, 2 = + general
// distribution records, -1 = the interpreter (programs without a score stream).  The stream variants carry no interpreter
// code and run 4 tiles per 256-thread block.
// GT: a program whose tile exceeds a CU's LDS (or has more sites than the block histogram) -- one wave per block, the tile in the
// engine's global scratch (fg_engine.hip), the block's counts added straight into its row of M.blk.  adaptive_smc has
// no size limit in the reference (smc.rs:455-581, 631-713).
// vsrc: the resampled population (k_smc_gather's buffer) when this is the first sweep behind a resampling step -- the sweep reads it and
// leaves every site in X.values: the copy back costs no launch.  pmax: the block's maximum of the new log-likelihoods (next_beta's first
// pass reads the block maxima: no reduction launch either); null on all sweeps but a step's last.
template <int SCORE, bool GT = false>
__global__ __launch_bounds__(GT ? FG_WAVE : FG_SMC_WPB(SCORE) * FG_WAVE, (GT || SCORE == 2) ? 1 : (SCORE < 0 ? FG_MIN_WAVES : 4)) void k_smc_rejuv(FgProgramDev P, FgChainCtx X, FgSmcDev M, const FgSmcScalars *st,
                                                                                                              uint32_t move_id, const long long *vsrc, double *pmax) {
    extern __shared__ double lds_[];
    __shared__ unsigned int hist[2][GT ? 1 : FG_SMC_HIST];          // the block's proposal / accept counts per site
    __shared__ double shmax[GT ? 1 : FG_SMC_WPB(SCORE)];
    constexpr int tw = FG_WAVE;
    const int lane = threadIdx.x & (FG_WAVE - 1), wv = (int)(threadIdx.x >> 6);
    unsigned int *row = M.blk + (long long)blockIdx.x * 2 * M.S;
    if (GT) { for (int j = lane; j < 2 * M.S; j += FG_WAVE) row[j] = 0u; __threadfence_block(); }
    else { for (int j = (int)threadIdx.x; j < 2 * FG_SMC_HIST; j += (int)blockDim.x) (&hist[0][0])[j] = 0u; }
    __syncthreads();
    const long long chain = ((long long)blockIdx.x * (blockDim.x >> 6) + wv) * tw + lane;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    double *slots = (GT ? X.gtile + (size_t)blockIdx.x * X.gtile_rows * FG_WAVE : lds_ + (long long)wv * P.n_slots * tw) + lane;   // one tile per wave
    if (vsrc) { FgChainCtx Xs = X; Xs.values = const_cast<long long *>(vsrc); fg_load_values(P, Xs, c, slots, tw); }
    else fg_load_values(P, X, c, slots, tw);
    const double beta = st->beta;
    const uint32_t sk0 = (uint32_t)X.seed, sk1 = (uint32_t)(X.seed >> 32), gchain = X.chain0 + (uint32_t)c;
    FgStream rng = fg_stream(X.seed, gchain, move_id, FG_RNG_SMC_REJUV);
    unsigned long long ra, rb;
    fg_rng_block(rng, ra, rb);
    const int k = (int)fg_pick(ra, (uint32_t)P.d);            // f64_sites[rng.gen_range(0..len)]  smc.rs:650
    const int site = P.f64_site[k];                           // sorted site index (adaptation / values row)
    const double scale = M.scale[site];                       // get_scale  smc.rs:651
    const double z = fg_cold_normal_pair(sk0, sk1, gchain, 1u, move_id, FG_RNG_SMC_REJUV).a;      // Normal(0,1).sample  smc.rs:655 (block 1)
    const double cur = slots[k * tw];                         // LDS slot of coordinate k is k
    const double prop = cur + scale * z;
    double pri[2], lik[2];
    for (int pass = 0; pass < 2; ++pass) {                    // score current, then proposed: two model runs  smc.rs:662-675
        slots[k * tw] = pass ? prop : cur;
        FgAcc3 A = {0.0, 0.0, 0.0};
        if (SCORE >= 0) fg_score_stream<(SCORE < 0 ? 0 : SCORE)>(P.sstream, P.n_sstream, P.pool, slots, tw, A);
        else fg_exec<FG_MODE_SCORE, false>(P.ins_fast, P.n_ins, P.pool, slots, tw, A, nullptr, nullptr, 0, false);
        pri[pass] = A.prior; lik[pass] = A.lik + A.fac;
    }
    const double log_alpha = (pri[1] - pri[0]) + beta * (lik[1] - lik[0]);                   // smc.rs:678-679
    const double u = fg_cold_u01_pair(sk0, sk1, gchain, 2u, move_id, FG_RNG_SMC_REJUV).a;     // block 2
    const bool accept = (log_alpha >= 0.0) || (u < fg_cold_exp(log_alpha));                  // smc.rs:680
    const double ll_new = accept ? lik[1] : lik[0];
    if (live) {
        if (vsrc) for (int j = 0; j < P.S; ++j) X.values[(long long)j * X.C + c] = vsrc[(long long)j * X.C + c];
        if (accept) X.values[(long long)site * X.C + c] = fg_as_i64(prop);
        M.lprior[c] = accept ? pri[1] : pri[0];               // the freshly scored trace is returned either way
        M.ll[c] = ll_new;
    }
    if (pmax) {                                               // block maximum of ll (fmax from -inf like k_smc_red_max; max is exact in any order)
        double m = live ? ll_new : -INFINITY;
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
        if (GT) { if (lane == 0) pmax[blockIdx.x] = m; }
        else if (lane == 0) shmax[wv] = m;
    }
    // per-sweep proposal / accept counts: one LDS add per distinct site in the wave, one row of counts per block in HBM
    // (k_smc_adapt adds the rows) -- a one-site model would otherwise send a million global atomics to one address
    unsigned long long todo = __ballot(live);
    const unsigned long long acc_mask = __ballot(live && accept);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int s_lead = __builtin_amdgcn_readlane(site, leader);
        const unsigned long long same = __ballot(live && site == s_lead);
        if (lane == leader) {
            const unsigned int na = (unsigned int)__popcll(same & acc_mask);
            if (GT) { atomicAdd(&row[s_lead], (unsigned int)__popcll(same)); if (na) atomicAdd(&row[M.S + s_lead], na); }
            else { atomicAdd(&hist[0][s_lead], (unsigned int)__popcll(same)); if (na) atomicAdd(&hist[1][s_lead], na); }
        }
        todo &= ~same;
    }
    if (GT) return;
    __syncthreads();
    for (int j = (int)threadIdx.x; j < M.S; j += (int)blockDim.x) { row[j] = hist[0][j]; row[M.S + j] = hist[1][j]; }
    if (pmax && threadIdx.x == 0) { double m = shmax[0]; for (int k = 1; k < (int)(blockDim.x >> 6); ++k) m = fmax(m, shmax[k]); pmax[blockIdx.x] = m; }
}
// The reference's OWN rejuvenation order (fg_smc_config.sequential_adaptation; smc.rs:482,544-553,698-713): particle-major, and ONE
// DiminishingAdaptation that every move of every particle updates before the next one reads its scale -- a recurrence through all
// N x steps moves, sequential by construction.  One wave walks the particles in order (all lanes carry the same particle; lane 0
// stores): the moves, their random numbers and the update are k_smc_rejuv's / DiminishingAdaptation::update (mcmc_utils.rs:88-150)
// one at a time.  Orders of magnitude slower than the batched sweeps (a few us per move): the mode exists so that parity with
// the reference's semantics can be CHECKED (tests/test_gpu_smc.py against the oracle's unbatched form), not to be fast.
template <int SCORE, bool GT = false>
__global__ __launch_bounds__(FG_WAVE) void k_smc_rejuv_seq(FgProgramDev P, FgChainCtx X, FgSmcDev M, const FgSmcScalars *st, uint32_t move0, int n_moves) {
    extern __shared__ double lds_[];
    constexpr int tw = FG_WAVE;
    const int lane = threadIdx.x;
    double *slots = (GT ? X.gtile : lds_) + lane;
    const double beta = st->beta;
    const uint32_t sk0 = (uint32_t)X.seed, sk1 = (uint32_t)(X.seed >> 32);
    for (long long c = 0; c < X.C; ++c) {
        fg_load_values(P, X, c, slots, tw);
        const uint32_t gchain = X.chain0 + (uint32_t)c;
        for (int r = 0; r < n_moves; ++r) {
            const uint32_t move_id = move0 + (uint32_t)r;
            FgStream rng = fg_stream(X.seed, gchain, move_id, FG_RNG_SMC_REJUV);
            unsigned long long ra, rb;
            fg_rng_block(rng, ra, rb);
            const int k = (int)fg_pick(ra, (uint32_t)P.d);            // f64_sites[rng.gen_range(0..len)]  smc.rs:650
            const int site = P.f64_site[k];
            const double scale = __builtin_nontemporal_load(&M.scale[site]);     // get_scale: what the previous move left  smc.rs:651
            const double z = fg_cold_normal_pair(sk0, sk1, gchain, 1u, move_id, FG_RNG_SMC_REJUV).a;
            const double cur = slots[k * tw];
            const double prop = cur + scale * z;
            double pri[2], lik[2];
            for (int pass = 0; pass < 2; ++pass) {                    // score current, then proposed  smc.rs:662-675
                slots[k * tw] = pass ? prop : cur;
                FgAcc3 A = {0.0, 0.0, 0.0};
                if (SCORE >= 0) fg_score_stream<(SCORE < 0 ? 0 : SCORE)>(P.sstream, P.n_sstream, P.pool, slots, tw, A);
                else fg_exec<FG_MODE_SCORE, false>(P.ins_fast, P.n_ins, P.pool, slots, tw, A, nullptr, nullptr, 0, false);
                pri[pass] = A.prior; lik[pass] = A.lik + A.fac;
            }
            const double log_alpha = (pri[1] - pri[0]) + beta * (lik[1] - lik[0]);                   // smc.rs:678-679
            const double u = fg_cold_u01_pair(sk0, sk1, gchain, 2u, move_id, FG_RNG_SMC_REJUV).a;
            const bool accept = (log_alpha >= 0.0) || (u < fg_cold_exp(log_alpha));                  // smc.rs:680
            if (!accept) slots[k * tw] = cur;
            // DiminishingAdaptation::update(site, accepted)  mcmc_utils.rs:88-150
            const long long tot = M.tot[site] + 1, acc = M.acc[site] + (accept ? 1 : 0);
            double sc = scale, ls = M.log_scale[site];
            if (tot >= 10) {
                const double rate = (double)acc / (double)tot;
                ls += (1.0 / pow((double)tot, 0.7)) * (rate - 0.44);
                const double ns = exp(ls);
                sc = (isfinite(ns) && ns > 0.0) ? fmin(fmax(ns, 0.001), 100.0) : 1.0;
                ls = (sc == 1.0) ? 0.0 : log(sc);
            }
            if (lane == 0) {
                if (accept) X.values[(long long)site * X.C + c] = fg_as_i64(prop);
                M.lprior[c] = accept ? pri[1] : pri[0];
                M.ll[c] = accept ? lik[1] : lik[0];
                M.tot[site] = tot; M.acc[site] = acc; M.scale[site] = sc; M.log_scale[site] = ls;
            }
            __threadfence();                                          // the next move reads this site's state
        }
    }
}

// per-sweep batched DiminishingAdaptation update (see file header; oracle: adapt_update_batched): block j adds site j's
// per-block counts, thread 0 applies the update
__global__ __launch_bounds__(256) void k_smc_adapt(FgSmcDev M, int S, int n_blk) {
    __shared__ unsigned int shn[4], sha[4];
    const int j = blockIdx.x;
    unsigned int cn = 0, ca = 0;
    for (int b = threadIdx.x; b < n_blk; b += blockDim.x) { cn += M.blk[(long long)b * 2 * S + j]; ca += M.blk[(long long)b * 2 * S + S + j]; }
    for (int o = 32; o > 0; o >>= 1) { cn += __shfl_down(cn, o, 64); ca += __shfl_down(ca, o, 64); }
    if ((threadIdx.x & 63) == 0) { shn[threadIdx.x >> 6] = cn; sha[threadIdx.x >> 6] = ca; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const long long n = (long long)shn[0] + shn[1] + shn[2] + shn[3], a = (long long)sha[0] + sha[1] + sha[2] + sha[3];
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


// ---- standalone population primitives (smc.rs:230-233, 326-349, 698-790) ----
// in-order-by-construction plain sums: block partials (fixed tree), then one block adds the partials in index order
__global__ __launch_bounds__(RED_THREADS) void k_sum_partials(const double *x, long long n, int square, double *part) {
    __shared__ double sh[RED_THREADS / 64];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) { const double v = x[i]; s += square ? v * v : v; }
    s = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(RED_THREADS) void k_sum_finish(const double *part, int nb, double *out) {
    __shared__ double sh[RED_THREADS / 64];
    double s = 0.0;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) s += part[k];
    s = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) *out = s;
}
// normalize_particles (smc.rs:719-755): weight = exp(log_weight - lse) (uniform when every log-weight is -inf), then / sum
__global__ void k_smc_norm_exp(const double *lw, double *w, long long n, const FgSmcScalars *st) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double lse = st->lse1;
    w[i] = (isinf(lse) && lse < 0.0) ? 1.0 / (double)n : exp(lw[i] - lse);
}
__global__ void k_smc_norm_div(double *w, long long n, const FgSmcScalars *st, const double *sum) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double lse = st->lse1;
    if ((isinf(lse) && lse < 0.0) || !(*sum > 0.0)) return;        // the uniform fallback returns before the renormalisation
    w[i] = w[i] / *sum;
}
__global__ void k_copy_f64(double *dst, const double *src, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ======================================================================================
// host side
// ======================================================================================
namespace {

// The host's looks at a running ladder (pass 1's bracket, beta after a reweight) wait for a few microseconds of queued work: polling the
// stream returns as soon as it drains, where hipStreamSynchronize may put the thread to sleep first (10 - 20 us per look, four per run).
static hipError_t smc_wait(hipStream_t s) {
    for (;;) {
        const hipError_t q = hipStreamQuery(s);
        if (q != hipErrorNotReady) return q;
    }
}
struct Reducer {     // scratch for the two-pass reductions
    double *part_max = nullptr, *part_sum = nullptr, *ess_part = nullptr;
    double *ess2 = nullptr;      // k_smc_ess2_pass: part[2][ESS2_BLOCKS][ESS_MAXC][2] | lmax[2] | FgEssBracket brk[2]
    static size_t ess2_doubles() { return (size_t)2 * ESS2_BLOCKS * ESS_MAXC * 2 + 2 + 2 * sizeof(FgEssBracket) / 8 + 2; }
    int init() {
        if (dev_alloc(&part_max, RED_BLOCKS) || dev_alloc(&part_sum, 2 * RED_BLOCKS) || dev_alloc(&ess_part, (size_t)ESS_BLOCKS * ESS_MAXC * 3 + 8) || dev_alloc(&ess2, ess2_doubles())) return FG_E_HIP;
        return FG_OK;
    }
    void free_all() { if (part_max) (void)hipFree(part_max); if (part_sum) (void)hipFree(part_sum); if (ess_part) (void)hipFree(ess_part); if (ess2) (void)hipFree(ess2);
                      part_max = part_sum = ess_part = ess2 = nullptr; }
    // the passes alone: part_max[0 .. n_pmax) holds block maxima of ll (their producer's: k_smc_split_acc_max or the step's last
    // rejuvenation sweep); what follows them is k_smc_ess2_apply.  The host looks at pass 1's bracket through pinned memory.
    int ess2_passes(hipStream_t s, const double *ll, long long n, const double *beta_ptr, double target, int n_pmax, FgSmcHostScalars *hs_host,
                    FgSmcHostScalars *hs_dev, int epoch, int *last_out, int *nb_out, bool *ends_at_one) {
        double *part = ess2, *lmax = ess2 + (size_t)2 * ESS2_BLOCKS * ESS_MAXC * 2;
        FgEssBracket *brk = (FgEssBracket *)(lmax + 2);
        const int nb = (int)std::min<long long>(ESS2_BLOCKS, (n + ESS2_THREADS - 1) / ESS2_THREADS);
        static const int zoom = []() { const char *z = std::getenv("FG_SMC_ZOOM"); return (z && std::atoi(z) == 0) ? 0 : 1; }();
        // Pass p + 1 is queued when pass p's flag arrives (block 0 writes it before its sums: the queue is never empty), and none once a
        // flag says the bracket is final: how many passes a step takes -- two when ESS(1) >= target, 8 - 10 with the zoom passes, 19 - 22
        // without -- is only known on the device.
        const int base = epoch << 2;
        volatile int *flag = hs_host->flag;
        int launched = 0, seen = 0;
        bool done = false;
        *ends_at_one = false;
        while (!done && seen < 64) {
            if (launched <= seen) {
                hipLaunchKernelGGL(k_smc_ess2_pass, dim3(nb), dim3(ESS2_THREADS), 0, s, ll, n, launched, beta_ptr, target, (const double *)part_max, n_pmax, brk, part, lmax,
                                   &hs_dev->flag[launched], base, zoom);
                ++launched;
            }
            unsigned spins = 0;
            int v;
            while ((((v = flag[seen]) >> 2) != epoch)) {
                if ((++spins & 0xfffu) == 0u) {                     // a failed launch or a fault must not leave the host spinning
                    const hipError_t q = hipStreamQuery(s);
                    if (q != hipErrorNotReady && ((flag[seen] >> 2) != epoch)) { fg_set_error(q == hipSuccess ? "next_beta: a pass left no flag" : hipGetErrorString(q)); return FG_E_HIP; }
                }
            }
            done = (v & 3) == 2;
            if (done && seen == 1) *ends_at_one = true;             // ESS(1) >= target: beta' = 1 ends the ladder
            ++seen;
        }
        HIPCHK(hipGetLastError());
        if (!done) { fg_set_error("next_beta: no final bracket after 64 passes"); return FG_E_STATE; }
        *last_out = launched - 1; *nb_out = nb;
        return FG_OK;
    }
    // the reweight's log-sum-exp behind k_smc_ess2_final (the beta' = beta + 1e-9 corner): its maximum is already in part_max
    int run_sum_only(hipStream_t s, const double *lw, const double *ll, long long n, FgSmcScalars *st, const double *b_ptr, int phase) {
        hipLaunchKernelGGL(k_smc_red_sum, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, s, lw, ll, n, b_ptr, (const double *)&st->beta,
                           (const double *)part_max, part_sum);
        hipLaunchKernelGGL(k_smc_finish, dim3(1), dim3(RED_THREADS), 0, s, st, (const double *)part_max, (const double *)part_sum, RED_BLOCKS, n, phase);
        HIPCHK(hipGetLastError());
        return FG_OK;
    }
    // next_beta (smc.rs:588-622) -> st->bnew: ESS at b = 1, then 64 bisections, three levels per pass (k_smc_ess_pass)
    // uniform_lw: the caller guarantees that every lw_i is the same number (adaptive_smc: always) -> the two-exp pass
    int next_beta(hipStream_t s, const double *lw, const double *ll, long long n, FgSmcScalars *st, bool uniform_lw = false) {
        hipLaunchKernelGGL(k_smc_ess_begin, dim3(1), dim3(1), 0, s, st);
        const int nb = (int)std::min<long long>(ESS_BLOCKS, (n + ESS_THREADS - 1) / ESS_THREADS);
        if (uniform_lw) {
            hipLaunchKernelGGL(k_smc_red_max, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, s, ll, (const double *)nullptr, n, (const double *)&st->one, (const double *)&st->beta, part_max);
            hipLaunchKernelGGL(k_smc_max_finish, dim3(1), dim3(RED_THREADS), 0, s, (const double *)part_max, RED_BLOCKS, ess_part + (size_t)ESS_BLOCKS * ESS_MAXC * 3);
        }
        for (int pass = 0; pass < 23; ++pass) {      // pass 0: b = 1 and levels 1-3; passes 1..20: three levels each; pass 21: the 64th; one spare (a no-op once done)
            if (uniform_lw) hipLaunchKernelGGL(k_smc_ess_pass_uniform, dim3(nb), dim3(ESS_THREADS), 0, s, ll, n, st, (const double *)(ess_part + (size_t)ESS_BLOCKS * ESS_MAXC * 3), ess_part);
            else hipLaunchKernelGGL(k_smc_ess_pass, dim3(nb), dim3(ESS_THREADS), 0, s, lw, ll, n, st, ess_part);
        }
        hipLaunchKernelGGL(k_smc_ess_end, dim3(1), dim3(1), 0, s, st);
        HIPCHK(hipGetLastError());
        return FG_OK;
    }
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
    bool external = false;      // buffers belong to the caller's arena
    int ensure(long long n) {
        if (n <= cap) return FG_OK;
        if (external) { fg_set_error("scan scratch too small"); return FG_E_BAD_ARG; }
        free_all();
        const long long nc = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
        if (dev_alloc(&chunk, (size_t)nc) || dev_alloc(&cum, (size_t)n)) return FG_E_HIP;
        cap = n;
        return FG_OK;
    }
    void free_all() { if (!external) { if (chunk) (void)hipFree(chunk); if (cum) (void)hipFree(cum); } chunk = cum = nullptr; cap = 0; }
    void search(hipStream_t s, long long n, int method, double U, const double *d_u, unsigned long long seed, uint32_t step, long long *d_idx) {
        const long long nc = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
        if (nc <= 4096)        // (the chunk ends fit a block's LDS)
            hipLaunchKernelGGL(k_resample_search2, dim3((unsigned)((n + 255) / 256)), dim3(256), (size_t)nc * 8, s, (const double *)cum, n, (int)nc, method, U, d_u, seed, step, d_idx);
        else hipLaunchKernelGGL(k_resample_search, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const double *)cum, n, method, U, d_u, seed, step, d_idx);
    }
    int indices(hipStream_t s, const double *w, long long n, int method, double U, const double *d_u, unsigned long long seed,
                uint32_t step, long long *d_idx) {
        int rc = ensure(n);
        if (rc) return rc;
        const int nc = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
        hipLaunchKernelGGL(k_scan_chunk_sums, dim3(nc), dim3(SCAN_THREADS), 0, s, w, n, chunk);
        hipLaunchKernelGGL(k_scan_chunk_offsets, dim3(1), dim3(SCAN_THREADS), 0, s, chunk, nc);
        hipLaunchKernelGGL(k_scan_cumsum, dim3(nc), dim3(SCAN_THREADS), 0, s, w, n, (const double *)chunk, cum);
        search(s, n, method, U, d_u, seed, step, d_idx);
        HIPCHK(hipGetLastError());
        return FG_OK;
    }
};


// The population state and scratch of an engine's particles: one arena, allocated on first use and kept (hipMalloc / hipFree
// per run cost more than a run).  fg_smc_run and the standalone entry points share it.
struct SmcWs {
    FgSmcDev M{}; FgSmcScalars *st = nullptr; Reducer R; Scanner SC;
    double *d_lw = nullptr, *d_w = nullptr, *d_ll2 = nullptr, *d_lp2 = nullptr, *d_red = nullptr, *d_chunk2 = nullptr;
    long long *d_vals2 = nullptr, *d_idx = nullptr;
    size_t o_ls = 0, o_st = 0;
    char *base = nullptr;
};
int smc_workspace(fg_engine *e, SmcWs &W) {
    const long long N = e->C;
    const int S = e->S;
    const size_t Sn = (size_t)std::max(1, S);
    const long long n_chunks = (N + SCAN_CHUNK - 1) / SCAN_CHUNK;
    const size_t max_blk = (size_t)((N + FG_WAVE - 1) / FG_WAVE);          // k_smc_rejuv blocks at one tile per block
    size_t arena_off = 0;
    auto carve = [&](size_t bytes) { const size_t o = arena_off; arena_off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_ll = carve(N * 8), o_lp = carve(N * 8), o_scale = carve(Sn * 8), o_ls = carve(Sn * 8), o_acc = carve(Sn * 8), o_tot = carve(Sn * 8),
                 o_st = carve(sizeof(FgSmcScalars)), o_lw = carve(N * 8), o_w = carve(N * 8), o_ll2 = carve(N * 8), o_lp2 = carve(N * 8), o_vals2 = carve(Sn * N * 8),
                 o_idx = carve(N * 8), o_pmax = carve(std::max<size_t>(RED_BLOCKS, max_blk) * 8), o_psum = carve(2 * RED_BLOCKS * 8), o_ess = carve(((size_t)ESS_BLOCKS * ESS_MAXC * 3 + 8) * 8),
                 o_ess2 = carve(Reducer::ess2_doubles() * 8),
                 o_chunk = carve((size_t)n_chunks * 8), o_chunk2 = carve((size_t)n_chunks * 8), o_cum = carve(N * 8), o_blk = carve(max_blk * 2 * Sn * 4), o_red = carve(64);
    if (e->smc_arena_bytes < arena_off) {
        if (e->smc_arena) { HIPCHK(hipStreamSynchronize(e->stream)); HIPCHK(hipFree(e->smc_arena)); e->smc_arena = nullptr; e->smc_arena_bytes = 0; }
        HIPCHK(hipMalloc(&e->smc_arena, arena_off));
        e->smc_arena_bytes = arena_off;
        e->smc_pop_ready = false;
    }
    char *ar = (char *)e->smc_arena;
    W.base = ar; W.o_ls = o_ls; W.o_st = o_st;
    FgSmcDev &M = W.M;
    M.ll = (double *)(ar + o_ll); M.lprior = (double *)(ar + o_lp); M.scale = (double *)(ar + o_scale); M.log_scale = (double *)(ar + o_ls);
    M.acc = (long long *)(ar + o_acc); M.tot = (long long *)(ar + o_tot); M.sw_n = nullptr; M.sw_a = nullptr;
    M.blk = (unsigned int *)(ar + o_blk); M.S = S;
    W.st = (FgSmcScalars *)(ar + o_st);
    W.d_lw = (double *)(ar + o_lw); W.d_w = (double *)(ar + o_w); W.d_ll2 = (double *)(ar + o_ll2); W.d_lp2 = (double *)(ar + o_lp2);
    W.d_vals2 = (long long *)(ar + o_vals2); W.d_idx = (long long *)(ar + o_idx); W.d_red = (double *)(ar + o_red);
    W.R.part_max = (double *)(ar + o_pmax); W.R.part_sum = (double *)(ar + o_psum); W.R.ess_part = (double *)(ar + o_ess); W.R.ess2 = (double *)(ar + o_ess2);
    W.d_chunk2 = (double *)(ar + o_chunk2);
    W.SC.chunk = (double *)(ar + o_chunk); W.SC.cum = (double *)(ar + o_cum); W.SC.cap = N; W.SC.external = true;
    return FG_OK;
}

int set_device_or_fail(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fg_set_error("no HIP device available: no CPU fallback (FG_E_NO_DEVICE)"); return FG_E_NO_DEVICE; }
    if (device < 0 || device >= ndev) { fg_set_error("bad device ordinal"); return FG_E_BAD_ARG; }
    HIPCHK(hipSetDevice(device));
    return FG_OK;
}

}  // namespace

// the engine's global tile scratch [tiles][rows][64] (fg_engine.hip allocates it for programs beyond one CU's LDS; an engine that only
// meets the limit in rejuvenation -- more sites than the block histogram -- gets it here, once)
static int smc_global_tile(fg_engine *e) {
    if (e->d_gtile) return FG_OK;
    e->X.gtile_rows = e->n_slots + e->d + 2 + FG_MW_MAX;
    if (int rc = dev_alloc(&e->d_gtile, (size_t)((e->C + e->tw - 1) / e->tw) * e->X.gtile_rows * e->tw)) return rc;
    e->X.gtile = e->d_gtile;
    return FG_OK;
}

extern "C" {

void fg_smc_config_default(fg_smc_config *c) {      // SMCConfig::default, smc.rs:181-189
    if (!c) return;
    c->resampling_method = FG_RESAMPLE_SYSTEMATIC; c->ess_threshold = 0.5; c->rejuvenation_steps = 0; c->sequential_adaptation = 0;
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
    rc = R.next_beta(nullptr, d_lw, d_ll, n, st);
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
    SmcWs WS;
    if (int rc0 = smc_workspace(e, WS)) return rc0;
    const size_t Sn = (size_t)std::max(1, S);
    auto cleanup = [&]() { (void)hipStreamSynchronize(s); };
    FgSmcDev &M = WS.M; Reducer &R = WS.R; Scanner &SC = WS.SC;
    FgSmcScalars *st = WS.st;
    double *d_lw = WS.d_lw, *d_w = WS.d_w, *d_ll2 = WS.d_ll2, *d_lp2 = WS.d_lp2;
    long long *d_vals2 = WS.d_vals2, *d_idx = WS.d_idx;
    int rc = FG_OK;
#define SMC_TRY(x) do { rc = (x); if (rc) { cleanup(); SC.free_all(); return rc; } } while (0)
#define SMC_HIP(x) do { if ((x) != hipSuccess) { fg_set_error(#x); cleanup(); SC.free_all(); return FG_E_HIP; } } while (0)
    if (!e->smc_host) SMC_HIP(hipHostMalloc(&e->smc_host, sizeof(FgSmcHostScalars), hipHostMallocMapped));
    FgSmcHostScalars *hs = (FgSmcHostScalars *)e->smc_host, *hs_dev = nullptr;
    SMC_HIP(hipHostGetDevicePointer((void **)&hs_dev, hs, 0));
    std::memset(hs, 0, sizeof(*hs));
    FgSmcScalars h; std::memset(&h, 0, sizeof(h));
    h.one = 1.0; h.beta = 0.0;
    h.target_ess = std::min(std::max(cfg->ess_threshold * (double)N, 1.0), (double)N);     // smc.rs:481
    // the run's scalars, and scale = 1, log_scale = 0, acc = tot = 0 (DiminishingAdaptation::new): one launch
    hipLaunchKernelGGL(k_smc_init, dim3((unsigned)((Sn + TB - 1) / TB)), dim3(TB), 0, s, st, h, M, (long long)Sn);
    // smc_prior_particles (smc.rs:764-790): through the compiled model for a large population (the draw is a tenth of a 1 048 576-particle
    // run on the interpreter kernel; the unit is compiled once per program and cached)
    if (e->gt || N < (1LL << 18) || fg_jit_prior_launch(e, 0, FG_RNG_SMC_PRIOR, e->d_acc, nullptr, true) != FG_OK)
        SMC_TRY(fg_launch_prior(e, 0, FG_RNG_SMC_PRIOR, e->d_acc, nullptr));
    const double lw0 = -std::log((double)N);                                // log_w = -ln N at every step's start (smc.rs:476,538-540)
    long long n_runs = N;
    int n_steps = 0;
    bool fin_ready = false;                                  // the last reweight left the sums of the final normalisation (k_smc_ess2_apply)
    double log_evidence = 0.0; bool have_evidence = false, evidence_late = false;   // (late: the pinned copy is read behind the final synchronisation)
    std::vector<double> betas;
    if (cfg->rejuvenation_steps == 0) {                      // single importance-sampling reweight: smc.rs:484-493
        hipLaunchKernelGGL(k_smc_split_acc, dim3(NB), dim3(TB), 0, s, (const double *)e->d_acc, M.lprior, M.ll, N);
        hipLaunchKernelGGL(k_smc_is_weights, dim3(NB), dim3(TB), 0, s, d_lw, (const double *)M.ll, N);
        SMC_TRY(R.run(s, d_lw, nullptr, N, st, (const double *)&st->one, 3));      // log_evidence = lse(combined)
        betas.push_back(1.0); n_steps = 1;
    } else {
        // particle_log_likelihood (smc.rs:381-383) and the block maxima of ll for the first pass of next_beta
        hipLaunchKernelGGL(k_smc_split_acc_max, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, s, (const double *)e->d_acc, M.lprior, M.ll, N, R.part_max);
        int n_pmax = RED_BLOCKS;
        const int n_chunks = (int)((N + SCAN_CHUNK - 1) / SCAN_CHUNK);
        double *lmax = R.ess2 + (size_t)2 * ESS2_BLOCKS * ESS_MAXC * 2;
        FgEssBracket *brk = (FgEssBracket *)(lmax + 2);
        double beta = 0.0;
        int steps = 0;
        while (beta < 1.0) {                                 // smc.rs:501-560
            steps += 1;
            // next_beta: ESS at b = 1, then 64 bisections on the device (smc.rs:588-622)
            if (steps >= 10000) { int one = 1; SMC_HIP(hipMemcpyAsync(&st->force_one, &one, sizeof(int), hipMemcpyHostToDevice, s)); }
            int last = 0, nb = 0;
            const double *beta_in = &st->beta2[steps & 1]; double *beta_out = &st->beta2[(steps + 1) & 1];   // (both start at 0: k_smc_init)
            bool ends_at_one = false;
            e->smc_epoch = (e->smc_epoch + 1) & 0x0fffffff;
            if (e->smc_epoch == 0) e->smc_epoch = 1;                // (0 is what an unwritten flag reads as)
            SMC_TRY(R.ess2_passes(s, M.ll, N, beta_in, h.target_ess, n_pmax, hs, hs_dev, e->smc_epoch, &last, &nb, &ends_at_one));
            // the last decision, reweight + evidence (smc.rs:512-529), weights and the chunk totals of the resampling prefix sum: one launch
            hipLaunchKernelGGL(k_smc_ess2_apply, dim3((unsigned)n_chunks), dim3(SCAN_THREADS), 0, s, (const double *)M.ll, N, last, nb, h.target_ess,
                               (const FgEssBracket *)brk, (const double *)R.ess2, (const double *)lmax, st, beta_in, beta_out, lw0, d_lw, d_w, SC.chunk, WS.d_chunk2, hs_dev,
                               (std::getenv("FG_SMC_FORCE_SUM") && std::atoi(std::getenv("FG_SMC_FORCE_SUM")) != 0) ? 1 : 0);
            SMC_HIP(hipGetLastError());
            // ESS(1) >= target (pass 1 said so): beta' = 1 ends the ladder -- nothing to look at before the final normalisation is queued
            const bool ends = ends_at_one && steps < 10000 && !(std::getenv("FG_SMC_FORCE_SUM") && std::atoi(std::getenv("FG_SMC_FORCE_SUM")) != 0);
            if (!ends) SMC_HIP(smc_wait(s));
            const bool fused = ends || hs->need_sum == 0;
            if (ends) { beta = 1.0; have_evidence = true; evidence_late = true; fin_ready = true; }
            else if (fused) { beta = hs->beta; log_evidence = hs->log_evidence; have_evidence = true; fin_ready = beta >= 1.0; }
            else {                                           // beta' = beta + 1e-9: the separate kernels (maximum, sum, finish, apply)
                hipLaunchKernelGGL(k_smc_ess2_final, dim3(1), dim3(ESS2_THREADS), 0, s, N, last, nb, (const double *)&st->beta, h.target_ess, brk, (const double *)R.ess2,
                                   (const double *)lmax, st, lw0, R.part_max, RED_BLOCKS);
                hipLaunchKernelGGL(k_fill, dim3(NB), dim3(TB), 0, s, d_lw, N, lw0);
                SMC_TRY(R.run_sum_only(s, d_lw, M.ll, N, st, (const double *)&st->bnew, 3));
                hipLaunchKernelGGL(k_smc_apply, dim3(NB), dim3(TB), 0, s, d_lw, (const double *)M.ll, d_w, N, (const FgSmcScalars *)st);
                SMC_HIP(hipMemcpyAsync(beta_out, &st->beta, sizeof(double), hipMemcpyDeviceToDevice, s));
                SMC_HIP(hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, s));
                SMC_HIP(hipStreamSynchronize(s));
                beta = h.beta; have_evidence = false; fin_ready = false;
            }
            betas.push_back(beta); n_steps++;
            if (beta < 1.0) {                                // resample + rejuvenate (smc.rs:534-559)
                double U = 0.0;
                if (cfg->resampling_method == FG_RESAMPLE_SYSTEMATIC) { FgStream rs = fg_stream(e->seed, 0, (uint32_t)steps, FG_RNG_SMC_RESAMPLE); U = fg_rng_u01(rs); }
                if (fused) {                                 // the chunk totals are there: offsets + prefix sum in one launch, then the search
                    hipLaunchKernelGGL(k_scan_cumsum_off, dim3((unsigned)n_chunks), dim3(SCAN_THREADS), 0, s, (const double *)d_w, N, (const double *)SC.chunk, n_chunks, SC.cum);
                    SC.search(s, N, cfg->resampling_method, U, nullptr, (unsigned long long)e->seed, (uint32_t)steps, d_idx);
                } else SMC_TRY(SC.indices(s, d_w, N, cfg->resampling_method, U, nullptr, e->seed, (uint32_t)steps, d_idx));
                const int score = !e->P.sstream ? -1 : (e->P.sstream_kinds == 0 ? 0 : (e->P.sstream_gen ? 2 : 3));
                // a batched sweep of a score-stream program reads the resampled population where k_smc_gather left it and writes every site
                // back (no copy), scores every particle again (ll / log-prior need no gather) and leaves the block maxima of ll
                const bool sweep_io = e->d > 0 && !cfg->sequential_adaptation;      // (every batched sweep kernel takes vsrc / pmax: k_smc_rejuv<...>, k_smc_jit_rejuv)
                if (sweep_io)
                    hipLaunchKernelGGL(k_smc_gather, dim3(NB), dim3(TB), 0, s, (const long long *)e->d_values, d_vals2, (const double *)nullptr, (double *)nullptr,
                                       (const double *)nullptr, (double *)nullptr, (const long long *)d_idx, S, N);
                else {
                    hipLaunchKernelGGL(k_smc_gather, dim3(NB), dim3(TB), 0, s, (const long long *)e->d_values, d_vals2, (const double *)M.ll, d_ll2,
                                       (const double *)M.lprior, d_lp2, (const long long *)d_idx, S, N);
                    SMC_HIP(hipMemcpyAsync(e->d_values, d_vals2, (size_t)S * N * 8, hipMemcpyDeviceToDevice, s));
                    SMC_HIP(hipMemcpyAsync(M.ll, d_ll2, (size_t)N * 8, hipMemcpyDeviceToDevice, s));
                    SMC_HIP(hipMemcpyAsync(M.lprior, d_lp2, (size_t)N * 8, hipMemcpyDeviceToDevice, s));
                }
                bool have_pmax = false;
                if (e->d > 0) {
                    // tiles (waves) per block: as many as the kernel is built for and 150 KB of LDS hold; a tile beyond one CU's LDS (or more
                    // sites than the block histogram has) lives in the engine's global scratch, one wave per block
                    const bool big = e->lds_score > 150 * 1024 || S > FG_SMC_HIST;
                    if (big) SMC_TRY(smc_global_tile(e));
                    const int wpb = big ? 1 : (int)std::max<size_t>(1, std::min<size_t>((size_t)FG_SMC_WPB(score), (150 * 1024) / std::max<size_t>(1, e->lds_score)));
                    const size_t lds_r = big ? 0 : e->lds_score * wpb;
                    const unsigned nblk = (unsigned)((N + (long long)e->tw * wpb - 1) / ((long long)e->tw * wpb));
                    if (cfg->sequential_adaptation) {               // the reference's order: one wave, particle by particle (k_smc_rejuv_seq)
                        const uint32_t mv0 = (uint32_t)((steps - 1) * cfg->rejuvenation_steps);
#define SMC_SEQ(SC_) do { if (big) hipLaunchKernelGGL((k_smc_rejuv_seq<SC_, true>), dim3(1), dim3(FG_WAVE), 0, s, e->P, e->X, M, (const FgSmcScalars *)st, mv0, cfg->rejuvenation_steps); \
                          else { SMC_TRY(set_lds(k_smc_rejuv_seq<SC_>, std::max<size_t>(e->lds_score, 64 * 1024 + 1))); \
                                 hipLaunchKernelGGL(k_smc_rejuv_seq<SC_>, dim3(1), dim3(FG_WAVE), e->lds_score, s, e->P, e->X, M, (const FgSmcScalars *)st, mv0, cfg->rejuvenation_steps); } } while (0)
                        if (score == 0) SMC_SEQ(0); else if (score == 3) SMC_SEQ(3); else if (score == 2) SMC_SEQ(2); else SMC_SEQ(-1);
#undef SMC_SEQ
                        n_runs += 2 * N * cfg->rejuvenation_steps;
                    } else
                    for (int r = 0; r < cfg->rejuvenation_steps; ++r) {
                        const uint32_t mv = (uint32_t)((steps - 1) * cfg->rejuvenation_steps + r);
                        const long long *vsrc = (sweep_io && r == 0) ? (const long long *)d_vals2 : (const long long *)nullptr;
                        double *pmax = (sweep_io && r == cfg->rejuvenation_steps - 1) ? R.part_max : (double *)nullptr;
#define SMC_REJUV(SC_) do { if (big) hipLaunchKernelGGL((k_smc_rejuv<SC_, true>), dim3(nblk), dim3(FG_WAVE), 0, s, e->P, e->X, M, (const FgSmcScalars *)st, mv, vsrc, pmax); \
                            else { SMC_TRY(set_lds(k_smc_rejuv<SC_>, std::max<size_t>(lds_r, 64 * 1024 + 1))); \
                                   hipLaunchKernelGGL(k_smc_rejuv<SC_>, dim3(nblk), dim3(e->tw * wpb), lds_r, s, e->P, e->X, M, (const FgSmcScalars *)st, mv, vsrc, pmax); } } while (0)
                        unsigned nb_adapt = nblk;
                        // the compiled model where there is one: programs without a score stream, and stream programs with general / option-select /
                        // Categorical records (hier_scale 3.2 -> 2.3 ms per 262 144-particle run, mixture 3.4 -> 2.4); fast-Normal streams keep k_smc_rejuv<0>
                        // (... and k_smc_rejuv<0> too once the unit exists -- a large population had it built for its prior draw --: 28.6 -> 24.7 us per sweep
                        // of 1 048 576 particles)
                        if ((score != 0 || (e->jit_state == 1 && N >= (1LL << 18))) && !big &&
                            fg_smc_jit_rejuv_launch(e, M, (const FgSmcScalars *)st, mv, &nb_adapt, vsrc, pmax) == FG_OK) { }
                        else if (score == 0) SMC_REJUV(0); else if (score == 3) SMC_REJUV(3); else if (score == 2) SMC_REJUV(2);
                        else SMC_REJUV(-1);
#undef SMC_REJUV
                        if (pmax) { have_pmax = true; n_pmax = (int)nb_adapt; }
                        hipLaunchKernelGGL(k_smc_adapt, dim3((unsigned)S), dim3(256), 0, s, M, S, (int)nb_adapt);
                        n_runs += 2 * N;
                    }
                }
                if (!have_pmax) {                            // the block maxima of ll for the next step's first pass
                    hipLaunchKernelGGL(k_smc_red_max, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, s, (const double *)M.ll, (const double *)nullptr, N, (const double *)&st->one, (const double *)&st->one, R.part_max);
                    n_pmax = RED_BLOCKS;
                }
                SMC_HIP(hipGetLastError());
            }
        }
    }
    // attach the final normalised weights (smc.rs:565-575)
    if (fin_ready) hipLaunchKernelGGL(k_smc_final_norm, dim3((unsigned)std::min<long long>(1024, (N + 4 * SCAN_THREADS - 1) / (4 * SCAN_THREADS))), dim3(SCAN_THREADS), 0, s, d_lw, d_w, N, (const double *)WS.d_chunk2, st);
    else {
        SMC_TRY(R.run(s, d_lw, nullptr, N, st, (const double *)&st->one, 4));
        hipLaunchKernelGGL(k_smc_normalize, dim3(NB), dim3(TB), 0, s, d_lw, d_w, N, (const FgSmcScalars *)st);
    }
    if (!have_evidence) SMC_HIP(hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, s));
    if (h_log_w) SMC_HIP(hipMemcpyAsync(h_log_w, d_lw, (size_t)N * 8, hipMemcpyDeviceToHost, s));
    if (h_weights) SMC_HIP(hipMemcpyAsync(h_weights, d_w, (size_t)N * 8, hipMemcpyDeviceToHost, s));
    SMC_HIP(hipGetLastError());
    SMC_HIP(hipStreamSynchronize(s));
    if (have_evidence) h.log_evidence = evidence_late ? hs->log_evidence : log_evidence;
    e->smc_pop_ready = true;
    res->log_evidence = h.log_evidence; res->n_steps = n_steps; res->n_model_runs = n_runs;
    if (h_betas) for (int i = 0; i < (int)betas.size() && i < max_betas; ++i) h_betas[i] = betas[i];
    cleanup();
    SC.free_all();
    return FG_OK;
#undef SMC_TRY
#undef SMC_HIP
}


// ---- the reference's standalone SMC building blocks over the engine's population (values [S][N] in the engine, log-weights /
// weights / log-likelihoods in the engine's SMC arena) -------------------------------------------------------------------
static int smc_pop(fg_engine *e, SmcWs &W, bool need_ready) {
    if (int rc = smc_workspace(e, W)) return rc;
    if (need_ready && !e->smc_pop_ready) { fg_set_error("no particle population: call fg_smc_prior_particles (or fg_smc_run) first"); return FG_E_STATE; }
    return FG_OK;
}
static int smc_normalize_impl(fg_engine *e, SmcWs &W) {                  // normalize_particles (smc.rs:719-755)
    const long long N = e->C;
    const int TB = 256, NB = (int)((N + TB - 1) / TB);
    hipStream_t s = e->stream;
    FgSmcScalars h; std::memset(&h, 0, sizeof(h)); h.one = 1.0;
    HIPCHK(hipMemcpyAsync(W.st, &h, sizeof(h), hipMemcpyHostToDevice, s));
    if (int rc = W.R.run(s, W.d_lw, nullptr, N, W.st, (const double *)&W.st->one, 4)) return rc;      // lse1 = log_sum_exp(log_weights)
    hipLaunchKernelGGL(k_smc_norm_exp, dim3(NB), dim3(TB), 0, s, (const double *)W.d_lw, W.d_w, N, (const FgSmcScalars *)W.st);
    hipLaunchKernelGGL(k_sum_partials, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, s, (const double *)W.d_w, N, 0, W.R.part_sum);
    hipLaunchKernelGGL(k_sum_finish, dim3(1), dim3(RED_THREADS), 0, s, (const double *)W.R.part_sum, RED_BLOCKS, W.d_red);
    hipLaunchKernelGGL(k_smc_norm_div, dim3(NB), dim3(TB), 0, s, W.d_w, N, (const FgSmcScalars *)W.st, (const double *)W.d_red);
    HIPCHK(hipGetLastError());
    return FG_OK;
}

int fg_smc_prior_particles(fg_engine *e, uint32_t iteration) {           // smc_prior_particles (smc.rs:764-790)
    NEED_ENGINE(e);
    SmcWs W;
    if (int rc = smc_pop(e, W, false)) return rc;
    const long long N = e->C;
    const int TB = 256, NB = (int)((N + TB - 1) / TB);
    if (int rc = fg_launch_prior(e, iteration, FG_RNG_SMC_PRIOR, e->d_acc, nullptr)) return rc;
    hipLaunchKernelGGL(k_smc_split_acc, dim3(NB), dim3(TB), 0, e->stream, (const double *)e->d_acc, W.M.lprior, W.M.ll, N);
    hipLaunchKernelGGL(k_copy_f64, dim3(NB), dim3(TB), 0, e->stream, W.d_lw, (const double *)W.M.ll, N);      // log_weight = log_likelihood + log_factors (FG-03)
    if (int rc = smc_normalize_impl(e, W)) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    e->smc_pop_ready = true;
    return FG_OK;
}
int fg_smc_normalize(fg_engine *e) {
    NEED_ENGINE(e);
    SmcWs W;
    if (int rc = smc_pop(e, W, true)) return rc;
    if (int rc = smc_normalize_impl(e, W)) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    return FG_OK;
}
int fg_smc_ess(fg_engine *e, double *out_ess) {                         // effective_sample_size (smc.rs:230-233): 1 / sum w^2
    NEED_ENGINE(e);
    if (!out_ess) return FG_E_BAD_ARG;
    SmcWs W;
    if (int rc = smc_pop(e, W, true)) return rc;
    hipLaunchKernelGGL(k_sum_partials, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, e->stream, (const double *)W.d_w, e->C, 1, W.R.part_sum);
    hipLaunchKernelGGL(k_sum_finish, dim3(1), dim3(RED_THREADS), 0, e->stream, (const double *)W.R.part_sum, RED_BLOCKS, W.d_red);
    double s2 = 0.0;
    HIPCHK(hipMemcpyAsync(&s2, W.d_red, 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    *out_ess = 1.0 / s2;
    return FG_OK;
}
int fg_smc_get_weights(fg_engine *e, double *h_log_w, double *h_w) {
    NEED_ENGINE(e);
    SmcWs W;
    if (int rc = smc_pop(e, W, true)) return rc;
    if (h_log_w) HIPCHK(hipMemcpyAsync(h_log_w, W.d_lw, (size_t)e->C * 8, hipMemcpyDeviceToHost, e->stream));
    if (h_w) HIPCHK(hipMemcpyAsync(h_w, W.d_w, (size_t)e->C * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return FG_OK;
}
int fg_smc_set_log_weights(fg_engine *e, const double *h_log_w) {       // a caller's own reweighting step; follow with fg_smc_normalize
    NEED_ENGINE(e);
    if (!h_log_w) return FG_E_BAD_ARG;
    SmcWs W;
    if (int rc = smc_pop(e, W, true)) return rc;
    HIPCHK(hipMemcpyAsync(W.d_lw, h_log_w, (size_t)e->C * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return FG_OK;
}
// resample_particles (smc.rs:326-349): ancestors by `method` from the current weights, particles cloned, weight = 1/N and
// log_weight = ln(1/N).  `step` selects the random sub-stream (as fg_smc_run's tempering step does).
int fg_smc_resample(fg_engine *e, int method, uint32_t step, int64_t *h_indices) {
    NEED_ENGINE(e);
    if (method < 0 || method > 2) return FG_E_BAD_ARG;
    SmcWs W;
    if (int rc = smc_pop(e, W, true)) return rc;
    const long long N = e->C;
    const int S = e->S, TB = 256, NB = (int)((N + TB - 1) / TB);
    hipStream_t s = e->stream;
    double U = 0.0;
    if (method == FG_RESAMPLE_SYSTEMATIC) { FgStream rs = fg_stream(e->seed, 0, step, FG_RNG_SMC_RESAMPLE); U = fg_rng_u01(rs); }
    if (int rc = W.SC.indices(s, W.d_w, N, method, U, nullptr, e->seed, step, W.d_idx)) return rc;
    hipLaunchKernelGGL(k_smc_gather, dim3(NB), dim3(TB), 0, s, (const long long *)e->d_values, W.d_vals2, (const double *)W.M.ll, W.d_ll2,
                       (const double *)W.M.lprior, W.d_lp2, (const long long *)W.d_idx, S, N);
    HIPCHK(hipMemcpyAsync(e->d_values, W.d_vals2, (size_t)S * N * 8, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(W.M.ll, W.d_ll2, (size_t)N * 8, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(W.M.lprior, W.d_lp2, (size_t)N * 8, hipMemcpyDeviceToDevice, s));
    const double uw = 1.0 / (double)N;
    hipLaunchKernelGGL(k_fill, dim3(NB), dim3(TB), 0, s, W.d_w, N, uw);
    hipLaunchKernelGGL(k_fill, dim3(NB), dim3(TB), 0, s, W.d_lw, N, std::log(uw));
    HIPCHK(hipGetLastError());
    if (h_indices) HIPCHK(hipMemcpyAsync(h_indices, W.d_idx, (size_t)N * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return FG_OK;
}
// rejuvenate_particles (smc.rs:698-713): `steps` pi_beta-invariant single-site MH moves per particle with a fresh
// DiminishingAdaptation (batched per sweep, as in fg_smc_run); weights are NOT touched (FG-13).
int fg_smc_rejuvenate(fg_engine *e, double beta, int steps, uint32_t first_move_id, double *h_accept_rate) {
    NEED_ENGINE(e);
    if (steps < 0) return FG_E_BAD_ARG;
    SmcWs W;
    if (int rc = smc_pop(e, W, true)) return rc;
    const long long N = e->C;
    const int S = e->S, TB = 256;
    const size_t Sn = (size_t)std::max(1, S);
    hipStream_t s = e->stream;
    if (e->d == 0 || steps == 0) { if (h_accept_rate) *h_accept_rate = 0.0; return FG_OK; }
    FgSmcScalars h; std::memset(&h, 0, sizeof(h)); h.one = 1.0; h.beta = beta;
    HIPCHK(hipMemcpyAsync(W.st, &h, sizeof(h), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(W.base + W.o_ls, 0, W.o_st - W.o_ls, s));            // log_scale, acc, tot = 0; scale = 1
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((Sn + TB - 1) / TB)), dim3(TB), 0, s, W.M.scale, (long long)Sn, 1.0);
    const int score = !e->P.sstream ? -1 : (e->P.sstream_kinds == 0 ? 0 : (e->P.sstream_gen ? 2 : 3));
    const bool big = e->lds_score > 150 * 1024 || S > FG_SMC_HIST;            // (as in fg_smc_run)
    if (big) { if (int rc_ = smc_global_tile(e)) return rc_; }
    const int wpb = big ? 1 : (int)std::max<size_t>(1, std::min<size_t>((size_t)FG_SMC_WPB(score), (150 * 1024) / std::max<size_t>(1, e->lds_score)));
    const size_t lds_r = big ? 0 : e->lds_score * wpb;
    const unsigned nblk = (unsigned)((N + (long long)e->tw * wpb - 1) / ((long long)e->tw * wpb));
    for (int r = 0; r < steps; ++r) {
        const uint32_t mv = first_move_id + (uint32_t)r;
#define SMC_REJUV(SC_) do { if (big) hipLaunchKernelGGL((k_smc_rejuv<SC_, true>), dim3(nblk), dim3(FG_WAVE), 0, s, e->P, e->X, W.M, (const FgSmcScalars *)W.st, mv, (const long long *)nullptr, (double *)nullptr); \
                            else { if (int rc_ = set_lds(k_smc_rejuv<SC_>, std::max<size_t>(lds_r, 64 * 1024 + 1))) return rc_; \
                                   hipLaunchKernelGGL(k_smc_rejuv<SC_>, dim3(nblk), dim3(e->tw * wpb), lds_r, s, e->P, e->X, W.M, (const FgSmcScalars *)W.st, mv, (const long long *)nullptr, (double *)nullptr); } } while (0)
        unsigned nb_adapt = nblk;
        if (score != 0 && !big && fg_smc_jit_rejuv_launch(e, W.M, (const FgSmcScalars *)W.st, mv, &nb_adapt) == FG_OK) { }      // (as in fg_smc_run)
        else if (score == 0) SMC_REJUV(0); else if (score == 3) SMC_REJUV(3); else if (score == 2) SMC_REJUV(2);
        else SMC_REJUV(-1);
#undef SMC_REJUV
        hipLaunchKernelGGL(k_smc_adapt, dim3((unsigned)S), dim3(256), 0, s, W.M, S, (int)nb_adapt);
    }
    HIPCHK(hipGetLastError());
    if (h_accept_rate) {
        std::vector<long long> acc(Sn), tot(Sn);
        HIPCHK(hipMemcpyAsync(acc.data(), W.M.acc, Sn * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(tot.data(), W.M.tot, Sn * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        long long a = 0, t = 0;
        for (size_t j = 0; j < Sn; j++) { a += acc[j]; t += tot[j]; }
        *h_accept_rate = t > 0 ? (double)a / (double)t : 0.0;
    }
    HIPCHK(hipStreamSynchronize(s));
    return FG_OK;
}

}  // extern "C"
