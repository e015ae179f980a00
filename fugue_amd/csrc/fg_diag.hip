// fg_diag.hip -- cross-chain diagnostics behind the C ABI: split R-hat (src/inference/diagnostics.rs:240-304), multi-chain ESS
// (src/inference/mcmc_utils.rs:231-339), pooled mean / std (diagnostics.rs:331-352).
//
// Draws stay on the GPU that produced them, laid out [n][d][C] (chain fastest): a thread owns one (coordinate, chain)
// column and walks its n draws with stride d*C -- coalesced across the 64 lanes of a wave.  Per chain only moments leave
// the kernel ([d][6][C]); per lag only the sum over chains of the lag's autocovariance ([d][lags]).  With several GPUs
// (one process per GPU, chains sharded) the ONLY exchange of the whole engine happens here, inside the library, over
// RCCL: an all-gather of the per-chain moments and an all-reduce of the pooled lag sums; the Geyer / R-hat combination
// is C++ (fg_diag_host.cpp).  RCCL is bound at run time (dlopen): the copy that sits next to the HIP runtime this library runs on.
#include "fg_engine_internal.h"

#include <dlfcn.h>

// moments [d][6][C]: full-chain mean, sum of squared deviations; then the same for the first and second half (half = n/2,
// the middle draw dropped when n is odd: split_f64_chains :240-253).  Two sweeps over the column: the three sums (each
// the in-order sum the reference forms, values.iter().sum() :275-278), then the three sums of squared deviations (:292-296).
__global__ void k_diag_moments(const double *draws, int n, int d, long long C, double *out) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (c >= C) return;
    const double *x = draws + (long long)i * C + c;
    const long long st = (long long)d * C;
    const int half = n / 2;
    double s_full = 0.0, s_h1 = 0.0, s_h2 = 0.0;
    for (int t = 0; t < n; ++t) {
        const double v = x[t * st];
        s_full += v;
        if (t == half - 1) s_h1 = s_full;                      // the in-order sum of the first half IS the running sum there
        if (t >= half && t < 2 * half) s_h2 += v;
    }
    const double m_full = n > 0 ? s_full / (double)n : NAN, m_h1 = half > 0 ? s_h1 / (double)half : NAN, m_h2 = half > 0 ? s_h2 / (double)half : NAN;
    double q_full = 0.0, q_h1 = 0.0, q_h2 = 0.0;
    for (int t = 0; t < n; ++t) {
        const double v = x[t * st];
        const double a = v - m_full; q_full += a * a;
        if (t < half) { const double b = v - m_h1; q_h1 += b * b; }
        else if (t < 2 * half) { const double b = v - m_h2; q_h2 += b * b; }
    }
    double *o = out + (long long)i * 6 * C + c;
    o[0] = m_full; o[C] = q_full; o[2 * C] = m_h1; o[3 * C] = q_h1; o[4 * C] = m_h2; o[5 * C] = q_h2;
}

// Sum over chains of the biased autocovariances acov_t = (1/n) sum_i c_i c_{i+t} (autocovariances, mcmc_utils.rs:231-244) for
// the FG_ACOV_LAGS lags [lag0, lag0 + FG_ACOV_LAGS): ONE sweep over the column with the last FG_ACOV_LAGS centred values of
// the trailing stream in registers (each draw is read twice per chunk, not twice per lag); per chain the products are
// added in the reference's order (ascending i).  Block partials go to `partial`, k_diag_acov_finish adds them in block
// order: the result does not depend on scheduling.
#define FG_ACOV_LAGS 32
__global__ __launch_bounds__(256) void k_diag_autocov(const double *draws, int n, int d, long long C, const double *moments, int lag0,
                                                       double *partial /*[d][FG_ACOV_LAGS][gridDim.x]*/) {
    __shared__ double sh[4][FG_ACOV_LAGS];
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    double acc[FG_ACOV_LAGS], win[FG_ACOV_LAGS];
#pragma unroll
    for (int k = 0; k < FG_ACOV_LAGS; ++k) { acc[k] = 0.0; win[k] = 0.0; }
    if (c < C) {
        const double *x = draws + (long long)i * C + c;
        const long long st = (long long)d * C;
        const double mean = moments[((long long)i * 6) * C + c];
        // u = the later index of a product c_{u - lag} c_u; win[k] = c_{u - lag0 - k} (0 before the chain starts)
        for (int u = lag0; u < n; ++u) {
            const double cur = x[u * st] - mean;
#pragma unroll
            for (int k = FG_ACOV_LAGS - 1; k > 0; --k) win[k] = win[k - 1];
            win[0] = x[(u - lag0) * st] - mean;
#pragma unroll
            for (int k = 0; k < FG_ACOV_LAGS; ++k) acc[k] += win[k] * cur;      // win[k] = 0 before the chain starts: adds +-0
        }
#pragma unroll
        for (int k = 0; k < FG_ACOV_LAGS; ++k) acc[k] /= (double)n;
    }
#pragma unroll
    for (int k = 0; k < FG_ACOV_LAGS; ++k) {
        double s = acc[k];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < FG_ACOV_LAGS) {
        const int k = threadIdx.x;
        partial[((long long)i * FG_ACOV_LAGS + k) * gridDim.x + blockIdx.x] = sh[0][k] + sh[1][k] + sh[2][k] + sh[3][k];
    }
}
__global__ void k_diag_acov_finish(const double *partial, int nblk, int n_lags, int d, double *out /*[d][n_lags]*/) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d * n_lags) return;
    const int i = j / n_lags, k = j % n_lags;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partial[((long long)i * FG_ACOV_LAGS + k) * nblk + b];
    out[j] = s;
}

// Sums over this engine's chains, for the all-reduce exchange (fg_diag_combine_reduced): block partials in a fixed tree, added in
// block order by k_diag_acov_finish -- the result does not depend on scheduling.
//   mode 0: row r = 6 i + k of the moments: sum_c mom[i][k][c]                                            (rows = 6 d)
//   mode 1: row r = 2 i + k: k = 0: sum_c (mean_c - ov[i][0])^2; k = 1: sum_c (mean_h1_c - ov[i][1])^2 + (mean_h2_c - ov[i][1])^2   (rows = 2 d)
__global__ __launch_bounds__(256) void k_diag_chain_sums(const double *mom, long long C, int mode, const double *ov, double *partial /*[rows][gridDim.x]*/) {
    __shared__ double sh[4];
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    double v = 0.0;
    if (c < C) {
        if (mode == 0) v = mom[(long long)r * C + c];
        else {
            const int i = r >> 1;
            const double *m6 = mom + (long long)i * 6 * C + c;
            if (!(r & 1)) { const double a = m6[0] - ov[2 * i]; v = a * a; }
            else { const double a = m6[2 * C] - ov[2 * i + 1], b = m6[4 * C] - ov[2 * i + 1]; v = a * a + b * b; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[(long long)r * gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void k_diag_sum_partials(const double *partial, int nblk, int rows, double *out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partial[(long long)r * nblk + b];
    out[r] = s;
}

// geweke_diagnostic (mcmc_utils.rs:354-421) of every (coordinate, chain) column: z = (mean of the first 10 % - mean of the
// last 50 %) / sqrt of the two segments' spectral variances of the mean (spectral_variance_of_mean :392-421: s^2 tau / n with
// tau summed over the initial positive autocorrelations).  One thread per column; the lag loop stops per chain.
__device__ double fg_spectral_var_of_mean(const double *x, long long st, int a, int b) {
    const int k = b - a;
    if (k < 2) return 0.0;
    double s = 0.0;
    for (int t = a; t < b; ++t) s += x[t * st];
    const double mean = s / (double)k;
    double q = 0.0;
    for (int t = a; t < b; ++t) { const double dv = x[t * st] - mean; q += dv * dv; }
    const double s2 = q / ((double)k - 1.0);
    if (s2 == 0.0) return 0.0;
    const int max_lag = k - 1 < 1024 ? k - 1 : 1024;
    const double var0 = q / (double)k;                         // acov[0]: the same sum of squares / n (mcmc_utils.rs:231-244)
    if (var0 <= 0.0) return 0.0;
    double tau = 1.0;
    for (int lag = 1; lag <= max_lag; ++lag) {
        double c = 0.0;
        for (int t = a; t + lag < b; ++t) c += (x[t * st] - mean) * (x[(t + lag) * st] - mean);
        const double rho = (c / (double)k) / var0;
        if (rho <= 0.0) break;
        tau += 2.0 * rho;
    }
    return s2 * tau / (double)k;
}
__global__ void k_diag_geweke(const double *draws, int n, int d, long long C, double *out /*[d][C]*/) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (c >= C) return;
    const double *x = draws + (long long)i * C + c;
    const long long st = (long long)d * C;
    double z = NAN;
    const int first_end = n / 10, last_start = n / 2;
    if (n >= 20 && first_end >= 2 && n - last_start >= 2) {
        double s1 = 0.0, s2 = 0.0;
        for (int t = 0; t < first_end; ++t) s1 += x[t * st];
        for (int t = last_start; t < n; ++t) s2 += x[t * st];
        const double mean1 = s1 / (double)first_end, mean2 = s2 / (double)(n - last_start);
        const double se = sqrt(fg_spectral_var_of_mean(x, st, 0, first_end) + fg_spectral_var_of_mean(x, st, last_start, n));
        z = se == 0.0 ? 0.0 : (mean1 - mean2) / se;
    }
    out[(long long)i * C + c] = z;
}

// ---- quantiles of summarize_f64_parameter (diagnostics.rs:355-371): sorted[round((len - 1) p)] over ALL draws of a coordinate, by
// radix select on the order-preserving 64-bit key of a double, eight bits per pass: every pass histograms the next digit of the
// elements that still match each wanted rank's prefix (block histograms in LDS, then 64-bit global counters); the host -- after an
// all-reduce of the counters when the chains are sharded over ranks -- walks the digits to the one that holds the rank.  Eight
// streaming passes over the draws, no sort, nothing proportional to the draw count leaves the GPU.
#define FG_Q_MAX 8
__device__ __forceinline__ unsigned long long fg_sort_key(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);          // ascending in the double's order (-0.0 just below +0.0, NaN above +inf)
}
__global__ __launch_bounds__(256) void k_diag_qhist(const double *draws, int n, int d, long long C, int pass, int n_q, const unsigned long long *prefix /*[d][n_q]*/,
                                                    unsigned long long *hist /*[d][n_q][256]*/) {
    __shared__ unsigned int sh[FG_Q_MAX * 256];
    const int i = blockIdx.y;
    for (int k = threadIdx.x; k < n_q * 256; k += blockDim.x) sh[k] = 0u;
    __syncthreads();
    unsigned long long pf[FG_Q_MAX];
    for (int q = 0; q < n_q; ++q) pf[q] = prefix[(long long)i * n_q + q];
    const int hi = 64 - 8 * pass;                                  // bits above the digit of this pass are decided
    const long long total = (long long)n * C;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long t = e / C, c = e - t * C;
        const unsigned long long key = fg_sort_key(draws[(t * d + i) * C + c]);
        const unsigned int digit = (unsigned int)(key >> (hi - 8)) & 255u;
        for (int q = 0; q < n_q; ++q)
            if (pass == 0 || (key >> hi) == (pf[q] >> hi)) atomicAdd(&sh[q * 256 + digit], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n_q * 256; k += blockDim.x)
        if (sh[k]) atomicAdd(&hist[(long long)i * n_q * 256 + k], (unsigned long long)sh[k]);
}

// ---- RCCL, bound at run time ------------------------------------------------------------------------------------------
struct FgUniqueId { char b[128]; };
namespace {
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, struct FgUniqueId, int) = nullptr;   // ncclUniqueId is passed BY VALUE: a 128-byte struct
    int (*CommDestroy)(void *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    int (*CommUserRank)(void *, int *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl *rccl() {
    static Rccl R;
    static bool tried = false;
    if (tried) return R.h ? &R : nullptr;
    tried = true;
    // The RCCL that belongs to the HIP runtime THIS library is bound to: a process may hold two ROCm stacks (the system's and
    // the one bundled with PyTorch), and a communicator must run on the runtime that owns the engine's device state.  So:
    // the librccl next to the loaded libamdhip64 first, then the usual names.
    std::vector<std::string> names;
    Dl_info info;
    if (dladdr((void *)(hipError_t (*)(void **, size_t))&hipMalloc, &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t sl = dir.rfind('/');
        if (sl != std::string::npos) { dir.resize(sl); names.push_back(dir + "/librccl.so.1"); names.push_back(dir + "/librccl.so"); }
    }
    for (const char *nm : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) names.push_back(nm);
    for (const std::string &nm : names) { R.h = dlopen(nm.c_str(), RTLD_NOW | RTLD_LOCAL); if (R.h) break; }
    if (!R.h) return nullptr;
    *(void **)&R.GetUniqueId = dlsym(R.h, "ncclGetUniqueId");
    *(void **)&R.CommInitRank = dlsym(R.h, "ncclCommInitRank");
    *(void **)&R.CommDestroy = dlsym(R.h, "ncclCommDestroy");
    *(void **)&R.CommCount = dlsym(R.h, "ncclCommCount");
    *(void **)&R.CommUserRank = dlsym(R.h, "ncclCommUserRank");
    *(void **)&R.AllGather = dlsym(R.h, "ncclAllGather");
    *(void **)&R.AllReduce = dlsym(R.h, "ncclAllReduce");
    *(void **)&R.GetErrorString = dlsym(R.h, "ncclGetErrorString");
    if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.CommCount || !R.AllGather || !R.AllReduce) { R.h = nullptr; return nullptr; }
    return &R;
}
int rccl_fail(Rccl *R, const char *what, int rc) {
    fg_set_error(std::string(what) + ": " + ((R && R->GetErrorString) ? R->GetErrorString(rc) : "RCCL error"));
    return FG_E_HIP;
}
const int kNcclFloat64 = 8, kNcclUint64 = 5, kNcclSum = 0;

struct AcovCtx { fg_engine *e; const double *d_draws; int n, d; const double *d_mom; void *comm; double *d_small; double *d_part; long long bytes; };
}  // namespace

extern "C" {

typedef int (*fg_acov_fn)(void *user, int lag0, int n_lags, double *h_sums);
typedef int (*fg_reduce_fn)(void *user, int stage, const double *h_in, double *h_out);
int fg_diag_combine_reduced(int64_t m, int n, int d, fg_reduce_fn reduce, fg_acov_fn acov, void *user, double *h_rhat, double *h_ess, double *h_mean, double *h_std);
int fg_diag_combine(const double *h_moments, int64_t m, int n, int d, fg_acov_fn acov, void *user, double *h_rhat, double *h_ess, double *h_mean, double *h_std);

int fg_diag_chain_moments(fg_engine *e, const double *d_draws, int n, int d, double *d_moments) {
    NEED_ENGINE(e);
    if (!d_draws || !d_moments || n <= 0 || d <= 0) return FG_E_BAD_ARG;
    hipLaunchKernelGGL(k_diag_moments, dim3((unsigned)((e->C + 255) / 256), (unsigned)d), dim3(256), 0, e->stream, d_draws, n, d, e->C, d_moments);
    HIPCHK(hipGetLastError());
    return FG_OK;
}

int fg_diag_geweke(fg_engine *e, const double *d_draws, int n, int d, double *d_z) {
    NEED_ENGINE(e);
    if (!d_draws || !d_z || n <= 0 || d <= 0) return FG_E_BAD_ARG;
    hipLaunchKernelGGL(k_diag_geweke, dim3((unsigned)((e->C + 255) / 256), (unsigned)d), dim3(256), 0, e->stream, d_draws, n, d, e->C, d_z);
    HIPCHK(hipGetLastError());
    return FG_OK;
}

// device result: d_sums [d][n_lags] (this engine's chains only)
static int acov_sums_device(fg_engine *e, const double *d_draws, int n, int d, const double *d_moments, int lag0, int n_lags, double *d_sums) {
    if (n_lags > FG_ACOV_LAGS) { fg_set_error("fg_diag_autocov_sums: at most 32 lags per call"); return FG_E_BAD_ARG; }
    const unsigned nb = (unsigned)((e->C + 255) / 256);
    double *d_part = nullptr;
    int rc = dev_alloc(&d_part, (size_t)d * FG_ACOV_LAGS * nb);
    if (rc) return rc;
    hipLaunchKernelGGL(k_diag_autocov, dim3(nb, (unsigned)d), dim3(256), 0, e->stream, d_draws, n, d, e->C, d_moments, lag0, d_part);
    hipLaunchKernelGGL(k_diag_acov_finish, dim3((unsigned)((d * n_lags + 127) / 128)), dim3(128), 0, e->stream, (const double *)d_part, (int)nb, n_lags, d, d_sums);
    hipError_t he = hipGetLastError();
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    (void)hipFree(d_part);
    if (he != hipSuccess) { fg_set_error(hipGetErrorString(he)); return FG_E_HIP; }
    return FG_OK;
}

int fg_diag_autocov_sums(fg_engine *e, const double *d_draws, int n, int d, const double *d_moments, int lag0, int n_lags, double *h_sums) {
    NEED_ENGINE(e);
    if (!d_draws || !d_moments || !h_sums || n <= 0 || d <= 0 || lag0 < 0 || n_lags <= 0) return FG_E_BAD_ARG;
    std::vector<double> out((size_t)d * n_lags, 0.0);
    double *d_sums = nullptr;
    int rc = dev_alloc(&d_sums, (size_t)d * FG_ACOV_LAGS);
    if (rc) return rc;
    for (int l0 = 0; l0 < n_lags && !rc; l0 += FG_ACOV_LAGS) {       // chunks of 32 lags
        const int nl = std::min(FG_ACOV_LAGS, n_lags - l0);
        rc = acov_sums_device(e, d_draws, n, d, d_moments, lag0 + l0, nl, d_sums);
        std::vector<double> tmp((size_t)d * nl);
        if (!rc && hipMemcpy(tmp.data(), d_sums, tmp.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = FG_E_HIP;
        for (int i = 0; i < d && !rc; ++i) for (int k = 0; k < nl; ++k) out[(size_t)i * n_lags + l0 + k] = tmp[(size_t)i * nl + k];
    }
    (void)hipFree(d_sums);
    if (rc) return rc;
    std::memcpy(h_sums, out.data(), out.size() * 8);
    return FG_OK;
}

// ---- communicator helpers: the ranks of one run share an RCCL communicator created from a 128-byte unique id that rank 0
// obtains and the host distributes (any transport: a file, MPI, torch.distributed's store)
int fg_comm_unique_id(void *out_128_bytes) {
    Rccl *R = rccl();
    if (!R) { fg_set_error("RCCL is not available (librccl.so not found)"); return FG_E_UNSUPPORTED; }
    if (!out_128_bytes) return FG_E_BAD_ARG;
    const int rc = R->GetUniqueId(out_128_bytes);
    return rc ? rccl_fail(R, "ncclGetUniqueId", rc) : FG_OK;
}
int fg_comm_init(fg_engine *e, int world_size, int rank, const void *unique_id_128_bytes, void **out_comm) {
    NEED_ENGINE(e);
    Rccl *R = rccl();
    if (!R) { fg_set_error("RCCL is not available (librccl.so not found)"); return FG_E_UNSUPPORTED; }
    if (!unique_id_128_bytes || !out_comm || world_size < 1 || rank < 0 || rank >= world_size) return FG_E_BAD_ARG;
    FgUniqueId id; std::memcpy(id.b, unique_id_128_bytes, 128);
    const int rc = R->CommInitRank(out_comm, world_size, id, rank);
    return rc ? rccl_fail(R, "ncclCommInitRank", rc) : FG_OK;
}
int fg_comm_destroy(void *comm) {
    Rccl *R = rccl();
    if (!R || !comm) return FG_E_BAD_ARG;
    const int rc = R->CommDestroy(comm);
    return rc ? rccl_fail(R, "ncclCommDestroy", rc) : FG_OK;
}

static int acov_cb(void *user, int lag0, int n_lags, double *h_sums) {
    AcovCtx *A = (AcovCtx *)user;
    fg_engine *e = A->e;
    double *d_sums = nullptr;
    int rc = dev_alloc(&d_sums, (size_t)A->d * FG_ACOV_LAGS);
    if (rc) return rc;
    rc = acov_sums_device(e, A->d_draws, A->n, A->d, A->d_mom, lag0, n_lags, d_sums);
    if (!rc && A->comm) {                                          // pooled over every rank's chains: all-reduce over RCCL / xGMI
        Rccl *R = rccl();
        const int nr = R->AllReduce(d_sums, d_sums, (size_t)A->d * n_lags, kNcclFloat64, kNcclSum, A->comm, e->stream);
        if (nr) rc = rccl_fail(R, "ncclAllReduce", nr);
        A->bytes += (long long)A->d * n_lags * 8;
    }
    if (!rc) {
        hipError_t he = hipMemcpyAsync(h_sums, d_sums, (size_t)A->d * n_lags * 8, hipMemcpyDeviceToHost, e->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
        if (he != hipSuccess) { fg_set_error(hipGetErrorString(he)); rc = FG_E_HIP; }
    }
    (void)hipFree(d_sums);
    return rc;
}

// The chain sums of fg_diag_combine_reduced: this engine's chains on the device, then -- with a communicator -- ONE all-reduce of
// 6 d (stage 1) or 2 d (stage 2) doubles over RCCL / xGMI.
static int reduce_cb(void *user, int stage, const double *h_in, double *h_out) {
    AcovCtx *A = (AcovCtx *)user;
    fg_engine *e = A->e;
    const int rows = stage == 1 ? 6 * A->d : 2 * A->d;
    const unsigned nb = (unsigned)((e->C + 255) / 256);
    hipError_t he = hipSuccess;
    double *d_ov = A->d_small + 8 * (size_t)A->d;                   // [2 d] overall means (stage 2 input)
    if (stage == 2) he = hipMemcpyAsync(d_ov, h_in, (size_t)2 * A->d * 8, hipMemcpyHostToDevice, e->stream);
    if (he != hipSuccess) { fg_set_error(hipGetErrorString(he)); return FG_E_HIP; }
    hipLaunchKernelGGL(k_diag_chain_sums, dim3(nb, (unsigned)rows), dim3(256), 0, e->stream, A->d_mom, e->C, stage == 1 ? 0 : 1, (const double *)d_ov, A->d_part);
    hipLaunchKernelGGL(k_diag_sum_partials, dim3((unsigned)((rows + 127) / 128)), dim3(128), 0, e->stream, (const double *)A->d_part, (int)nb, rows, A->d_small);
    he = hipGetLastError();
    if (he != hipSuccess) { fg_set_error(hipGetErrorString(he)); return FG_E_HIP; }
    if (A->comm) {
        Rccl *R = rccl();
        const int nr = R->AllReduce(A->d_small, A->d_small, (size_t)rows, kNcclFloat64, kNcclSum, A->comm, e->stream);
        if (nr) return rccl_fail(R, "ncclAllReduce", nr);
        A->bytes += (long long)rows * 8;
    }
    he = hipMemcpyAsync(h_out, A->d_small, (size_t)rows * 8, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (he != hipSuccess) { fg_set_error(hipGetErrorString(he)); return FG_E_HIP; }
    return FG_OK;
}

int fg_diag_quantiles(fg_engine *e, const double *d_draws, int n, int d, void *comm, const double *h_probs, int n_probs, double *h_out) {
    NEED_ENGINE(e);
    if (!d_draws || !h_probs || !h_out || n <= 0 || d <= 0 || n_probs < 1 || n_probs > FG_Q_MAX) return FG_E_BAD_ARG;
    Rccl *R = comm ? rccl() : nullptr;
    if (comm && !R) { fg_set_error("RCCL is not available (librccl.so not found)"); return FG_E_UNSUPPORTED; }
    int world = 1;
    if (comm) { const int rc = R->CommCount(comm, &world); if (rc) return rccl_fail(R, "ncclCommCount", rc); }
    const unsigned long long len = (unsigned long long)world * (unsigned long long)e->C * (unsigned long long)n;
    std::vector<unsigned long long> rank((size_t)d * n_probs), prefix((size_t)d * n_probs, 0ull);
    for (int q = 0; q < n_probs; ++q) {
        if (!(h_probs[q] >= 0.0 && h_probs[q] <= 1.0)) { fg_set_error("fg_diag_quantiles: probabilities must lie in [0, 1]"); return FG_E_BAD_ARG; }
        const unsigned long long idx = (unsigned long long)std::round((double)(len - 1) * h_probs[q]);     // f64::round: half away from zero
        for (int i = 0; i < d; ++i) rank[(size_t)i * n_probs + q] = idx;
    }
    unsigned long long *d_prefix = nullptr, *d_hist = nullptr;
    const size_t nh = (size_t)d * n_probs * 256;
    int rc = dev_alloc(&d_prefix, (size_t)d * n_probs);
    if (!rc) rc = dev_alloc(&d_hist, nh);
    std::vector<unsigned long long> hist(nh);
    const long long total = (long long)n * e->C;
    const unsigned nb = (unsigned)std::min<long long>((total + 255) / 256, 4096);
    for (int pass = 0; pass < 8 && !rc; ++pass) {
        hipError_t he = hipMemcpyAsync(d_prefix, prefix.data(), prefix.size() * 8, hipMemcpyHostToDevice, e->stream);
        if (he == hipSuccess) he = hipMemsetAsync(d_hist, 0, nh * 8, e->stream);
        if (he == hipSuccess) {
            hipLaunchKernelGGL(k_diag_qhist, dim3(nb, (unsigned)d), dim3(256), 0, e->stream, d_draws, n, d, e->C, pass, n_probs, (const unsigned long long *)d_prefix, d_hist);
            he = hipGetLastError();
        }
        if (he == hipSuccess && comm) { const int nr = R->AllReduce(d_hist, d_hist, nh, kNcclUint64, kNcclSum, comm, e->stream); if (nr) { rc = rccl_fail(R, "ncclAllReduce", nr); break; } }
        if (he == hipSuccess) he = hipMemcpyAsync(hist.data(), d_hist, nh * 8, hipMemcpyDeviceToHost, e->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
        if (he != hipSuccess) { fg_set_error(hipGetErrorString(he)); rc = FG_E_HIP; break; }
        const int lo = 56 - 8 * pass;
        for (size_t k = 0; k < (size_t)d * n_probs; ++k) {            // the digit whose cumulative count first exceeds the rank
            const unsigned long long *hh = &hist[k * 256];
            unsigned long long cum = 0; int dg = 0;
            for (; dg < 255; ++dg) { if (cum + hh[dg] > rank[k]) break; cum += hh[dg]; }
            rank[k] -= cum;
            prefix[k] |= (unsigned long long)dg << lo;
        }
    }
    if (d_prefix) (void)hipFree(d_prefix);
    if (d_hist) (void)hipFree(d_hist);
    if (rc) return rc;
    for (size_t k = 0; k < (size_t)d * n_probs; ++k) {
        const unsigned long long key = prefix[k];
        const unsigned long long u = (key >> 63) ? (key & 0x7fffffffffffffffull) : ~key;
        std::memcpy(&h_out[k], &u, 8);
    }
    return FG_OK;
}

int fg_diag_set_exchange(fg_engine *e, int mode) {
    NEED_ENGINE(e);
    if (mode != FG_DIAG_REDUCE && mode != FG_DIAG_GATHER) return FG_E_BAD_ARG;
    e->diag_mode = mode;
    return FG_OK;
}
int64_t fg_diag_exchange_bytes(const fg_engine *e) { return e ? (int64_t)e->diag_bytes : 0; }

// r_hat_f64 + effective_sample_size_multichain + summarize_f64_parameter for every coordinate of d_draws [n][d][C] over the
// chains of EVERY rank of `comm` (NULL: this engine's chains).  All ranks call it with equal n, d and C and get the same
// numbers.  h_* [d], any may be NULL.
int fg_diag_rhat_ess(fg_engine *e, const double *d_draws, int n, int d, void *comm, double *h_rhat, double *h_ess, double *h_mean, double *h_std,
                     int64_t *out_total_chains) {
    NEED_ENGINE(e);
    if (!d_draws || n <= 0 || d <= 0) return FG_E_BAD_ARG;
    Rccl *R = comm ? rccl() : nullptr;
    if (comm && !R) { fg_set_error("RCCL is not available (librccl.so not found)"); return FG_E_UNSUPPORTED; }
    int world = 1;
    if (comm) { const int rc = R->CommCount(comm, &world); if (rc) return rccl_fail(R, "ncclCommCount", rc); }
    const size_t per = (size_t)d * 6 * e->C;
    double *d_mom = nullptr, *d_all = nullptr, *d_small = nullptr, *d_part = nullptr;
    int rc = dev_alloc(&d_mom, per);
    if (rc) return rc;
    rc = fg_diag_chain_moments(e, d_draws, n, d, d_mom);
    std::vector<double> mom;
    const int64_t m = (int64_t)world * e->C;
    e->diag_bytes = 0;
    if (!rc && e->diag_mode == FG_DIAG_REDUCE) {
        // the default exchange: chains enter R-hat, the pooled moments and the ESS only through sums over chains -- all-reduces of
        // 6 d, 2 d and 32 d (per lag chunk) doubles, nothing proportional to the chain count leaves the GPU
        rc = dev_alloc(&d_small, (size_t)10 * d);
        if (!rc) rc = dev_alloc(&d_part, (size_t)6 * d * ((e->C + 255) / 256));
        if (!rc) {
            AcovCtx A{ e, d_draws, n, d, d_mom, comm, d_small, d_part, 0 };
            rc = fg_diag_combine_reduced(m, n, d, reduce_cb, acov_cb, &A, h_rhat, h_ess, h_mean, h_std);
            e->diag_bytes = A.bytes;
        }
        if (d_small) (void)hipFree(d_small);
        if (d_part) (void)hipFree(d_part);
        if (out_total_chains) *out_total_chains = m;
        (void)hipFree(d_mom);
        return rc;
    }
    if (!rc && comm) {                                              // FG_DIAG_GATHER: every rank gets every chain's moments (all-gather over RCCL / xGMI)
        rc = dev_alloc(&d_all, per * world);
        if (!rc) { const int nr = R->AllGather(d_mom, d_all, per, kNcclFloat64, comm, e->stream); if (nr) rc = rccl_fail(R, "ncclAllGather", nr); }
        e->diag_bytes += (long long)per * 8;
    }
    if (!rc) {
        std::vector<double> raw(per * world);
        hipError_t he = hipMemcpyAsync(raw.data(), comm ? d_all : d_mom, raw.size() * 8, hipMemcpyDeviceToHost, e->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
        if (he != hipSuccess) { fg_set_error(hipGetErrorString(he)); rc = FG_E_HIP; }
        else if (world == 1) mom.swap(raw);
        else {                                                      // [rank][d][6][C] -> [d][6][rank * C + c]: global chain order
            mom.resize(per * world);
            for (int r = 0; r < world; ++r)
                for (int i = 0; i < d * 6; ++i)
                    std::memcpy(&mom[((size_t)i * world + r) * e->C], &raw[((size_t)r * d * 6 + i) * e->C], (size_t)e->C * 8);
        }
    }
    if (!rc) {
        AcovCtx A{ e, d_draws, n, d, d_mom, comm, nullptr, nullptr, 0 };
        rc = fg_diag_combine(mom.data(), m, n, d, acov_cb, &A, h_rhat, h_ess, h_mean, h_std);
        e->diag_bytes += A.bytes;
    }
    if (out_total_chains) *out_total_chains = m;
    (void)hipFree(d_mom);
    if (d_all) (void)hipFree(d_all);
    return rc;
}

}  // extern "C"
