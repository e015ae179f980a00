// fg_diag.hip -- per-chain statistics for the cross-chain diagnostics (split R-hat:
// src/inference/diagnostics.rs:240-304; multi-chain ESS: src/inference/mcmc_utils.rs:231-339).
//
// Draws stay on the GPU that produced them, laid out [n][d][C] (chain fastest): each thread owns
// one (coordinate, chain) column and walks the n draws with stride d*C -- coalesced across the 64
// lanes of a wave.  Only O(d*C) moments (for R-hat) and O(d*lags) pooled autocovariance sums
// (for ESS) leave the GPU; with several GPUs those are what the RCCL all-gather / all-reduce
// moves (fugue_amd/diagnostics.py).
#include "fg_engine_internal.h"

// moments [d][6][C]: full-chain mean, sum of squared deviations; then the same for the first and
// second half (half = n/2, the middle draw dropped when n is odd: split_f64_chains :240-253).
__global__ void k_diag_moments(const double *draws, int n, int d, long long C, double *out) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (c >= C) return;
    const double *x = draws + (long long)i * C + c;
    const long long st = (long long)d * C;
    const int half = n / 2;
    const int lo[3] = { 0, 0, half }, hi[3] = { n, half, 2 * half };
    for (int part = 0; part < 3; ++part) {
        const int a = lo[part], b = hi[part], len = b - a;
        double s = 0.0;
        for (int t = a; t < b; ++t) s += x[t * st];
        const double mean = len > 0 ? s / (double)len : NAN;            // values.iter().sum() / len  :275-278
        double ssd = 0.0;
        for (int t = a; t < b; ++t) { const double dv = x[t * st] - mean; ssd += dv * dv; }   // :292-296
        out[((long long)i * 6 + 2 * part) * C + c] = mean;
        out[((long long)i * 6 + 2 * part + 1) * C + c] = ssd;
    }
}

// Sum over chains of the biased autocovariances acov_t = (1/n) sum_i c_i c_{i+t}
// (autocovariances, mcmc_utils.rs:231-244) for lags [lag0, lag0 + n_lags).
// grid = (chain blocks, lags, d); per-block partial sums are written out and added on the host in
// block order, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void k_diag_autocov(const double *draws, int n, int d, long long C, const double *moments, int lag0,
                                                       double *partial /*[d][n_lags][gridDim.x]*/) {
    __shared__ double sh[4];
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lag = lag0 + blockIdx.y, i = blockIdx.z;
    double s = 0.0;
    if (c < C && lag < n) {
        const double *x = draws + (long long)i * C + c;
        const long long st = (long long)d * C;
        const double mean = moments[((long long)i * 6) * C + c];
        for (int t = 0; t + lag < n; ++t) s += (x[t * st] - mean) * (x[(t + lag) * st] - mean);
        s /= (double)n;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[((long long)i * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

extern "C" {

int fg_diag_chain_moments(fg_engine *e, const double *d_draws, int n, int d, double *d_moments) {
    NEED_ENGINE(e);
    if (!d_draws || !d_moments || n <= 0 || d <= 0) return FG_E_BAD_ARG;
    hipLaunchKernelGGL(k_diag_moments, dim3((unsigned)((e->C + 255) / 256), (unsigned)d), dim3(256), 0, e->stream, d_draws, n, d, e->C, d_moments);
    HIPCHK(hipGetLastError());
    return FG_OK;
}

int fg_diag_autocov_sums(fg_engine *e, const double *d_draws, int n, int d, const double *d_moments, int lag0, int n_lags, double *h_sums) {
    NEED_ENGINE(e);
    if (!d_draws || !d_moments || !h_sums || n <= 0 || d <= 0 || lag0 < 0 || n_lags <= 0) return FG_E_BAD_ARG;
    const unsigned nb = (unsigned)((e->C + 255) / 256);
    double *d_part = nullptr;
    int rc = dev_alloc(&d_part, (size_t)d * n_lags * nb);
    if (rc) return rc;
    hipLaunchKernelGGL(k_diag_autocov, dim3(nb, (unsigned)n_lags, (unsigned)d), dim3(256), 0, e->stream, d_draws, n, d, e->C, d_moments, lag0, d_part);
    std::vector<double> part((size_t)d * n_lags * nb);
    hipError_t he = hipGetLastError();
    if (he == hipSuccess) he = hipMemcpyAsync(part.data(), d_part, part.size() * 8, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    (void)hipFree(d_part);
    if (he != hipSuccess) { fg_set_error(hipGetErrorString(he)); return FG_E_HIP; }
    for (size_t k = 0; k < (size_t)d * n_lags; ++k) {
        double s = 0.0;
        for (unsigned b = 0; b < nb; ++b) s += part[k * nb + b];
        h_sums[k] = s;
    }
    return FG_OK;
}

}  // extern "C"
