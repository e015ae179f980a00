// Does the core clock hold during a train of SHORT kernels?  N back-to-back launches of a kernel that runs `n` dependent f64
// FMAs per wave on every SIMD (one block per CU x 4 waves); block 0 reads the shader clock (clock64) and the 100 MHz real-time
// counter (wall_clock64) at its start and end.  Reported per train: clock inside a kernel, duration of a kernel by the counters,
// gap between kernels, and the HIP-event time of the train / N.  (An SMC run is ~80 kernels of 5-25 us.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(double *out, long long *t, int n, int slot) {
    double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) a = __builtin_fma(a, b, c);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[4 * slot] = c0; t[4 * slot + 1] = c1; t[4 * slot + 2] = w0; t[4 * slot + 3] = w1; }
}
int main() {
    double *d; long long *t;
    const int N = 200;
    hipMalloc(&d, 256 * 256 * 8); hipMemset(d, 0, 256 * 256 * 8); hipMalloc(&t, N * 32);
    std::vector<long long> h(4 * N);
    for (int n : {50, 200, 1000, 5000, 50000}) {                      // 16 n dependent FMAs at ~6 cycles: 2 us ... 2 ms per kernel
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            for (int s = 0; s < N; ++s) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, t, n, s);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), t, N * 32, hipMemcpyDeviceToHost);
            double mhz = 0, dur = 0, gap = 0;
            for (int s = N / 2; s < N; ++s) {
                mhz += (double)(h[4 * s + 1] - h[4 * s]) / ((double)(h[4 * s + 3] - h[4 * s + 2]) / 100.0);
                dur += (double)(h[4 * s + 3] - h[4 * s + 2]) / 100.0;
                if (s + 1 < N) gap += (double)(h[4 * (s + 1) + 2] - h[4 * s + 3]) / 100.0;
            }
            const int m = N - N / 2;
            if (rep == 1) printf("n=%6d: clock inside a kernel %.0f MHz, kernel %.2f us (by its counters), gap to the next kernel %.2f us, train / N = %.2f us (HIP events)\n", n, mhz / m, dur / m, gap / (m - 1), ms * 1e3 / N);
        }
    }
    return 0;
}
