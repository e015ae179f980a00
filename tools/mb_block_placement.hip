// Where and when do the blocks of a one-block-per-CU grid run?  Every block records its start / end on the 100 MHz real-time
// counter and its hardware id (XCC_ID register, HW_ID: SE / CU / SIMD / wave slot).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void k(double *out, long long *t, int n) {
    double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
    const long long w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) a = __builtin_fma(a, b, c);
    }
    const long long w1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a;
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        t[4 * blockIdx.x] = w0; t[4 * blockIdx.x + 1] = w1; t[4 * blockIdx.x + 2] = hw; t[4 * blockIdx.x + 3] = xcc;
    }
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device: %s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    double *d; long long *t;
    for (int blocks : {256, 512}) for (int threads : {256, 512, 1024}) {
        hipMalloc(&d, (size_t)blocks * threads * 8); hipMemset(d, 0, (size_t)blocks * threads * 8); hipMalloc(&t, blocks * 32);
        std::vector<long long> h(4 * blocks);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, t, 1000);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), t, blocks * 32, hipMemcpyDeviceToHost);
        long long t0 = h[0], t1 = h[1];
        std::map<long long, int> per_cu;
        for (int b = 0; b < blocks; ++b) {
            t0 = std::min(t0, h[4 * b]); t1 = std::max(t1, h[4 * b + 1]);
            const unsigned hw = (unsigned)h[4 * b + 2], xcc = (unsigned)h[4 * b + 3] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            per_cu[((long long)xcc << 16) | (se << 8) | (sh << 4) | cu]++;
        }
        double dur = 0, late = 0;
        for (int b = 0; b < blocks; ++b) { dur += (h[4 * b + 1] - h[4 * b]) / 100.0; late = std::max(late, (double)(h[4 * b] - t0) / 100.0); }
        std::map<int, int> hist;
        for (auto &kv : per_cu) hist[kv.second]++;
        printf("%d blocks x %d threads: kernel %.1f us first start to last end, average block %.1f us, latest block start +%.1f us; distinct (xcc, se, sh, cu) = %zu; blocks per CU:", blocks,
               threads, (t1 - t0) / 100.0, dur / blocks, late, per_cu.size());
        for (auto &kv : hist) printf(" %d CUs x %d", kv.second, kv.first);
        printf("\n");
        hipFree(d); hipFree(t);
    }
    return 0;
}
