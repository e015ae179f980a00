// shader clock under f64 load: s_memtime (core clock) vs s_memrealtime (100 MHz) around a dependent f64 FMA chain,
// plus issue cost of dependent / independent f64 VALU chains and of SALU instructions with one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, long long *t, int n, int mode) {
    double a = out[threadIdx.x], b = 1.0000001, c = 1e-9, a2 = a + 1.0, a3 = a + 2.0, a4 = a + 3.0;
    int sacc = 0;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
        if (mode == 0) {
#pragma unroll
            for (int j = 0; j < 16; j++) a = __builtin_fma(a, b, c);                       // dependent chain
        } else if (mode == 1) {
#pragma unroll
            for (int j = 0; j < 8; j++) { a = __builtin_fma(a, b, c); a2 = __builtin_fma(a2, b, c); }   // 2 chains
        } else if (mode == 2) {
#pragma unroll
            for (int j = 0; j < 4; j++) { a = __builtin_fma(a, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c); a4 = __builtin_fma(a4, b, c); }
        } else if (mode == 3) {
#pragma unroll
            for (int j = 0; j < 8; j++) { a = a * b; a2 = a2 + c; }                         // mul / add, 2 chains
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) { a = __builtin_fma(a, b, c); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc)); }   // VALU + SALU alternating
        }
    }
    long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a + a2 + a3 + a4 + sacc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
int main() {
    double *d; long long *t, h[2];
    hipMalloc(&d, 1024 * 64 * 8 * 8); hipMemset(d, 0, 1024 * 64 * 8 * 8); hipMalloc(&t, 16);
    const char *names[] = {"16 dependent fma", "2 chains x 8 fma", "4 chains x 4 fma", "2 chains mul/add x8", "8 fma + 8 s_add interleaved"};
    for (int waves = 1; waves <= 8; waves *= 2)
        for (int mode = 0; mode < 5; mode++) {
            const int n = 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(1024 * waves), dim3(64), 0, 0, d, t, n, mode);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(1024 * waves), dim3(64), 0, 0, d, t, n, mode);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            printf("waves/SIMD=%d %-30s core cycles/iter %.1f (%.2f per instr)  clock %.0f MHz  kernel %.3f ms -> %.1f TFLOP/s (fma=2)\n", waves, names[mode], (double)h[0] / n,
                   (double)h[0] / n / 16.0, (double)h[0] / ((double)h[1] / 100.0), ms, 1024.0 * waves * 64 * 16 * n * 2 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
