// Issue cost of VALU instruction kinds on gfx950 as a function of waves per SIMD (1024 x W blocks of one wave):
// per-wave core cycles per instruction (s_memtime around the loop, block 0) and the aggregate rate from HIP events.
// Finding (profiles/round1_f64_issue_microbench.txt): a lone wave issues a dependent f64 VALU op every ~6 cycles; the
// SIMD's f64 pipe saturates at one op per ~4.4-5 cycles with >= 2 waves, and the core clock drops from 2.4 to
// ~2.0-2.1 GHz under sustained f64 work.  32-bit VALU ops: ~9 cycles dependent in one wave, ~2.4 per SIMD saturated;
// v_mov_b64 runs at the f64 rate.  (An earlier version of this file branched on the mode inside the timed loop and
// over-reported every figure by the cost of that branch chain.)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void k(double *out, long long *t, int n) {
    double a = out[threadIdx.x], b = 1.0000001, c = 1e-9, a2 = a + 1.0, a3 = a + 2.0, a4 = a + 3.0;
    int sacc = 0, x = threadIdx.x, y = 1, z1 = threadIdx.x + 3, z2 = threadIdx.x + 5;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; j++) a = __builtin_fma(a, b, c);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; j++) { a = __builtin_fma(a, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c); a4 = __builtin_fma(a4, b, c); }
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 8; j++) { a = a * b; a2 = a2 + c; }
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 8; j++) { a = __builtin_fma(a, b, c); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc)); }
        } else if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < 8; j++) { asm volatile("v_mov_b64 %0, %1" : "=v"(a2) : "v"(a)); asm volatile("v_mov_b64 %0, %1" : "=v"(a) : "v"(a2)); }
        } else if (MODE == 5) {
#pragma unroll
            for (int j = 0; j < 8; j++) { asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "v"(x)); asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y)); }
        } else if (MODE == 6) {
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
        } else if (MODE == 7) {
#pragma unroll
            for (int j = 0; j < 8; j++) { a = __builtin_fma(a, b, c); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y)); }
        } else if (MODE == 9) {                       // the two multiplies of a Philox round
#pragma unroll
            for (int j = 0; j < 8; j++) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(y)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(y) : "v"(x)); }
        } else if (MODE == 10) {                      // independent: 16 v_mul_hi_u32 on four registers
#pragma unroll
            for (int j = 0; j < 4; j++) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(sacc)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(y) : "v"(sacc));
                                          asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(z1) : "v"(sacc)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(z2) : "v"(sacc)); }
        } else if (MODE == 11) {                      // 8 v_mul_hi_u32 + 8 v_add_f64 (do they share a pipe?)
#pragma unroll
            for (int j = 0; j < 8; j++) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(y)); a = a + c; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(y) : "v"(x), "v"(y)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(y), "v"(x)); }
        }
    }
    long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a + a2 + a3 + a4 + sacc + x + y + z1 + z2;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
template <int MODE> void run(const char *name, int waves, double *d, long long *t) {
    const int n = 20000; long long h[2];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(1024 * waves), dim3(64), 0, 0, d, t, n);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(1024 * waves), dim3(64), 0, 0, d, t, n);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)h[0] / ((double)h[1] / 100.0);
    printf("waves/SIMD=%d %-44s %5.1f cycles/instr/wave  clock %.0f MHz  kernel %.3f ms: one instr per %.1f cycles per SIMD (at the measured clock)\n", waves, name,
           (double)h[0] / n / 16.0, mhz, ms, (ms * 1e-3 * mhz * 1e6) / (16.0 * n * waves));
}
int main() {
    double *d; long long *t;
    hipMalloc(&d, 1024 * 64 * 8 * 8); hipMemset(d, 0, 1024 * 64 * 8 * 8); hipMalloc(&t, 16);
    for (int waves = 1; waves <= 8; waves *= 2) {
        run<0>("16 dependent v_fma_f64", waves, d, t);
        run<1>("4 chains x 4 v_fma_f64", waves, d, t);
        run<2>("v_mul_f64 / v_add_f64 x8", waves, d, t);
        run<4>("16 v_mov_b64", waves, d, t);
        run<5>("16 v_mov_b32", waves, d, t);
        run<6>("16 v_add_u32", waves, d, t);
        run<7>("8 v_fma_f64 + 8 v_add_u32", waves, d, t);
        run<9>("8 x (v_mul_hi_u32, v_mul_lo_u32) dependent", waves, d, t);
        run<10>("16 v_mul_hi_u32 on 4 registers", waves, d, t);
        run<11>("8 v_mul_hi_u32 + 8 v_add_f64", waves, d, t);
    }
    return 0;
}
