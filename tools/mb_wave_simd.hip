// Which SIMD does wave w of a workgroup run on, when several workgroups share a CU?  (round 4: the multi-wave MH kernel has ONE heavy
// wave per tile -- if wave 0 of every resident tile sits on the same SIMD, that SIMD is the kernel's bottleneck.)
// Every wave records HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]) and XCC_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void k(double *out, unsigned *ids, int n) {
    extern __shared__ double lds[];
    double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
    lds[threadIdx.x] = a;
    __syncthreads();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) a = __builtin_fma(a, b, c);
        __syncthreads();
    }
    out[threadIdx.x + (size_t)blockIdx.x * blockDim.x] = a + lds[threadIdx.x ^ 1];
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned w = threadIdx.x >> 6, W = blockDim.x >> 6;
        ids[2 * (blockIdx.x * W + w)] = hw; ids[2 * (blockIdx.x * W + w) + 1] = xcc;
    }
}
int main() {
    double *d; unsigned *ids;
    struct Cfg { int blocks, W, lds; };
    for (Cfg c : {Cfg{1024, 4, 39 * 1024}, Cfg{1024, 4, 70 * 1024}, Cfg{512, 8, 70 * 1024}, Cfg{256, 16, 100 * 1024}, Cfg{2048, 2, 19 * 1024}, Cfg{4096, 8, 70 * 1024}, Cfg{1024, 3, 39 * 1024}, Cfg{1024, 5, 39 * 1024}}) {
        const int threads = 64 * c.W;
        hipMalloc(&d, (size_t)c.blocks * threads * 8); hipMemset(d, 0, (size_t)c.blocks * threads * 8); hipMalloc(&ids, (size_t)c.blocks * c.W * 8);
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(k, dim3(c.blocks), dim3(threads), c.lds, 0, d, ids, 2000);
        hipDeviceSynchronize();
        std::vector<unsigned> h((size_t)c.blocks * c.W * 2);
        hipMemcpy(h.data(), ids, h.size() * 4, hipMemcpyDeviceToHost);
        // histogram: for wave index w, which SIMD; and per CU: how many wave-0s per SIMD
        std::vector<std::vector<int>> simd_of_w(c.W, std::vector<int>(4, 0));
        std::map<unsigned, std::vector<int>> w0_per_cu;
        for (int b = 0; b < c.blocks; ++b) for (int w = 0; w < c.W; ++w) {
            const unsigned hw = h[2 * ((size_t)b * c.W + w)], xcc = h[2 * ((size_t)b * c.W + w) + 1] & 0xf;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            simd_of_w[w][simd]++;
            if (w == 0) { auto &v = w0_per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]; v.resize(4); v[simd]++; }
        }
        printf("%d blocks x %d waves, %d KB LDS:\n", c.blocks, c.W, c.lds / 1024);
        for (int w = 0; w < c.W; ++w) printf("  wave %2d on SIMD 0..3: %5d %5d %5d %5d\n", w, simd_of_w[w][0], simd_of_w[w][1], simd_of_w[w][2], simd_of_w[w][3]);
        std::map<std::vector<int>, int> pat;
        for (auto &kv : w0_per_cu) pat[kv.second]++;
        printf("  wave-0s per SIMD of a CU (pattern x CUs):");
        int shown = 0;
        for (auto &kv : pat) { if (shown++ < 12) printf("  [%d %d %d %d] x %d", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second); }
        printf("\n  first blocks: ");
        for (int b = 0; b < 12 && b < c.blocks; ++b) { const unsigned hw = h[2 * ((size_t)b * c.W)], xcc = h[2 * ((size_t)b * c.W) + 1] & 0xf; printf(" b%d:x%u/se%u/cu%u/simd%u", b, xcc, (hw >> 13) & 7, (hw >> 8) & 0xf, (hw >> 4) & 3); }
        printf("\n");
        hipFree(d); hipFree(ids);
    }
    return 0;
}
