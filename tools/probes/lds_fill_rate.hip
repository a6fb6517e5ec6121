// How fast can one CU fill LDS from L2?  512-thread workgroups (one per CU, like the stream conv kernel) copy 64 KB "K-tiles" from an
// L2-resident 4 MB buffer into LDS, (0) with global_load_lds b128 (LDS-DMA), (1) with global_load_dwordx4 + ds_write_b128,
// (2) half of the tile each way.  Prints bytes per clock per CU at the measured kernel time (clock from s_memtime deltas).
// hipcc --offload-arch=gfx950 -O3 lds_fill_rate.hip -o lds_fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
template <int MODE>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, int tiles, unsigned long long* clk, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[131072];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + ((size_t)blockIdx.x * 65536 % (4 << 20));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        char* dst = lds + (t & 1) * 65536;
        const char* s = base + (size_t)(t & 7) * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) {                       // 8 x (512 threads x 16 B) = 64 KB
            const int off = (i * 8 + wave) * 1024;
            const bool dma = MODE == 0 || (MODE == 2 && i < 4);
            if (dma) __builtin_amdgcn_global_load_lds((gptr_t)(s + off + lane * 16), (lds_ptr_t)(dst + off), 16, 0, 0);
            else *reinterpret_cast<uint4*>(dst + off + lane * 16) = *reinterpret_cast<const uint4*>(s + off + lane * 16);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
    if (lds[tid * 16] == 123 && sink) sink[0] = 1.f;
}
int main() {
    char* src; hipMalloc(&src, 8 << 20); hipMemset(src, 1, 8 << 20);
    unsigned long long* clk; hipMalloc(&clk, 256 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int tiles = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, src, tiles, clk, nullptr);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, src, tiles, clk, nullptr);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, src, tiles, clk, nullptr);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        unsigned long long h[256]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        double ticks = 0; for (int i = 0; i < 256; ++i) ticks += (double)h[i]; ticks /= 256;
        // s_memtime counts at 100 MHz; the shader clock follows from the kernel time
        const double us = ms * 1e3, bytes = (double)tiles * 65536;
        printf("%s: %.1f us per launch, %.2f us per 64 KB tile, %.1f GB/s per CU, %.1f TB/s chip (s_memtime ticks/tile %.1f)\n",
               mode == 0 ? "LDS-DMA (global_load_lds b128)   " : mode == 1 ? "registers (global_load + ds_write)" : "half and half                     ",
               us, us / tiles, bytes / us / 1e3, bytes * 256 / us / 1e6, ticks / tiles);
    }
    return 0;
}
