// Throughput of fp32 atomic adds into a small (1 MB) buffer from all CUs: agent scope (what the weight-gradient epilogue uses)
// against workgroup scope (executed in the XCD's L2) into one buffer per XCD.   hipcc --offload-arch=gfx950 -O3 atomic_scope.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SCOPE>
__global__ __launch_bounds__(256) void k(float* buf, int n_per_wg, int words, int per_xcd) {
    unsigned xcc = 0;
    if (per_xcd) xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;   // HW_REG_XCC_ID
    float* b = buf + (size_t)xcc * words;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = 0; i < n_per_wg / 256; ++i) {
        const int idx = ((i * 4 + wave) * 64 + lane) % words;
        if (SCOPE == 0) unsafeAtomicAdd(b + idx, 1.0f);
        else __hip_atomic_fetch_add(b + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
int main() {
    const int words = 262144;  // 1 MB
    float* buf; hipMalloc(&buf, (size_t)8 * words * 4); hipMemset(buf, 0, (size_t)8 * words * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int wgs : {256, 512}) for (int mode = 0; mode < 3; ++mode) {
        const int n = 65536;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, buf, n, words, 0);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, buf, n, words, 1);
            if (mode == 2) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, buf, n, words, 1);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("wgs %d mode %s: %.1f us for %.1f M atomics -> %.2f T/s\n", wgs,
               mode == 0 ? "agent, one buffer" : mode == 1 ? "workgroup scope, buffer per XCD" : "agent, buffer per XCD", ms * 1e3,
               (double)wgs * n / 1e6, (double)wgs * n / ms / 1e9);
    }
    float h[4]; hipMemcpy(h, buf, 16, hipMemcpyDeviceToHost);
    float tot = 0; for (int x = 0; x < 8; ++x) { float v; hipMemcpy(&v, buf + (size_t)x * words, 4, hipMemcpyDeviceToHost); tot += v; }
    printf("sum over the 8 buffers of word 0: %.0f\n", tot);
    return 0;
}
