// split_bw.hip -- stand-alone bandwidth probe of the fp32 -> two binary16 planes split (elementwise.hip: split_planes_body):
// which loop shape streams at the HBM rate?  hipcc --offload-arch=gfx950 -O3 -o /tmp/split_bw tools/probes/split_bw.hip
// Reads 4 B / element, writes 2 x 2 B / element: traffic 8 B / element.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ uint32_t pk(float a, float b) { const f32x2 v = {a, b}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h16x2)); }
__device__ __forceinline__ void upk(uint32_t w, float& a, float& b) { const h16x2 h = __builtin_bit_cast(h16x2, w); a = (float)h[0]; b = (float)h[1]; }

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void split_k(const float* __restrict__ x, uint16_t* __restrict__ planes, long nvec, long n, float sc) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < nvec; i += U * stride) {
        f32x4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4* p = reinterpret_cast<const f32x4*>(x + (i + u * stride) * 8);
            if (NTL) { a[u] = __builtin_nontemporal_load(p); b[u] = __builtin_nontemporal_load(p + 1); }
            else { a[u] = p[0]; b[u] = p[1]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float r[8] = {a[u].x * sc, a[u].y * sc, a[u].z * sc, a[u].w * sc, b[u].x * sc, b[u].y * sc, b[u].z * sc, b[u].w * sc};
            u32x4 h, l;
            float q[8];
            h.x = pk(r[0], r[1]); h.y = pk(r[2], r[3]); h.z = pk(r[4], r[5]); h.w = pk(r[6], r[7]);
            upk(h.x, q[0], q[1]); upk(h.y, q[2], q[3]); upk(h.z, q[4], q[5]); upk(h.w, q[6], q[7]);
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] -= q[k];
            l.x = pk(r[0], r[1]); l.y = pk(r[2], r[3]); l.z = pk(r[4], r[5]); l.w = pk(r[6], r[7]);
            u32x4* o0 = reinterpret_cast<u32x4*>(planes + (i + u * stride) * 8);
            u32x4* o1 = reinterpret_cast<u32x4*>(planes + n + (i + u * stride) * 8);
            if (NTS) { __builtin_nontemporal_store(h, o0); __builtin_nontemporal_store(l, o1); }
            else { *o0 = h; *o1 = l; }
        }
    }
    for (; i < nvec; i += stride) {
        const f32x4* p = reinterpret_cast<const f32x4*>(x + i * 8);
        const f32x4 a = p[0], b = p[1];
        float r[8] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc, b.x * sc, b.y * sc, b.z * sc, b.w * sc};
        u32x4 h, l;
        float q[8];
        h.x = pk(r[0], r[1]); h.y = pk(r[2], r[3]); h.z = pk(r[4], r[5]); h.w = pk(r[6], r[7]);
        upk(h.x, q[0], q[1]); upk(h.y, q[2], q[3]); upk(h.z, q[4], q[5]); upk(h.w, q[6], q[7]);
        for (int k = 0; k < 8; ++k) r[k] -= q[k];
        l.x = pk(r[0], r[1]); l.y = pk(r[2], r[3]); l.z = pk(r[4], r[5]); l.w = pk(r[6], r[7]);
        *reinterpret_cast<u32x4*>(planes + i * 8) = h;
        *reinterpret_cast<u32x4*>(planes + n + i * 8) = l;
    }
}
// 16-byte-per-lane loads that are contiguous ACROSS the wave (1 KiB per load instruction): lane handles elements 4*lane..4*lane+3
// of two consecutive 1 KiB segments?  No: packing 8 consecutive 16-bit values per 16-byte store needs 8 consecutive elements per
// lane, so this form stores 8 BYTES per lane and plane (512 B per store instruction) -- the other corner of the trade.
template <int U, bool NTS>
__global__ __launch_bounds__(256) void split_k4(const float* __restrict__ x, uint16_t* __restrict__ planes, long nvec4, long n, float sc) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < nvec4; i += U * stride) {
        f32x4 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = *reinterpret_cast<const f32x4*>(x + (i + u * stride) * 4);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float r[4] = {a[u].x * sc, a[u].y * sc, a[u].z * sc, a[u].w * sc}, q[4];
            uint2 h, l;
            h.x = pk(r[0], r[1]); h.y = pk(r[2], r[3]);
            upk(h.x, q[0], q[1]); upk(h.y, q[2], q[3]);
            for (int k = 0; k < 4; ++k) r[k] -= q[k];
            l.x = pk(r[0], r[1]); l.y = pk(r[2], r[3]);
            uint2* o0 = reinterpret_cast<uint2*>(planes + (i + u * stride) * 4);
            uint2* o1 = reinterpret_cast<uint2*>(planes + n + (i + u * stride) * 4);
            if (NTS) { __builtin_nontemporal_store(h.x, &o0->x); __builtin_nontemporal_store(h.y, &o0->y); __builtin_nontemporal_store(l.x, &o1->x); __builtin_nontemporal_store(l.y, &o1->y); }
            else { *o0 = h; *o1 = l; }
        }
    }
}
__global__ __launch_bounds__(256) void copy_k(const f32x4* __restrict__ x, f32x4* __restrict__ y, long nvec) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += stride) y[i] = x[i];
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)
template <typename F> static float time_us(F&& f, int it = 20) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(s); for (int i = 0; i < it; ++i) f(); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms * 1e3f / it;
}
int main() {
    const long sizes[] = {4l << 20, 33554432l, 67108864l, 102760448l, 134217728l};
    float* x; uint16_t* p;
    const long nmax = 134217728l;
    CK(hipMalloc(&x, nmax * 4)); CK(hipMalloc(&p, nmax * 4));
    CK(hipMemset(x, 0x3c, nmax * 4));
    for (long n : sizes) {
        const long nvec = n / 8;
        printf("n = %ld (%.0f MB fp32), GB/s of 8 B/element:\n", n, n * 4e-6);
        auto rep = [&](const char* name, float us) { printf("  %-28s %8.1f us  %7.0f GB/s\n", name, us, n * 8.0 / us * 1e-3); };
        for (int cap : {8192, 4096, 2048, 1024}) {
            long nb = (nvec + 255) / 256; if (nb > cap) nb = cap;
            char nm[64];
#define RUN(U, NTL, NTS) snprintf(nm, sizeof nm, "U%d ntl%d nts%d grid%d", U, NTL, NTS, cap); \
            rep(nm, time_us([&] { hipLaunchKernelGGL((split_k<U, NTL, NTS>), dim3(nb), dim3(256), 0, 0, x, p, nvec, n, 0.5f); }));
            RUN(1, false, false) RUN(2, false, false) RUN(4, false, false)
            RUN(1, false, true) RUN(2, false, true) RUN(4, false, true)
            RUN(2, true, true) RUN(4, true, true)
            long nb4 = (n / 4 + 255) / 256; if (nb4 > cap) nb4 = cap;
            snprintf(nm, sizeof nm, "k4 U4 nts0 grid%d", cap);
            rep(nm, time_us([&] { hipLaunchKernelGGL((split_k4<4, false>), dim3(nb4), dim3(256), 0, 0, x, p, n / 4, n, 0.5f); }));
            snprintf(nm, sizeof nm, "k4 U8 nts0 grid%d", cap);
            rep(nm, time_us([&] { hipLaunchKernelGGL((split_k4<8, false>), dim3(nb4), dim3(256), 0, 0, x, p, n / 4, n, 0.5f); }));
        }
        long nbc = (n / 4 + 255) / 256; if (nbc > 8192) nbc = 8192;
        rep("float4 copy grid8192", time_us([&] { hipLaunchKernelGGL(copy_k, dim3(nbc), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)p, n / 4); }));
    }
    return 0;
}
