// loft_common.h -- shared device helpers for the gfx950 (CDNA4) LOFT kernels.
// gfx950 only: wave = 64 lanes, no CUDA / multi-backend paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LOFT_EXPORT extern "C" __attribute__((visibility("default")))

// dtype codes used across the C-ABI (include/loft_hip.h)
#define LOFT_F32 0
#define LOFT_BF16 1

typedef uint16_t bf16_t;  // raw bfloat16 bits

__host__ __device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)v) << 16;
    return c.f;
}

// round-to-nearest-even, NaN preserved (matches torch's float->bfloat16 cast)
__host__ __device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 4 consecutive channels as one vector access (16 B fp32 / 8 B bf16).
__device__ __forceinline__ void ld4(const float* p, float v[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void ld4(const bf16_t* p, float v[4]) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ __forceinline__ void st4(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
// two fp32 -> packed bf16 pair with the hardware conversion (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN) -- the
// software form above costs a compare + divergent branch per element, which in the conv epilogues meant thousands of tiny basic
// blocks (and register spills) per workgroup
typedef __attribute__((ext_vector_type(2))) float loft_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 loft_bf16x2;
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    const loft_f32x2 v = {lo, hi};
    const loft_bf16x2 r = __builtin_convertvector(v, loft_bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ void st4(bf16_t* p, const float v[4]) {
    uint2 t;
    t.x = pack2_bf16(v[0], v[1]);
    t.y = pack2_bf16(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = t;
}

#define LOFT_LAUNCH_CHECK()                          \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

static inline int loft_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
