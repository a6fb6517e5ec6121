// loft_common.h -- shared device helpers for the gfx950 (CDNA4) LOFT kernels.
// gfx950 only: wave = 64 lanes, no CUDA / multi-backend paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LOFT_EXPORT extern "C" __attribute__((visibility("default")))

// dtype codes used across the C-ABI (include/loft_hip.h)
#define LOFT_F32 0
#define LOFT_BF16 1

#define LOFT_F16 2

// The 16-bit activation / operand type of THIS BUILD of the library.  The sources are compiled twice (bonai_amd/build.py):
// libloft_hip.so with bfloat16 (default) and libloft_hip_f16.so with IEEE binary16 (-DLOFT_ACT_F16: the reference's
// `fp16 = dict(loss_scale=512.)` configs, mmdet/core/fp16/hooks.py:11-135).  Both export the same C-ABI; in the entry-point
// names and in `bf16_t` "bf16" then reads "the build's 16-bit type" -- loft_act16_dtype() says which one a library was built for.
// Everything type-specific goes through the helpers below (conversions, the packed-pair forms, the constant 1.0 and the MFMA).
typedef uint16_t bf16_t;  // raw bits of the 16-bit type
typedef __attribute__((ext_vector_type(2))) float loft_f32x2;
typedef __attribute__((ext_vector_type(8))) short loft_s16x8;
typedef __attribute__((ext_vector_type(16))) float loft_f32x16;

#ifdef LOFT_ACT_F16
#define LOFT_ACT16 LOFT_F16
#define LOFT_ONE16 0x3c00          /* 1.0 */
typedef __attribute__((ext_vector_type(2))) _Float16 loft_h16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 loft_h16x8;

__host__ __device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// round-to-nearest-even, overflow -> inf, NaN preserved (torch's float->half cast)
__host__ __device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
__device__ __forceinline__ void unpack2_16(uint32_t w, float& lo, float& hi) {
    const loft_h16x2 h = __builtin_bit_cast(loft_h16x2, w);
    lo = (float)h[0]; hi = (float)h[1];
}
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    const loft_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, loft_h16x2));
}
#define LOFT_MFMA_32x32x16(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(loft_h16x8, a), __builtin_bit_cast(loft_h16x8, b), c, 0, 0, 0)
#else
#define LOFT_ACT16 LOFT_BF16
#define LOFT_ONE16 0x3f80          /* 1.0 */
typedef __attribute__((ext_vector_type(2))) __bf16 loft_bf16x2;

__host__ __device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)v) << 16;
    return c.f;
}

// two fp32 -> packed pair with the hardware conversion (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN) -- the
// software form below costs a compare + divergent branch per element, which in the conv epilogues meant thousands of tiny basic
// blocks (and register spills) per workgroup
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    const loft_f32x2 v = {lo, hi};
    const loft_bf16x2 r = __builtin_convertvector(v, loft_bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
// round-to-nearest-even, NaN preserved (matches torch's float->bfloat16 cast); on the device the hardware conversion
__host__ __device__ __forceinline__ bf16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (bf16_t)(pack2_bf16(f, 0.f) & 0xffffu);
#else
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
#endif
}
__device__ __forceinline__ void unpack2_16(uint32_t w, float& lo, float& hi) {
    lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u);
}
#define LOFT_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

// 8 consecutive 16-bit values (one 16-byte access) <-> fp32
__device__ __forceinline__ void unpack8_16(const uint4 t, float v[8]) {
    unpack2_16(t.x, v[0], v[1]); unpack2_16(t.y, v[2], v[3]); unpack2_16(t.z, v[4], v[5]); unpack2_16(t.w, v[6], v[7]);
}
__device__ __forceinline__ uint4 pack8_16(const float v[8]) {
    uint4 t;
    t.x = pack2_bf16(v[0], v[1]); t.y = pack2_bf16(v[2], v[3]); t.z = pack2_bf16(v[4], v[5]); t.w = pack2_bf16(v[6], v[7]);
    return t;
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
    unpack2_16(t.x, v[0], v[1]); unpack2_16(t.y, v[2], v[3]);
}
__device__ __forceinline__ void st4(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
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
