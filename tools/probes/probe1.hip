// probe: glds 16B layout, ds_read_tr16_b64 mapping, mfma 32x32x16 bf16 layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

__global__ void k_glds(const uint32_t* src, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4 * 2];
    int lane = threadIdx.x;
    // each lane loads 16B from src + perm(lane)*16B; LDS dest should be base + lane*16
    int srcidx = (lane * 7) % 64;
    __builtin_amdgcn_global_load_lds((gptr_t)(src + srcidx * 4), (lds_ptr_t)(lds + 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[256 + lane * 4 + i];
}

__global__ void k_tr(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 64];
    int lane = threadIdx.x;
    for (int i = lane; i < 64 * 64; i += 64) lds[i] = (uint16_t)i;  // value = linear index; row = i/64, col = i%64
    __syncthreads();
    // each lane supplies address: within 16-lane group g=lane>>4, il = lane&15: row = il>>2 (0..3), col = (il&3)*4 + 16*g
    int il = lane & 15, g = lane >> 4;
    uint16_t* p = lds + (il >> 2) * 64 + (il & 3) * 4 + 16 * g;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}

__global__ void k_mfma(float* out) {
    // A[i][k] = (i==k) for i<16 ; B[k][j] = k*100 + j  -> D[i][j] = B[i][j] for i<16, 0 else
    int lane = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        int i = lane & 31, k = 8 * (lane >> 5) + e;
        float av = (i == k) ? 1.f : 0.f;
        int j = lane & 31;
        float bv = (float)(k * 8 + (j % 8));  // small ints exactly representable in bf16
        union { float f; uint32_t u; } ca, cb; ca.f = av; cb.f = bv;
        a[e] = (short)(ca.u >> 16); b[e] = (short)(cb.u >> 16);
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}

int main() {
    uint32_t h_src[256], *d_src, *d_out, h_out[256];
    for (int i = 0; i < 256; ++i) h_src[i] = i;
    hipMalloc(&d_src, 1024); hipMalloc(&d_out, 1024);
    hipMemcpy(d_src, h_src, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_glds, dim3(1), dim3(64), 0, 0, d_src, d_out);
    hipMemcpy(h_out, d_out, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) if (h_out[l * 4 + i] != (uint32_t)(((l * 7) % 64) * 4 + i)) ok = 0;
    printf("GLDS lane-linear: %s (lane1 got %u %u %u %u)\n", ok ? "OK" : "MISMATCH", h_out[4], h_out[5], h_out[6], h_out[7]);

    uint16_t *d_o16, h_o16[256];
    hipMalloc(&d_o16, 512);
    hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, d_o16);
    hipMemcpy(h_o16, d_o16, 512, hipMemcpyDeviceToHost);
    printf("TR16 mapping (lane: 4 values as row,col):\n");
    for (int l = 0; l < 64; ++l) {
        printf(" l%02d:", l);
        for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h_o16[l * 4 + j] / 64, h_o16[l * 4 + j] % 64);
        if (l % 4 == 3) printf("\n");
    }
    // check hypothesis: lane l (il=l&15,g) elem j == row j, col il + 16 g
    ok = 1;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        int il = l & 15, g = l >> 4;
        if (h_o16[l * 4 + j] != j * 64 + il + 16 * g) ok = 0;
    }
    printf("TR16 hypothesis (out[l][j] = M[row j][col l&15 + 16g]): %s\n", ok ? "OK" : "MISMATCH");

    float *d_f, h_f[1024];
    hipMalloc(&d_f, 4096);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, d_f);
    hipMemcpy(h_f, d_f, 4096, hipMemcpyDeviceToHost);
    ok = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float want = row < 16 ? (float)(row * 8 + (col % 8)) : 0.f;
        if (h_f[l * 16 + r] != want) ok = 0;
    }
    printf("MFMA 32x32x16 C layout (col=l&31,row=(r&3)+8(r>>2)+4(l>>5)), A[i][k] lane(i=l&31,k=8(l>>5)+e): %s\n", ok ? "OK" : "MISMATCH");
    return 0;
}
