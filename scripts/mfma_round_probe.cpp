// How do the gfx950 matrix cores round when they add products into the fp32 accumulator?
//   v_mfma_f32_32x32x16_bf16 (bf16 operands, what the split-bf16 kernels use) against v_mfma_f32_32x32x2_f32.
// One wave; every output element gets C + sum_k a_k * b_k with hand-picked values:
//   case 0   C = 1, one product = 0.75 ulp(1)             RNE -> 1 + ulp   truncate -> 1
//   case 1   C = 1, one product = 0.25 ulp(1)             RNE -> 1         round-up -> 1 + ulp
//   case 2   C = -1, one product = -0.75 ulp(1)           RNE -> -(1+ulp)  truncate (toward zero) -> -1
//   case 3   C = 1, two products of 0.5 ulp each (k = 0, 1): exact sum 1 + ulp; per-product truncation -> 1
//   case 4   C = 1, sixteen products of 0.25 ulp (all k of one MFMA): exact 1 + 4 ulp; per-product truncation -> 1
// build: hipcc --offload-arch=gfx950 -O2 -o build/mfma_round_probe scripts/mfma_round_probe.cpp ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k_bf16(float* out, int which, float c0) {
    // lane l holds A[row = l & 31][k = 8 * (l >> 5) .. +7] and B[k = 8 * (l >> 5) .. +7][col = l & 31]
    const int half = threadIdx.x >> 5;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)0.0f; b[i] = (__bf16)0.0f; }
    const float ulp = 1.1920928955078125e-07f;      // 2^-23
    if (which == 0 && half == 0) { a[0] = (__bf16)0.75f; b[0] = (__bf16)ulp; }
    if (which == 1 && half == 0) { a[0] = (__bf16)0.25f; b[0] = (__bf16)ulp; }
    if (which == 2 && half == 0) { a[0] = (__bf16)-0.75f; b[0] = (__bf16)ulp; }
    if (which == 3 && half == 0) { a[0] = (__bf16)0.5f; b[0] = (__bf16)ulp; a[1] = (__bf16)0.5f; b[1] = (__bf16)ulp; }
    if (which == 4) for (int i = 0; i < 8; ++i) { a[i] = (__bf16)0.25f; b[i] = (__bf16)ulp; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = c0;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

__global__ void k_f32(float* out, int which, float c0) {
    // v_mfma_f32_32x32x2_f32: lane l holds A[row = l & 31][k = l >> 5], B[k = l >> 5][col = l & 31]
    const int half = threadIdx.x >> 5;
    const float ulp = 1.1920928955078125e-07f;
    float a = 0.f, b = 0.f;
    if (which == 0 && half == 0) { a = 0.75f; b = ulp; }
    if (which == 1 && half == 0) { a = 0.25f; b = ulp; }
    if (which == 2 && half == 0) { a = -0.75f; b = ulp; }
    if (which == 3) { a = 0.5f; b = ulp; }
    if (which == 4) { a = 0.25f; b = ulp; }         // (two products only: 1 + 0.5 ulp -> tie -> even = 1)
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = c0;
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

int main() {
    float* d;
    hipMalloc(&d, 4);
    const float c0[5] = {1.f, 1.f, -1.f, 1.f, 1.f};
    for (int w = 0; w < 5; ++w) {
        float hb, hf;
        hipLaunchKernelGGL(k_bf16, dim3(1), dim3(64), 0, 0, d, w, c0[w]);
        hipMemcpy(&hb, d, 4, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, d, w, c0[w]);
        hipMemcpy(&hf, d, 4, hipMemcpyDeviceToHost);
        unsigned ub, uf, u1;
        memcpy(&ub, &hb, 4); memcpy(&uf, &hf, 4); memcpy(&u1, &c0[w], 4);
        printf("case %d  C = %+g   bf16 MFMA -> C %+d ulp   f32 MFMA -> C %+d ulp\n", w, c0[w], (int)(ub - u1), (int)(uf - u1));
    }
    return 0;
}
