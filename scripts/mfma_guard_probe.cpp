// How many bits below ulp(C) does v_mfma_f32_32x32x16_bf16 keep when it aligns the 16 products of a lane row to the
// accumulator?  C = 1, product 0 = 0.5 ulp(1) (a tie: round-to-nearest-even gives 1), product 1 = 2^-g ulp(1): if the
// small product survives the alignment it breaks the tie (result 1 + ulp), if it is chopped the result is 1.
// Second sweep: C = 0 and product 0 = 1 instead (is the alignment relative to the largest PRODUCT the same?).
// build: hipcc --offload-arch=gfx950 -O2 -o build/mfma_guard_probe scripts/mfma_guard_probe.cpp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k(float* out, float c0, float a0, float b0, float a1, float b1, float a2, float b2) {
    const int half = threadIdx.x >> 5;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)0.0f; b[i] = (__bf16)0.0f; }
    if (half == 0) { a[0] = (__bf16)a0; b[0] = (__bf16)b0; a[1] = (__bf16)a1; b[1] = (__bf16)b1; a[2] = (__bf16)a2; b[2] = (__bf16)b2; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = c0;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

static float run(float* d, float c0, float a0, float b0, float a1, float b1, float a2, float b2) {
    float h;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c0, a0, b0, a1, b1, a2, b2);
    (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    return h;
}

int main() {
    float* d;
    (void)hipMalloc(&d, 4);
    const float ulp = ldexpf(1.0f, -23);
    printf("C = 1, p0 = 0.5 ulp, p1 = 2^-g ulp: survives (tie broken upward) for g =");
    for (int g = 1; g <= 40; ++g)
        if (run(d, 1.0f, 0.5f, ulp, ldexpf(1.0f, -g), ulp, 0.f, 0.f) != 1.0f) printf(" %d", g);
    printf("\nC = 0, p0 = 1, p1 = 0.5 ulp, p2 = 2^-g ulp: survives for g =");
    for (int g = 1; g <= 40; ++g)
        if (run(d, 0.0f, 1.0f, 1.0f, 0.5f, ulp, ldexpf(1.0f, -g), ulp) != 1.0f) printf(" %d", g);
    printf("\nC = 1, p0 = -(0.5 ulp of 0.5) [result exactly between 1 - ulp/4 ...]: p0 = -2^-25, p1 = -2^-g ulp: result != 1 for g =");
    for (int g = 1; g <= 40; ++g)
        if (run(d, 1.0f, -0.5f, ldexpf(1.0f, -24), -ldexpf(1.0f, -g), ulp, 0.f, 0.f) != 1.0f) printf(" %d", g);
    printf("\n");
    return 0;
}
