// Does vector-ALU / LDS work of a SECOND wave on a SIMD run under the fp32 MFMAs of the first?  (gfx950)
// Workgroup = 512 threads = 2 waves per SIMD, one workgroup per CU.  Waves 0..3 run a chain of independent
// v_mfma_f32_32x32x2_f32 (8 accumulators); waves 4..7 run `mode`:
//   0 exit at once   1 the same MFMA chain   2 v_pk_fma_f32 chain   3 v_fma_f32 chain   4 ds_write_b64 stream
//   5 (single wave per SIMD) the MFMA wave itself issues 2 v_pk_fma_f32 per MFMA
// build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_valu_probe scripts/mfma_valu_probe.cpp ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* ticks, int iters) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    const long long t0 = wall_clock64();
    if (wave < 4) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        float a = threadIdx.x * 1e-3f, b = 1.0f;
        f2 p = {a, b}, q = {b, a}, s = {0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                if (MODE == 5) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(s) : "v"(p), "v"(q));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(s), "v"(q));
                }
            }
        }
        for (int i = 0; i < 8; ++i) r += acc[i][0];
        r += s.x + p.y;
    } else if (MODE == 1) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        float a = threadIdx.x * 1e-3f, b = 1.0f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) r += acc[i][0];
    } else if (MODE == 2) {
        f2 p[8], q = {1.0001f, 0.9999f}, c = {1e-6f, 1e-6f};
        for (int i = 0; i < 8; ++i) p[i] = f2{threadIdx.x * 1e-3f + i, 1.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(q), "v"(c));
        for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
    } else if (MODE == 3) {
        float p[8], q = 1.0001f, c = 1e-6f;
        for (int i = 0; i < 8; ++i) p[i] = threadIdx.x * 1e-3f + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(q), "v"(c));
        for (int i = 0; i < 8; ++i) r += p[i];
    } else if (MODE == 4) {
        f2 v = {threadIdx.x * 1.f, 2.f};
        float* dst = lds + (threadIdx.x - 256) * 2;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 16; ++rep) {
                *reinterpret_cast<volatile f2*>(dst + (rep & 7) * 512) = v;
            }
        r += lds[threadIdx.x & 255];
    }
    const long long t1 = wall_clock64();
    if (blockIdx.x == 7 && (threadIdx.x & 63) == 0) ticks[wave] = t1 - t0;
    if (r == 123.456f) out[0] = r;
}

template <int MODE>
void run(const char* what, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    long long* ticks;
    hipMalloc(&ticks, 64);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, ticks, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, ticks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[8];
    hipMemcpy(h, ticks, 64, hipMemcpyDeviceToHost);
    hipFree(ticks);
    // wall_clock64 ticks at 100 MHz: 10 ns each
    const double mfma_wave_ns = h[0] * 10.0 / ((double)iters * 8), other_wave_ns = h[4] * 10.0 / (double)iters;
    const double mfma_per_simd = (double)iters * 8 * (MODE == 1 ? 2 : 1);
    const double tf = 256.0 * 4 * mfma_per_simd * 32 * 32 * 2 * 2 / (ms * 1e-3) / 1e12;
    printf("mode %d %-44s %8.3f ms | MFMA wave: %6.1f ns per MFMA | other wave: %7.1f ns per iteration | %6.1f TFLOP/s\n",
           MODE, what, ms, mfma_wave_ns, other_wave_ns, tf);
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("MFMA wave alone", out, iters);
        run<1>("two MFMA waves per SIMD", out, iters);
        run<2>("+ wave with 32 v_pk_fma_f32 per 8 MFMA", out, iters);
        run<3>("+ wave with 64 v_fma_f32 per 8 MFMA", out, iters);
        run<4>("+ wave with 16 ds_write_b64 per 8 MFMA", out, iters);
        run<5>("one wave: 2 v_pk_fma_f32 behind each MFMA", out, iters);
    }
    return 0;
}
