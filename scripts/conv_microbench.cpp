// Standalone micro-benchmark / ablation harness for the MFMA convolution kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DSR_ABL_...] scripts/conv_microbench.cpp -o /tmp/cmb
// Runs sr_conv2d_mfma / sr_conv2d_wgrad_mfma on generator-sized layers with random data and
// prints TFLOP/s per launch (hipEvents on the launch stream).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../stylerenderer_amd/csrc/conv_mfma.hip"
#ifndef SR_NO_WGRAD
#include "../stylerenderer_amd/csrc/conv_wgrad_mfma.hip"
#endif

static float* dev_random(size_t n, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
    }
    float* d;
    hipMalloc(&d, n * sizeof(float));
    hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    return d;
}

static void run(const char* tag, int B, int C, int N, int IH, int k, int stride, int pad, int tr, int wgrad) {
    const int OH = tr ? 2 * IH + 1 : (IH + 2 * pad - k) / stride + 1;
    float* x = dev_random((size_t)B * C * IH * IH, 1);
    float* w = dev_random((size_t)k * k * C * N, 2);
    float* is = dev_random((size_t)B * C, 3);
    float* os = dev_random((size_t)B * N, 4);
    float* y = dev_random((size_t)B * N * OH * OH, 5);
    float* scratch = nullptr;
    float* cscratch = nullptr;
    if (!wgrad) {
        const long long nf = sr_conv2d_scratch_floats(B, C, N, IH, IH, OH, OH, k, stride, pad, tr);
        if (nf > 0) hipMalloc(&cscratch, nf * sizeof(float));
    }
    if (wgrad) {
        const long long nf = sr_conv2d_wgrad_scratch_floats(B, C, N, IH, IH, OH, OH, k, stride, pad, tr);
        hipMalloc(&scratch, nf * sizeof(float));
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 10;
    float best = 1e30f, tot = 0;
    for (int it = 0; it < iters + 2; ++it) {
        hipEventRecord(e0, 0);
        int rc = wgrad ? sr_conv2d_wgrad_mfma(w, x, y, is, os, B, C, N, IH, IH, OH, OH, k, stride, pad, tr, scratch, 0)
                       : sr_conv2d_mfma(y, x, w, is, os, nullptr, B, C, N, N, IH, IH, OH, OH, k, stride, pad, tr,
                                        cscratch, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        if (rc != 0) { printf("%s: rc=%d\n", tag, rc); break; }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) { tot += ms; if (ms < best) best = ms; }
    }
    const int G = tr ? IH : OH;
    const double fl = 2.0 * B * G * G * (double)C * N * k * k;
    printf("%-28s avg %.3f ms  %.1f TF   best %.3f ms %.1f TF\n", tag, tot / iters, fl / (tot / iters) / 1e9,
           best, fl / best / 1e9);
    hipFree(x); hipFree(w); hipFree(is); hipFree(os); hipFree(y);
    if (scratch) hipFree(scratch);
}

int main(int argc, char** argv) {
    const int which = argc > 1 ? atoi(argv[1]) : 0xFF;
    if (which & 1) {
        run("conv256 128->128", 16, 128, 128, 256, 3, 1, 1, 0, 0);
        run("conv128 256->256", 16, 256, 256, 128, 3, 1, 1, 0, 0);
        run("conv64 512->512", 16, 512, 512, 64, 3, 1, 1, 0, 0);
    }
    if (which & 8) {
        run("conv32 512->512", 16, 512, 512, 32, 3, 1, 1, 0, 0);
        run("conv16 512->512", 16, 512, 512, 16, 3, 1, 1, 0, 0);
        run("conv8 512->512", 16, 512, 512, 8, 3, 1, 1, 0, 0);
        run("conv4 512->512", 16, 512, 512, 4, 3, 1, 1, 0, 0);
        run("up8 512->512 (convT)", 16, 512, 512, 4, 3, 2, 0, 1, 0);
        run("up32 512->512 (convT)", 16, 512, 512, 16, 3, 2, 0, 1, 0);
        run("dgrad-up16 s2", 16, 512, 512, 17, 3, 2, 0, 0, 0);
    }
    if (which & 2) {
        run("up256 256->128 (convT)", 16, 256, 128, 128, 3, 2, 0, 1, 0);
        run("up128 512->256 (convT)", 16, 512, 256, 64, 3, 2, 0, 1, 0);
        run("dgrad-up256 128->256 s2", 16, 128, 256, 257, 3, 2, 0, 0, 0);
    }
#ifndef SR_NO_WGRAD
    if (which & 4) {
        run("wgrad256 128x128", 16, 128, 128, 256, 3, 1, 1, 0, 1);
        run("wgrad64 512x512", 16, 512, 512, 64, 3, 1, 1, 0, 1);
        run("wgrad up256 256x128", 16, 256, 128, 128, 3, 2, 0, 1, 1);
    }
#endif
    return 0;
}
