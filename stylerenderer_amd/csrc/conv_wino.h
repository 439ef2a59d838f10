// Internal interface of the Winograd F(2x2,3x3) path (csrc/conv_wino.hip), used by sr_conv2d_mfma.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct WinoParams {
    const float* in;
    const float* u;        // transformed weights, chunk-ordered (k_wino_weights)
    const float* iscale;
    const float* oscale;
    const float* obias;
    float* out;
    int B, C, N, H, W;
    int tiles_x, tiles_y, tiles_n;
    // optional StyledConv tail fused into the store: lrelu((y + nz_w[0]*nz[b, p]) + abias[n], alpha) * gain
    int nba;
    const float* nz;       // [B or 1, 1, H, W] or NULL
    const float* nz_w;
    const float* abias;    // [N] or NULL
    int64_t nz_bstride;
    float alpha, gain;
    // split-K (few tiles, long channel loop: batch 1 of the inversion loop, the 32^2 / 64^2 layers at batch 4): slice s
    // of `ks` handles chunks [s * kchunks, (s + 1) * kchunks) and writes its raw output-transformed tile to
    // partial[s] (layout of `out`); k_wino_reduce adds the slices in order and applies oscale / bias / the fused tail
    int ks, kchunks;
    float* partial;
};

struct WinoNba {
    const float* nz;
    const float* nz_w;
    const float* abias;
    int64_t nz_bstride;
    float alpha, gain;
};

// stride-1 3x3 pad-1 convolution with H % 8 == 0, W % 32 == 0, C % 8 == 0, N % 64 == 0
bool sr_wino_eligible(int64_t B, int64_t C, int64_t N, int64_t H, int64_t W, const void* in, const void* out);
int64_t sr_wino_scratch_floats(int64_t C, int64_t N);
// number of K slices for a call geometry (1 = no split) and the floats of partial outputs behind the U block
int sr_wino_split(int64_t B, int64_t C, int64_t N, int64_t H, int64_t W);
int64_t sr_wino_partial_floats(int64_t B, int64_t C, int64_t N, int64_t H, int64_t W);
int sr_wino_conv3x3(float* out, const float* in, const float* wt, int64_t ldw, const float* iscale,
                    const float* oscale, const float* obias, int64_t B, int64_t C, int64_t N, int64_t H, int64_t W,
                    float* u_scratch, hipStream_t st, const WinoNba* nba = nullptr, bool u_ready = false);

// Winograd weight gradient of the same convolution (csrc/conv_wgrad_wino.hip): H % 2 == 0, W % 16 == 0,
// C % 64 == 0, N % 64 == 0, B <= 32
bool sr_wgrad_wino_eligible(int64_t B, int64_t C, int64_t N, int64_t H, int64_t W, const void* x, const void* gy);
int64_t sr_wgrad_wino_scratch_floats(int64_t B, int64_t C, int64_t N, int64_t H, int64_t W);
int sr_wgrad_wino_3x3(float* dwt, const float* x, const float* gy, const float* xscale, const float* gscale,
                      int64_t B, int64_t C, int64_t N, int64_t H, int64_t W, float* scratch, hipStream_t st);
