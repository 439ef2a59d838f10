// Internal interface of the GEMM-shaped 1x1 convolution (csrc/conv1x1_gemm.hip), used by sr_conv2d_mfma_ex.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// stride-1 1x1 convolution with C % 16 == 0, N % 128 == 0, pixels % 128 == 0, 16-byte aligned operands, >= 256 tiles
bool sr_conv1x1_gemm_eligible(int64_t B, int64_t C, int64_t N, int64_t ldw, int64_t P, const void* in, const void* wt,
                              const void* out);
int sr_conv1x1_gemm_launch(float* out, const float* in, const float* wt, int64_t ldw, const float* iscale,
                           const float* oscale, const float* obias, int64_t B, int64_t C, int64_t N, int64_t P,
                           hipStream_t st, const float* addend = nullptr);

// tap-split stride-2 transposed 3x3 convolution (nine shifted 1x1 convolutions over the (IH+1) x (IW+1) phase grid) on the
// same tile, pixels = the flattened (sample, grid point) index; raw sums to partial[slice * 9 + tap][b][n][grid point]
bool sr_convt_taps_gemm_eligible(int64_t B, int64_t C, int64_t N, int64_t IW, int64_t ldw, int c_per_slice,
                                 const void* wt);
int sr_convt_taps_gemm_launch(float* partial, const float* in, const float* wt, int64_t ldw, const float* iscale, int64_t B,
                              int64_t C, int64_t N, int64_t IH, int64_t IW, int ks, int c_per_slice, hipStream_t st);
