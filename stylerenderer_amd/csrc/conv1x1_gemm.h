// Internal interface of the GEMM-shaped 1x1 convolution (csrc/conv1x1_gemm.hip), used by sr_conv2d_mfma_ex.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// stride-1 1x1 convolution with C % 16 == 0, N % 128 == 0, pixels % 128 == 0, 16-byte aligned operands, >= 256 tiles
bool sr_conv1x1_gemm_eligible(int64_t B, int64_t C, int64_t N, int64_t ldw, int64_t P, const void* in, const void* wt,
                              const void* out);
int sr_conv1x1_gemm_launch(float* out, const float* in, const float* wt, int64_t ldw, const float* iscale,
                           const float* oscale, const float* obias, int64_t B, int64_t C, int64_t N, int64_t P,
                           hipStream_t st);
