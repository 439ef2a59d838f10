// Split-bf16 stride-2 3x3 convolution (opt-in spike, see conv_s2_bf16x3.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

bool sr_conv_s2_bf16x3_eligible(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW, int64_t OH, int64_t OW);
int64_t sr_conv_s2_bf16x3_scratch_floats(int64_t C, int64_t N);
int sr_conv_s2_bf16x3_launch(float* out, const float* in, const float* wt, int64_t ldw, const float* iscale,
                             const float* oscale, const float* obias, int64_t B, int64_t C, int64_t N, int64_t IH,
                             int64_t IW, int64_t OH, int64_t OW, float* scratch, hipStream_t st);

bool sr_convt_bf16x3_eligible(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW);
int64_t sr_convt_bf16x3_scratch_floats(int64_t C, int64_t N);
int sr_convt_bf16x3_launch(float* out, const float* in, const float* wt, int64_t ldw, const float* iscale,
                           const float* oscale, const float* obias, int64_t B, int64_t C, int64_t N, int64_t IH,
                           int64_t IW, float* scratch, hipStream_t st);
