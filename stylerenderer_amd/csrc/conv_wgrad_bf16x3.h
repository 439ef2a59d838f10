// Split-bf16 weight gradient of the 1x1 convolution (opt-in spike, see conv_wgrad_bf16x3.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

bool sr_wgrad_bf16x3_enabled(char kind);   // kind: 'w' 'g' 'c' 't', see the definition
bool sr_wgrad_bf16x3_eligible(int64_t B, int64_t CU, int64_t CV, int64_t HW, const void* u, const void* v);
int64_t sr_wgrad_bf16x3_scratch_floats(int64_t B, int64_t CU, int64_t CV, int64_t HW);
int sr_wgrad_bf16x3_launch(const float* U, const float* V, const float* uscale, const float* vscale, float* partial,
                           int64_t B, int64_t CU, int64_t CV, int64_t HW, int* ks, int* UP, int* VP, hipStream_t st);

bool sr_wgrad_s2_bf16x3_eligible(int64_t B, int64_t CU, int64_t CV, int64_t UH, int64_t UW, int64_t GH, int64_t GW,
                                 const void* v);
int64_t sr_wgrad_s2_bf16x3_scratch_floats(int64_t B, int64_t CU, int64_t CV, int64_t GH, int64_t GW);
int sr_wgrad_s2_bf16x3_launch(const float* U, const float* V, const float* uscale, const float* vscale, float* partial,
                              int64_t B, int64_t CU, int64_t CV, int64_t UH, int64_t UW, int64_t GH, int64_t GW, int* ks,
                              int* UP, int* VP, hipStream_t st);
