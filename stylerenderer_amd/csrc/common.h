// Shared helpers for the gfx950 kernels of libstylerenderer_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stylerenderer_amd.h"

#define SR_WAVE 64          // CDNA wavefront width
#define SR_NUM_CU 256       // MI355X
#define SR_NUM_XCD 8

static inline hipStream_t sr_stream(sr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Launch-error check: no host synchronisation, just the enqueue status.
static inline int sr_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SR_OK : static_cast<int>(e);
}

static inline int64_t sr_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Memory-bound grid sizing: enough workgroups to fill 256 CUs several times over, then
// grid-stride (guide §6 G11).
static inline int sr_stream_grid(int64_t work_items, int per_block) {
    int64_t g = sr_ceil_div(work_items, per_block);
    const int64_t cap = (int64_t)SR_NUM_CU * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return static_cast<int>(g);
}

// wave64 sum via DPP-friendly shuffles
__device__ __forceinline__ float sr_wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, SR_WAVE);
    return x;
}

#if __HIP_DEVICE_COMPILE__
// Buffer resource over [base, base + bytes) for raw buffer loads / LDS-DMA.  Every input goes through readfirstlane
// so that the descriptor provably lives in SGPRs: a descriptor the compiler believes divergent turns each buffer
// operation into a waterfall loop.  The builtins exist in the device pass only, hence the guard (kernels that use
// them keep their bodies under the same guard; the host pass needs just the launch stub).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const float* base, int bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
#endif
