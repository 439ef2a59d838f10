// One layer of the LPIPS distance (reference lpips/networks_basic.py:62-85 with lpips/__init__.py:42-44) as two passes
// over the feature map instead of ~25 ATen launches:
//
//   n0 = f0 / (sqrt(sum_c f0^2) + eps)                 normalize_tensor
//   d  = mean_hw sum_c lin_c * (n0_c - t_c)^2          lin layer (1x1 convolution, no bias) + spatial average
//
// f0 [B, C, H, W] are the raw trunk features of the image being optimised, t [B | 1, C, H, W] the NORMALISED features
// of the fixed target, lin [C] the learned channel weights.  A latent-inversion step (BASELINE config[4]) runs at batch
// 1, where every launch is ~5 us of a 10 ms step: launch count, not bytes.
//   k_lpips_fwd   a workgroup = 16 pixels x 16 channel slices; two sweeps over the channels (norm, then the weighted
//                 squared distance), per-pixel sums through LDS; a block sum per workgroup into a partial, summed in
//                 fixed order by k_lpips_finish (deterministic)
//   k_lpips_bwd   g f0_c = (gd / HW) * 2 * [ lin_c u_c inv  -  A inv^2 f0_c / nrm ],  u = n0 - t,  A = sum_c lin_c u_c f0_c,
//                 inv = 1 / (nrm + eps): three sweeps (norm; A; the gradient).  A pixel whose features are all zero
//                 gets a zero second term (torch's sqrt backward would give NaN there).
#include "common.h"

namespace {

// A 256-thread workgroup = 16 consecutive pixels x 16 channel slices (slice sl owns channels sl, sl + 16, ...): the
// deep layers have few pixels and many channels (512 x 16^2 at a 256^2 input), one lane per pixel would walk 512
// channels alone.  Per-pixel sums over the slices go through LDS in a fixed order.
constexpr int LB = 256, PX = 16, SL = 16;

__device__ __forceinline__ float pixel_sum(float v, float (*s_red)[PX], int sl, int px) {
    s_red[sl][px] = v;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int i = 0; i < SL; ++i) tot += s_red[i][px];
    __syncthreads();
    return tot;
}

__global__ __launch_bounds__(LB) void k_lpips_fwd(float* __restrict__ partial, const float* __restrict__ f0,
                                                  const float* __restrict__ t, const float* __restrict__ lin, int C,
                                                  int64_t hw, int64_t t_bstride, float eps) {
    __shared__ float s_red[SL][PX];
    __shared__ float s_part[LB / 64];
    const int px = threadIdx.x & (PX - 1), sl = threadIdx.x / PX;
    const int64_t b = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * PX + px;
    const bool live = p < hw;
    const float* f = f0 + b * C * hw + (live ? p : 0);
    const float* tt = t + b * t_bstride + (live ? p : 0);
    float ss = 0.0f;
    for (int c = sl; c < C; c += SL) {
        const float v = f[(int64_t)c * hw];
        ss += v * v;
    }
    ss = pixel_sum(ss, s_red, sl, px);
    const float den = sqrtf(ss) + eps;
    float acc = 0.0f;
    for (int c = sl; c < C; c += SL) {
        const float u = f[(int64_t)c * hw] / den - tt[(int64_t)c * hw];
        acc += lin[c] * (u * u);
    }
    if (!live) acc = 0.0f;
    acc = sr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = s_part[0];
        for (int i = 1; i < LB / 64; ++i) tot += s_part[i];
        partial[b * gridDim.x + blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(256) void k_lpips_finish(float* __restrict__ d, const float* __restrict__ partial, int nblk,
                                                      float inv_hw) {
    __shared__ float s_part[4];
    const int64_t b = blockIdx.x;
    float s = 0.0f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partial[b * nblk + i];     // fixed assignment, fixed tree
    s = sr_wave_sum(s);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) d[b] = (((s_part[0] + s_part[1]) + s_part[2]) + s_part[3]) * inv_hw;
}

__global__ __launch_bounds__(LB) void k_lpips_bwd(float* __restrict__ gf, const float* __restrict__ gd,
                                                  const float* __restrict__ f0, const float* __restrict__ t,
                                                  const float* __restrict__ lin, int C, int64_t hw, int64_t t_bstride,
                                                  float eps, float inv_hw) {
    __shared__ float s_red[SL][PX];
    const int px = threadIdx.x & (PX - 1), sl = threadIdx.x / PX;
    const int64_t b = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * PX + px;
    const bool live = p < hw;
    const float* f = f0 + b * C * hw + (live ? p : 0);
    const float* tt = t + b * t_bstride + (live ? p : 0);
    float ss = 0.0f;
    for (int c = sl; c < C; c += SL) {
        const float v = f[(int64_t)c * hw];
        ss += v * v;
    }
    ss = pixel_sum(ss, s_red, sl, px);
    const float nrm = sqrtf(ss), den = nrm + eps;
    float A = 0.0f;
    for (int c = sl; c < C; c += SL) {
        const float v = f[(int64_t)c * hw];
        const float u = v / den - tt[(int64_t)c * hw];
        A += lin[c] * u * v;
    }
    A = pixel_sum(A, s_red, sl, px);
    if (!live) return;
    float* g = gf + b * C * hw + p;
    const float k = 2.0f * gd[b] * inv_hw;
    const float second = nrm > 0.0f ? A / (den * den * nrm) : 0.0f;
    for (int c = sl; c < C; c += SL) {
        const float v = f[(int64_t)c * hw];
        const float u = v / den - tt[(int64_t)c * hw];
        g[(int64_t)c * hw] = k * (lin[c] * u / den - second * v);
    }
}

}  // namespace

extern "C" int64_t sr_lpips_layer_scratch_floats(int64_t b, int64_t hw) {
    if (b <= 0 || hw <= 0) return 1;
    return b * sr_ceil_div(hw, PX);
}

extern "C" int sr_lpips_layer_fwd(float* d, const float* f0, const float* t, const float* lin, int64_t b, int64_t c,
                                  int64_t hw, int64_t t_bstride, float eps, float* scratch, sr_stream_t stream) {
    if (b < 0 || c <= 0 || hw <= 0) return SR_EINVAL;
    if (b == 0) return SR_OK;
    if (!d || !f0 || !t || !lin || !scratch || (t_bstride != 0 && t_bstride != c * hw)) return SR_EINVAL;
    if (b > 65535 || c > (1 << 20) || hw > (1LL << 31)) return SR_ERANGE;
    const int nblk = (int)sr_ceil_div(hw, PX);
    hipStream_t st = sr_stream(stream);
    hipLaunchKernelGGL(k_lpips_fwd, dim3((unsigned)nblk, (unsigned)b), dim3(LB), 0, st, scratch, f0, t, lin, (int)c, hw,
                       t_bstride, eps);
    hipLaunchKernelGGL(k_lpips_finish, dim3((unsigned)b), dim3(256), 0, st, d, scratch, nblk, 1.0f / (float)hw);
    return sr_launch_status();
}

extern "C" int sr_lpips_layer_bwd(float* gf, const float* gd, const float* f0, const float* t, const float* lin,
                                  int64_t b, int64_t c, int64_t hw, int64_t t_bstride, float eps, sr_stream_t stream) {
    if (b < 0 || c <= 0 || hw <= 0) return SR_EINVAL;
    if (b == 0) return SR_OK;
    if (!gf || !gd || !f0 || !t || !lin || (t_bstride != 0 && t_bstride != c * hw)) return SR_EINVAL;
    if (b > 65535 || c > (1 << 20) || hw > (1LL << 31)) return SR_ERANGE;
    hipLaunchKernelGGL(k_lpips_bwd, dim3((unsigned)sr_ceil_div(hw, PX), (unsigned)b), dim3(LB), 0, sr_stream(stream), gf, gd,
                       f0, t, lin, (int)c, hw, t_bstride, eps, 1.0f / (float)hw);
    return sr_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Pixel term of the inversion loss: mean((a - b)^2) and its gradient w.r.t. a, one launch each (the loop of BASELINE
// config[4] runs at batch 1, where every launch is ~5 us of a 5.5 ms step: sub / pow / mean forward and their four
// backward launches become two).  One workgroup, fixed-order tree: deterministic.
namespace {

__global__ __launch_bounds__(1024) void k_mse_fwd(float* __restrict__ out, const float* __restrict__ a,
                                                  const float* __restrict__ b, int64_t n, float inv_n) {
    __shared__ float part[16];
    float acc = 0.0f;
    const int64_t n4 = n >> 2;
    // (sixteen loads in flight per lane: one workgroup walking a 256^2 image 48 dependent steps deep took 30 us)
#pragma unroll 8
    for (int64_t i = threadIdx.x; i < n4; i += 1024) {
        const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        acc += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 1024) {
        const float d = a[i] - b[i];
        acc += d * d;
    }
    acc = sr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int w = 0; w < 16; ++w) t += part[w];
        out[0] = t * inv_n;
    }
}

__global__ __launch_bounds__(256) void k_mse_bwd(float* __restrict__ ga, const float* __restrict__ gout,
                                                 const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                 float two_inv_n) {
    const float k = gout[0] * two_inv_n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) ga[i] = k * (a[i] - b[i]);
}

}  // namespace

extern "C" int sr_mse_fwd(float* out, const float* a, const float* b, int64_t n, sr_stream_t stream) {
    if (n <= 0 || !out || !a || !b) return SR_EINVAL;
    if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) != 0) return SR_EINVAL;
    hipLaunchKernelGGL(k_mse_fwd, dim3(1), dim3(1024), 0, sr_stream(stream), out, a, b, n, 1.0f / (float)n);
    return sr_launch_status();
}

extern "C" int sr_mse_bwd(float* ga, const float* gout, const float* a, const float* b, int64_t n, sr_stream_t stream) {
    if (n <= 0 || !ga || !gout || !a || !b) return SR_EINVAL;
    hipLaunchKernelGGL(k_mse_bwd, dim3(sr_stream_grid(n, 256)), dim3(256), 0, sr_stream(stream), ga, gout, a, b, n,
                       2.0f / (float)n);
    return sr_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// 2 x 2 / stride 2 max pooling of the VGG trunk (torchvision features 4 / 9 / 16 / 23, reference
// lpips/pretrained_networks.py:97-135) and its gradient, one launch each.  The gradient recomputes the arg-max from the
// saved input with torch's rule (the first maximum of the window in row-major order wins; a NaN wins) and writes EVERY
// input pixel — no zero-fill launch in front of it, no index tensor kept.
namespace {

__device__ __forceinline__ int pool2_argmax(float a, float b, float c, float d) {
    int k = 0;
    float m = a;
    if (b > m || b != b) { m = b; k = 1; }
    if (c > m || c != c) { m = c; k = 2; }
    if (d > m || d != d) { m = d; k = 3; }
    return k;
}

__global__ __launch_bounds__(256) void k_maxpool2_fwd(float* __restrict__ out, const float* __restrict__ x, int64_t planes,
                                                      int ih, int iw, int oh, int ow) {
    const int64_t total = planes * oh * ow;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % ow);
        const int oy = (int)((i / ow) % oh);
        const int64_t pl = i / ((int64_t)ow * oh);
        const float* s = x + (pl * ih + 2 * oy) * iw + 2 * ox;
        const float2 r0 = *reinterpret_cast<const float2*>(s), r1 = *reinterpret_cast<const float2*>(s + iw);
        const float v[4] = {r0.x, r0.y, r1.x, r1.y};
        out[i] = v[pool2_argmax(r0.x, r0.y, r1.x, r1.y)];
    }
}

__global__ __launch_bounds__(256) void k_maxpool2_bwd(float* __restrict__ gx, const float* __restrict__ gy,
                                                      const float* __restrict__ x, int64_t planes, int ih, int iw, int oh,
                                                      int ow) {
    // one lane per 2 x 2 window (the rows / columns beyond 2 * oh, 2 * ow of an odd map get zero from the last window's lane)
    const int64_t total = planes * oh * ow;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % ow);
        const int oy = (int)((i / ow) % oh);
        const int64_t pl = i / ((int64_t)ow * oh);
        const int64_t o = (pl * ih + 2 * oy) * iw + 2 * ox;
        const float2 r0 = *reinterpret_cast<const float2*>(x + o), r1 = *reinterpret_cast<const float2*>(x + o + iw);
        const int k = pool2_argmax(r0.x, r0.y, r1.x, r1.y);
        const float g = gy[i];
        *reinterpret_cast<float2*>(gx + o) = make_float2(k == 0 ? g : 0.0f, k == 1 ? g : 0.0f);
        *reinterpret_cast<float2*>(gx + o + iw) = make_float2(k == 2 ? g : 0.0f, k == 3 ? g : 0.0f);
    }
}

}  // namespace

extern "C" int sr_maxpool2_fwd(float* out, const float* x, int64_t planes, int64_t ih, int64_t iw, sr_stream_t stream) {
    if (planes < 0 || ih < 0 || iw < 0) return SR_EINVAL;
    if ((ih | iw) & 1) return SR_EINVAL;                                  // even maps (every trunk map is 2^k wide)
    const int64_t oh = ih / 2, ow = iw / 2, total = planes * oh * ow;
    if (total == 0) return SR_OK;
    if (!out || !x || (reinterpret_cast<uintptr_t>(x) & 7)) return SR_EINVAL;
    if (ih > 0x7FFFFFFF || iw > 0x7FFFFFFF) return SR_ERANGE;
    hipLaunchKernelGGL(k_maxpool2_fwd, dim3(sr_stream_grid(total, 256)), dim3(256), 0, sr_stream(stream), out, x, planes,
                       (int)ih, (int)iw, (int)oh, (int)ow);
    return sr_launch_status();
}

extern "C" int sr_maxpool2_bwd(float* gx, const float* gy, const float* x, int64_t planes, int64_t ih, int64_t iw,
                               sr_stream_t stream) {
    if (planes < 0 || ih < 0 || iw < 0) return SR_EINVAL;
    if ((ih | iw) & 1) return SR_EINVAL;
    const int64_t oh = ih / 2, ow = iw / 2, total = planes * oh * ow;
    if (total == 0) return SR_OK;
    if (!gx || !gy || !x || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gx)) & 7)) return SR_EINVAL;
    if (ih > 0x7FFFFFFF || iw > 0x7FFFFFFF) return SR_ERANGE;
    hipLaunchKernelGGL(k_maxpool2_bwd, dim3(sr_stream_grid(total, 256)), dim3(256), 0, sr_stream(stream), gx, gy, x, planes,
                       (int)ih, (int)iw, (int)oh, (int)ow);
    return sr_launch_status();
}
