// Fused element-wise passes of the StyledConv tail and of the modulated-conv backward (HBM bound).
//
// The reference runs these as separate full-tensor kernels:
//   NoiseInjection  image + w * noise                      reference layers.py:328-332
//   FusedLeakyReLU  lrelu(x + bias) * sqrt(2)              reference op/fused_act.py:52-62
//   grad_bias       grad_input.sum(dims)                   reference op/fused_act.py:33-38
// and autograd adds mul+sum pairs for the noise strength and the style / demodulation gradients.
// Here:
//   k_nba_fwd   y = lrelu(x + w*noise[b,hw] + bias[c]) * scale            one pass,  8 B/element
//   k_nba_bwd   gx = lrelu'(y) * gy * scale ; gbias[c] = sum gx ; gw = sum gx*noise
//                                                                          one pass, 12 B/element
//   k_rowdot    dot[r] = sum_i a[r,i]*b[r,i]  (+ optionally out[r,i] = b[r,i]*s[r])
//               = style gradient sum_p x*dxu together with dx = s*dxu, and the demodulation
//               gradient sum_p g*y                                         8 (12) B/element
// Reductions are wave-shuffle -> LDS -> one partial per workgroup -> fixed-order finish kernel:
// deterministic, no float atomics.  Arithmetic order of the activation matches csrc/fused_bias_act.hip
// (-ffp-contract=off), so y equals  fused_leaky_relu(x + w*noise, bias)  computed in two steps
// up to the single extra rounding of the fused add chain (documented tolerance: 1 ulp of the sum).
#include "common.h"

namespace {

constexpr int EB = 256;             // threads per workgroup
constexpr int ECHUNK = EB * 16;     // floats per workgroup sweep

__device__ __forceinline__ void block_sum2(float& a, float& b, float* lds8) {
    a = sr_wave_sum(a);
    b = sr_wave_sum(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { lds8[wave] = a; lds8[4 + wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = (lds8[0] + lds8[1]) + (lds8[2] + lds8[3]);
        b = (lds8[4] + lds8[5]) + (lds8[6] + lds8[7]);
    }
}

// grid = (chunks, n*c).  REF: activation slope taken from `ref` (double backward) instead of the sum.
template <bool REF>
__global__ __launch_bounds__(EB) void k_nba_fwd(float* __restrict__ y, const float* __restrict__ x,
                                                const float* __restrict__ noise,
                                                const float* __restrict__ noise_w,
                                                const float* __restrict__ bias,
                                                const float* __restrict__ ref, float alpha, float scale,
                                                int c, int64_t inner, int64_t noise_bstride) {
    const int64_t row = blockIdx.y;
    const int64_t b = row / c;
    const int ch = (int)(row - b * c);
    const float nw = noise ? noise_w[0] : 0.0f;
    const float bb = bias ? bias[ch] : 0.0f;
    const int64_t off = (int64_t)blockIdx.x * ECHUNK;
    const int64_t remain = inner - off;
    const int n4 = (int)((remain < ECHUNK ? remain : ECHUNK) / 4);
    const float4* xs = reinterpret_cast<const float4*>(x + row * inner + off);
    const float4* ns = noise ? reinterpret_cast<const float4*>(noise + b * noise_bstride + off) : nullptr;
    const float4* rs = REF ? reinterpret_cast<const float4*>(ref + row * inner + off) : nullptr;
    float4* ys = reinterpret_cast<float4*>(y + row * inner + off);
    for (int i = threadIdx.x; i < n4; i += EB) {
        float4 v = xs[i];
        if (noise) {
            const float4 nz = ns[i];
            v.x = v.x + nw * nz.x; v.y = v.y + nw * nz.y; v.z = v.z + nw * nz.z; v.w = v.w + nw * nz.w;
        }
        v.x += bb; v.y += bb; v.z += bb; v.w += bb;
        float4 r = v;
        if (REF) r = rs[i];
        float4 o;
        o.x = ((r.x > 0.0f) ? v.x : v.x * alpha) * scale;
        o.y = ((r.y > 0.0f) ? v.y : v.y * alpha) * scale;
        o.z = ((r.z > 0.0f) ? v.z : v.z * alpha) * scale;
        o.w = ((r.w > 0.0f) ? v.w : v.w * alpha) * scale;
        ys[i] = o;
    }
}

__global__ __launch_bounds__(EB) void k_nba_bwd(float* __restrict__ gx, float* __restrict__ partial,
                                                const float* __restrict__ gy,
                                                const float* __restrict__ out,
                                                const float* __restrict__ noise, float alpha, float scale,
                                                int c, int64_t inner, int64_t noise_bstride, int chunks) {
    __shared__ float lds8[8];
    const int64_t row = blockIdx.y;
    const int64_t b = row / c;
    const int64_t off = (int64_t)blockIdx.x * ECHUNK;
    const int64_t remain = inner - off;
    const int n4 = (int)((remain < ECHUNK ? remain : ECHUNK) / 4);
    const float4* gs = reinterpret_cast<const float4*>(gy + row * inner + off);
    const float4* os = reinterpret_cast<const float4*>(out + row * inner + off);
    const float4* ns = noise ? reinterpret_cast<const float4*>(noise + b * noise_bstride + off) : nullptr;
    float4* xs = reinterpret_cast<float4*>(gx + row * inner + off);
    float sb = 0.0f, sn = 0.0f;
    for (int i = threadIdx.x; i < n4; i += EB) {
        const float4 g = gs[i], o = os[i];
        float4 r;
        r.x = ((o.x > 0.0f) ? g.x : g.x * alpha) * scale;
        r.y = ((o.y > 0.0f) ? g.y : g.y * alpha) * scale;
        r.z = ((o.z > 0.0f) ? g.z : g.z * alpha) * scale;
        r.w = ((o.w > 0.0f) ? g.w : g.w * alpha) * scale;
        xs[i] = r;
        sb += (r.x + r.y) + (r.z + r.w);
        if (noise) {
            const float4 nz = ns[i];
            sn += (r.x * nz.x + r.y * nz.y) + (r.z * nz.z + r.w * nz.w);
        }
    }
    block_sum2(sb, sn, lds8);
    if (threadIdx.x == 0) {
        partial[(row * chunks + blockIdx.x) * 2] = sb;
        partial[(row * chunks + blockIdx.x) * 2 + 1] = sn;
    }
}

// one wave per channel (+ one extra workgroup for the scalar noise gradient)
__global__ __launch_bounds__(64) void k_nba_finish(float* __restrict__ gb, float* __restrict__ gnw,
                                                   const float* __restrict__ partial, int64_t n, int c,
                                                   int chunks) {
    if ((int)blockIdx.x < c) {
        const int ch = blockIdx.x;
        float acc = 0.0f;
        const int64_t total = n * chunks;
        for (int64_t i = threadIdx.x; i < total; i += 64) {
            const int64_t s = i / chunks, k = i % chunks;
            acc += partial[((s * c + ch) * chunks + k) * 2];
        }
        acc = sr_wave_sum(acc);
        if (threadIdx.x == 0 && gb) gb[ch] = acc;
    } else if (gnw) {
        float acc = 0.0f;
        const int64_t total = n * (int64_t)c * chunks;
        for (int64_t i = threadIdx.x; i < total; i += 64) acc += partial[i * 2 + 1];
        acc = sr_wave_sum(acc);
        if (threadIdx.x == 0) gnw[0] = acc;
    }
}

// grid = (chunks, rows): partial dot products of two [rows, inner] tensors, optional scaled copy
template <bool SCALE_OUT>
__global__ __launch_bounds__(EB) void k_rowdot(float* __restrict__ partial, float* __restrict__ out,
                                               const float* __restrict__ a, const float* __restrict__ b,
                                               const float* __restrict__ s, int64_t inner, int chunks) {
    __shared__ float lds8[8];
    const int64_t row = blockIdx.y;
    const int64_t off = (int64_t)blockIdx.x * ECHUNK;
    const int64_t remain = inner - off;
    const int n4 = (int)((remain < ECHUNK ? remain : ECHUNK) / 4);
    const float4* as = reinterpret_cast<const float4*>(a + row * inner + off);
    const float4* bs = reinterpret_cast<const float4*>(b + row * inner + off);
    float4* os = SCALE_OUT ? reinterpret_cast<float4*>(out + row * inner + off) : nullptr;
    const float sc = SCALE_OUT ? s[row] : 1.0f;
    float acc = 0.0f, dummy = 0.0f;
    for (int i = threadIdx.x; i < n4; i += EB) {
        const float4 x = as[i], y = bs[i];
        acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
        if (SCALE_OUT) os[i] = make_float4(y.x * sc, y.y * sc, y.z * sc, y.w * sc);
    }
    block_sum2(acc, dummy, lds8);
    if (threadIdx.x == 0) partial[row * chunks + blockIdx.x] = acc;
}

__global__ __launch_bounds__(64) void k_rowdot_finish(float* __restrict__ dots,
                                                      const float* __restrict__ partial, int chunks) {
    const int64_t row = blockIdx.x;
    float acc = 0.0f;
    for (int i = threadIdx.x; i < chunks; i += 64) acc += partial[row * chunks + i];
    acc = sr_wave_sum(acc);
    if (threadIdx.x == 0) dots[row] = acc;
}

inline bool vec_ok(int64_t inner, const void* p0, const void* p1, const void* p2, const void* p3) {
    const uintptr_t m = (uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2 | (uintptr_t)p3;
    return inner % 4 == 0 && (m & 15) == 0;
}

}  // namespace

extern "C" int sr_noise_bias_act(float* y, const float* x, const float* noise, const float* noise_w,
                                 const float* bias, const float* ref, float alpha, float scale,
                                 int64_t n, int64_t c, int64_t inner, int64_t noise_bstride,
                                 sr_stream_t stream) {
    if (n < 0 || c < 0 || inner < 0) return SR_EINVAL;
    if (n * c * inner == 0) return SR_OK;
    if (!y || !x || (noise && !noise_w) || n * c > 65535) return SR_EINVAL;
    if (!vec_ok(inner, y, x, noise, ref) || (noise && noise_bstride % 4 != 0)) return SR_EINVAL;
    const int chunks = (int)sr_ceil_div(inner, ECHUNK);
    const dim3 grid(chunks, (unsigned)(n * c));
    hipStream_t st = sr_stream(stream);
    if (ref)
        hipLaunchKernelGGL(k_nba_fwd<true>, grid, dim3(EB), 0, st, y, x, noise, noise_w, bias, ref, alpha,
                           scale, (int)c, inner, noise_bstride);
    else
        hipLaunchKernelGGL(k_nba_fwd<false>, grid, dim3(EB), 0, st, y, x, noise, noise_w, bias, ref, alpha,
                           scale, (int)c, inner, noise_bstride);
    return sr_launch_status();
}

extern "C" int64_t sr_noise_bias_act_bwd_scratch_floats(int64_t n, int64_t c, int64_t inner) {
    if (n <= 0 || c <= 0 || inner <= 0) return 2;
    return 2 * n * c * sr_ceil_div(inner, ECHUNK) + 2;
}

extern "C" int sr_noise_bias_act_bwd(float* gx, float* gbias, float* gnoise_w, const float* gy,
                                     const float* out, const float* noise, float alpha, float scale,
                                     int64_t n, int64_t c, int64_t inner, int64_t noise_bstride,
                                     float* scratch, sr_stream_t stream) {
    if (n < 0 || c < 0 || inner < 0) return SR_EINVAL;
    if (n * c * inner == 0) return SR_OK;
    if (!gx || !gy || !out || !scratch || n * c > 65535) return SR_EINVAL;
    if (!vec_ok(inner, gx, gy, out, noise) || (noise && noise_bstride % 4 != 0)) return SR_EINVAL;
    const int chunks = (int)sr_ceil_div(inner, ECHUNK);
    hipStream_t st = sr_stream(stream);
    hipLaunchKernelGGL(k_nba_bwd, dim3(chunks, (unsigned)(n * c)), dim3(EB), 0, st, gx, scratch, gy, out,
                       noise, alpha, scale, (int)c, inner, noise_bstride, chunks);
    hipLaunchKernelGGL(k_nba_finish, dim3((unsigned)c + 1), dim3(64), 0, st, gbias,
                       noise ? gnoise_w : nullptr, scratch, n, (int)c, chunks);
    return sr_launch_status();
}

extern "C" int64_t sr_rowdot_scratch_floats(int64_t rows, int64_t inner) {
    if (rows <= 0 || inner <= 0) return 1;
    return rows * sr_ceil_div(inner, ECHUNK) + 1;
}

extern "C" int sr_rowdot(float* dots, float* out_scaled, const float* a, const float* b,
                         const float* scale, int64_t rows, int64_t inner, float* scratch,
                         sr_stream_t stream) {
    if (rows < 0 || inner < 0) return SR_EINVAL;
    if (rows == 0) return SR_OK;
    if (!dots || !a || !b || !scratch || rows > 65535 || (out_scaled && !scale)) return SR_EINVAL;
    if (!vec_ok(inner, a, b, out_scaled, nullptr)) return SR_EINVAL;
    hipStream_t st = sr_stream(stream);
    const int chunks = (int)sr_ceil_div(inner > 0 ? inner : 1, ECHUNK);
    if (out_scaled)
        hipLaunchKernelGGL(k_rowdot<true>, dim3(chunks, (unsigned)rows), dim3(EB), 0, st, scratch, out_scaled,
                           a, b, scale, inner, chunks);
    else
        hipLaunchKernelGGL(k_rowdot<false>, dim3(chunks, (unsigned)rows), dim3(EB), 0, st, scratch, out_scaled,
                           a, b, scale, inner, chunks);
    hipLaunchKernelGGL(k_rowdot_finish, dim3((unsigned)rows), dim3(64), 0, st, dots, scratch, chunks);
    return sr_launch_status();
}
