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

// DOT: also the row sum of gx * y0, y0 = the pre-activation input of the forward pass rebuilt from its
// output ( out / scale or out / (alpha * scale), minus noise and bias ): the demodulation gradient
// sum_p g * y of the modulated convolution in front (reference layers.py:298-300), which would otherwise
// be one more pass over two full tensors (and would keep the convolution output alive for backward).
template <bool DOT>
__global__ __launch_bounds__(EB) void k_nba_bwd(float* __restrict__ gx, float* __restrict__ partial,
                                                const float* __restrict__ gy,
                                                const float* __restrict__ out,
                                                const float* __restrict__ noise, float alpha, float scale,
                                                int c, int64_t inner, int64_t noise_bstride, int chunks,
                                                const float* __restrict__ noise_w,
                                                const float* __restrict__ bias, float inv_pos, float inv_neg,
                                                float* __restrict__ dot_partial) {
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
    float sb = 0.0f, sn = 0.0f, sd = 0.0f;
    const float nw = (DOT && noise) ? noise_w[0] : 0.0f;
    const float bb = (DOT && bias) ? bias[row - b * c] : 0.0f;
    for (int i = threadIdx.x; i < n4; i += EB) {
        const float4 g = gs[i], o = os[i];
        float4 r;
        r.x = ((o.x > 0.0f) ? g.x : g.x * alpha) * scale;
        r.y = ((o.y > 0.0f) ? g.y : g.y * alpha) * scale;
        r.z = ((o.z > 0.0f) ? g.z : g.z * alpha) * scale;
        r.w = ((o.w > 0.0f) ? g.w : g.w * alpha) * scale;
        xs[i] = r;
        sb += (r.x + r.y) + (r.z + r.w);
        float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
        if (noise) {
            nz = ns[i];
            sn += (r.x * nz.x + r.y * nz.y) + (r.z * nz.z + r.w * nz.w);
        }
        if (DOT) {
            const float y0 = ((o.x > 0.0f) ? o.x * inv_pos : o.x * inv_neg) - nw * nz.x - bb;
            const float y1 = ((o.y > 0.0f) ? o.y * inv_pos : o.y * inv_neg) - nw * nz.y - bb;
            const float y2 = ((o.z > 0.0f) ? o.z * inv_pos : o.z * inv_neg) - nw * nz.z - bb;
            const float y3 = ((o.w > 0.0f) ? o.w * inv_pos : o.w * inv_neg) - nw * nz.w - bb;
            sd += (r.x * y0 + r.y * y1) + (r.z * y2 + r.w * y3);
        }
    }
    float dummy = 0.0f;
    block_sum2(sb, sn, lds8);
    if (DOT) {
        __syncthreads();
        block_sum2(sd, dummy, lds8);
    }
    if (threadIdx.x == 0) {
        partial[(row * chunks + blockIdx.x) * 2] = sb;
        partial[(row * chunks + blockIdx.x) * 2 + 1] = sn;
        if (DOT) dot_partial[row * chunks + blockIdx.x] = sd;
    }
}

// one wave per channel: both sums of the channel (bias gradient, and the channel's share of the
// noise-strength gradient into chan_nw[ch]); k_nba_finish2 then adds the c shares in order.
__global__ __launch_bounds__(64) void k_nba_finish(float* __restrict__ gb, float* __restrict__ chan_nw,
                                                   const float* __restrict__ partial, int64_t n, int c,
                                                   int chunks) {
    const int ch = blockIdx.x;
    float acc = 0.0f, accn = 0.0f;
    const int64_t total = n * chunks;
    for (int64_t i = threadIdx.x; i < total; i += 64) {
        const int64_t s = i / chunks, k = i % chunks;
        const float2 v = reinterpret_cast<const float2*>(partial)[(s * c + ch) * chunks + k];
        acc += v.x;
        accn += v.y;
    }
    acc = sr_wave_sum(acc);
    accn = sr_wave_sum(accn);
    if (threadIdx.x == 0) {
        if (gb) gb[ch] = acc;
        chan_nw[ch] = accn;
    }
}

__global__ __launch_bounds__(64) void k_nba_finish2(float* __restrict__ gnw,
                                                    const float* __restrict__ chan_nw, int c) {
    float acc = 0.0f;
    for (int i = threadIdx.x; i < c; i += 64) acc += chan_nw[i];
    acc = sr_wave_sum(acc);
    if (threadIdx.x == 0) gnw[0] = acc;
}

// grid = (chunks, rows): partial dot products of two [rows, inner] tensors, optional scaled copy
template <bool SCALE_OUT, bool VEC>
__global__ __launch_bounds__(EB) void k_rowdot(float* __restrict__ partial, float* __restrict__ out,
                                               const float* __restrict__ a, const float* __restrict__ b,
                                               const float* __restrict__ s, int64_t inner, int chunks,
                                               const float* __restrict__ rdiv) {
    __shared__ float lds8[8];
    const int64_t row = blockIdx.y;
    const int64_t off = (int64_t)blockIdx.x * ECHUNK;
    const int64_t remain = inner - off;
    const int n = (int)(remain < ECHUNK ? remain : ECHUNK);
    const float sc = SCALE_OUT ? s[row] : 1.0f;
    float acc = 0.0f, dummy = 0.0f;
    if (VEC) {
        const float4* as = reinterpret_cast<const float4*>(a + row * inner + off);
        const float4* bs = reinterpret_cast<const float4*>(b + row * inner + off);
        float4* os = SCALE_OUT ? reinterpret_cast<float4*>(out + row * inner + off) : nullptr;
        for (int i = threadIdx.x; i < n / 4; i += EB) {
            const float4 x = as[i], y = bs[i];
            acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
            if (SCALE_OUT) os[i] = make_float4(y.x * sc, y.y * sc, y.z * sc, y.w * sc);
        }
    } else {
        // rows whose length is not a multiple of 4 (the (2^k+1)^2 maps around the stride-2 transposed
        // convolution) start at unaligned addresses: dword accesses, still fully coalesced
        const float* as = a + row * inner + off;
        const float* bs = b + row * inner + off;
        float* os = SCALE_OUT ? out + row * inner + off : nullptr;
#pragma unroll 4
        for (int i = threadIdx.x; i < n; i += EB) {
            const float x = as[i], y = bs[i];
            acc += x * y;
            if (SCALE_OUT) os[i] = y * sc;
        }
    }
    block_sum2(acc, dummy, lds8);
    // rdiv (sr_rowdot_div, one chunk per row only): the row's sum divided by rdiv[row], in place
    if (threadIdx.x == 0) partial[row * chunks + blockIdx.x] = rdiv ? acc / rdiv[row] : acc;
}

__global__ __launch_bounds__(64) void k_rowdot_finish(float* __restrict__ dots,
                                                      const float* __restrict__ partial, int chunks,
                                                      const float* __restrict__ rdiv = nullptr) {
    const int64_t row = blockIdx.x;
    float acc = 0.0f;
    for (int i = threadIdx.x; i < chunks; i += 64) acc += partial[row * chunks + i];
    acc = sr_wave_sum(acc);
    if (threadIdx.x == 0) dots[row] = rdiv ? acc / rdiv[row] : acc;
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
    return 2 * n * c * sr_ceil_div(inner, ECHUNK) + c + 2;      // partial pairs + per-channel shares
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
    hipLaunchKernelGGL(k_nba_bwd<false>, dim3(chunks, (unsigned)(n * c)), dim3(EB), 0, st, gx, scratch, gy, out,
                       noise, alpha, scale, (int)c, inner, noise_bstride, chunks, (const float*)nullptr,
                       (const float*)nullptr, 0.0f, 0.0f, (float*)nullptr);
    float* chan_nw = scratch + 2 * n * c * (int64_t)chunks;
    // gbias == NULL: the caller's bias and noise strength are frozen (sampling, inversion, the discriminator inside the
    // generator's phase) — no finish launches
    if (gbias) {
        hipLaunchKernelGGL(k_nba_finish, dim3((unsigned)c), dim3(64), 0, st, gbias, chan_nw, scratch, n, (int)c,
                           chunks);
        if (noise && gnoise_w)
            hipLaunchKernelGGL(k_nba_finish2, dim3(1), dim3(64), 0, st, gnoise_w, chan_nw, (int)c);
    }
    return sr_launch_status();
}

// The three finish launches behind a pass that left per-workgroup partials in the layout of k_nba_bwd<true>
// (csrc/upfirdn2d.hip k_fir4_nba_bwd shares them): gbias NULL = frozen parameters, rowdot NULL = one chunk (written in
// place by the producer).
int sr_nba_finish_launch(float* gbias, float* gnoise_w, float* rowdot, const float* partial, const float* dot_partial,
                         float* chan_nw, int64_t n, int64_t c, int chunks, bool has_noise, hipStream_t st) {
    if (gbias) {
        hipLaunchKernelGGL(k_nba_finish, dim3((unsigned)c), dim3(64), 0, st, gbias, chan_nw, partial, n, (int)c, chunks);
        if (has_noise && gnoise_w)
            hipLaunchKernelGGL(k_nba_finish2, dim3(1), dim3(64), 0, st, gnoise_w, chan_nw, (int)c);
    }
    if (rowdot) hipLaunchKernelGGL(k_rowdot_finish, dim3((unsigned)(n * c)), dim3(64), 0, st, rowdot, dot_partial, chunks);
    return sr_launch_status();
}

extern "C" int64_t sr_noise_bias_act_bwd_dot_scratch_floats(int64_t n, int64_t c, int64_t inner) {
    if (n <= 0 || c <= 0 || inner <= 0) return 3;
    return 3 * n * c * sr_ceil_div(inner, ECHUNK) + c + 2;      // partial pairs + row-dot partials + channel shares
}

extern "C" int sr_noise_bias_act_bwd_dot(float* gx, float* gbias, float* gnoise_w, float* rowdot, const float* gy,
                                         const float* out, const float* noise, const float* noise_w,
                                         const float* bias, float alpha, float scale, int64_t n, int64_t c,
                                         int64_t inner, int64_t noise_bstride, float* scratch,
                                         sr_stream_t stream) {
    if (n < 0 || c < 0 || inner < 0) return SR_EINVAL;
    if (n * c * inner == 0) return SR_OK;
    if (!gx || !gy || !out || !scratch || !rowdot || n * c > 65535 || (noise && !noise_w) || scale == 0.0f ||
        alpha == 0.0f)
        return SR_EINVAL;
    if (!vec_ok(inner, gx, gy, out, noise) || (noise && noise_bstride % 4 != 0)) return SR_EINVAL;
    const int chunks = (int)sr_ceil_div(inner, ECHUNK);
    hipStream_t st = sr_stream(stream);
    float* dot_scratch = scratch + 2 * n * c * (int64_t)chunks;
    float* chan_nw = dot_scratch + n * c * (int64_t)chunks;
    float* dot_partial = chunks == 1 ? rowdot : dot_scratch;              // (one chunk: in place, see sr_rowdot)
    hipLaunchKernelGGL(k_nba_bwd<true>, dim3(chunks, (unsigned)(n * c)), dim3(EB), 0, st, gx, scratch, gy, out, noise,
                       alpha, scale, (int)c, inner, noise_bstride, chunks, noise_w, bias, 1.0f / scale,
                       1.0f / (alpha * scale), dot_partial);
    if (gbias) {                                            // (NULL: frozen bias / noise strength)
        hipLaunchKernelGGL(k_nba_finish, dim3((unsigned)c), dim3(64), 0, st, gbias, chan_nw, scratch, n, (int)c, chunks);
        if (noise && gnoise_w)
            hipLaunchKernelGGL(k_nba_finish2, dim3(1), dim3(64), 0, st, gnoise_w, chan_nw, (int)c);
    }
    if (chunks > 1)
        hipLaunchKernelGGL(k_rowdot_finish, dim3((unsigned)(n * c)), dim3(64), 0, st, rowdot, dot_partial, chunks);
    return sr_launch_status();
}

// ------------------------------------------------------------------------------------------------
// StyledMapConv tail (reference model.py:49-54): per-pixel affine from the rasterised normal map between the
// modulated convolution and the noise injection,
//     y = lrelu( (x * a[b,p] + s[b,p]) + w * noise[b,p] + bias[c] ) * scale
// as ONE pass (the reference: mul, add, add, bias+act = four).  Backward in one pass as well: lanes own 4 pixels
// and walk the channels, so the per-pixel sums over channels (gradients of the two maps) stay in registers while
// the per-channel sums (bias / noise-strength gradients) leave as per-wave partials:
//     g = lrelu'(y) * gy * scale;  gx = g * a;  ga[b,p] = sum_c g * x;  gs[b,p] = sum_c g;
//     gbias[c] = sum_{b,p} g;  gw = sum g * noise                      16 B/element instead of ~60.
namespace {

constexpr int APIX = EB * 4;        // pixels per workgroup of the affine kernels

__global__ __launch_bounds__(EB) void k_nba_aff_fwd(float* __restrict__ y, const float* __restrict__ x,
                                                    const float* __restrict__ amap, const float* __restrict__ smap,
                                                    int64_t map_bstride, const float* __restrict__ noise,
                                                    const float* __restrict__ noise_w,
                                                    const float* __restrict__ bias, float alpha, float scale, int c,
                                                    int64_t inner, int64_t noise_bstride, int cgroup) {
    const int64_t b = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * APIX + threadIdx.x * 4;
    if (p >= inner) return;
    const int c0 = blockIdx.z * cgroup, c1 = (c0 + cgroup < c) ? c0 + cgroup : c;
    const float4 a = *reinterpret_cast<const float4*>(amap + b * map_bstride + p);
    const float4 sft = *reinterpret_cast<const float4*>(smap + b * map_bstride + p);
    const float nw = noise ? noise_w[0] : 0.0f;
    float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
    if (noise) nz = *reinterpret_cast<const float4*>(noise + b * noise_bstride + p);
    nz.x *= nw; nz.y *= nw; nz.z *= nw; nz.w *= nw;
    for (int ch = c0; ch < c1; ++ch) {
        const int64_t o = (b * c + ch) * inner + p;
        float4 v = *reinterpret_cast<const float4*>(x + o);
        const float bb = bias ? bias[ch] : 0.0f;
        v.x = v.x * a.x + sft.x; v.y = v.y * a.y + sft.y; v.z = v.z * a.z + sft.z; v.w = v.w * a.w + sft.w;
        if (noise) { v.x += nz.x; v.y += nz.y; v.z += nz.z; v.w += nz.w; }
        v.x += bb; v.y += bb; v.z += bb; v.w += bb;
        float4 r;
        r.x = ((v.x > 0.0f) ? v.x : v.x * alpha) * scale;
        r.y = ((v.y > 0.0f) ? v.y : v.y * alpha) * scale;
        r.z = ((v.z > 0.0f) ? v.z : v.z * alpha) * scale;
        r.w = ((v.w > 0.0f) ? v.w : v.w * alpha) * scale;
        *reinterpret_cast<float4*>(y + o) = r;
    }
}

// partial layout: [(row * chunks + chunk) * 4 + wave] float2 (bias share, noise share): k_nba_finish sums them
// with chunks' = 4 * chunks.  Channels are split into grid.z groups (a 256^2 map at batch 4 is only 256
// pixel-chunk workgroups); each group leaves its per-pixel sums in its own plane and k_plane_sum adds the planes
// in group order.
__global__ __launch_bounds__(EB) void k_nba_aff_bwd(float* __restrict__ gx, float* __restrict__ gamap,
                                                    float* __restrict__ gsmap, float* __restrict__ partial,
                                                    const float* __restrict__ gy, const float* __restrict__ out,
                                                    const float* __restrict__ x, const float* __restrict__ amap,
                                                    int64_t map_bstride, const float* __restrict__ noise,
                                                    float alpha, float scale, int c, int64_t inner,
                                                    int64_t noise_bstride, int chunks, int cgroup,
                                                    int64_t plane) {
    const int64_t b = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * APIX + threadIdx.x * 4;
    const bool live = p < inner;
    const int c0 = blockIdx.z * cgroup, c1 = (c0 + cgroup < c) ? c0 + cgroup : c;
    gamap += (int64_t)blockIdx.z * plane;
    gsmap += (int64_t)blockIdx.z * plane;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), nz = a, ga = a, gs = a;
    if (live) {
        a = *reinterpret_cast<const float4*>(amap + b * map_bstride + p);
        if (noise) nz = *reinterpret_cast<const float4*>(noise + b * noise_bstride + p);
    }
    for (int ch = c0; ch < c1; ++ch) {
        const int64_t row = b * c + ch;
        float sb = 0.0f, sn = 0.0f;
        if (live) {
            const int64_t o = row * inner + p;
            const float4 g = *reinterpret_cast<const float4*>(gy + o);
            const float4 yo = *reinterpret_cast<const float4*>(out + o);
            const float4 xv = *reinterpret_cast<const float4*>(x + o);
            float4 r;
            r.x = ((yo.x > 0.0f) ? g.x : g.x * alpha) * scale;
            r.y = ((yo.y > 0.0f) ? g.y : g.y * alpha) * scale;
            r.z = ((yo.z > 0.0f) ? g.z : g.z * alpha) * scale;
            r.w = ((yo.w > 0.0f) ? g.w : g.w * alpha) * scale;
            *reinterpret_cast<float4*>(gx + o) = make_float4(r.x * a.x, r.y * a.y, r.z * a.z, r.w * a.w);
            ga.x += r.x * xv.x; ga.y += r.y * xv.y; ga.z += r.z * xv.z; ga.w += r.w * xv.w;
            gs.x += r.x; gs.y += r.y; gs.z += r.z; gs.w += r.w;
            sb = (r.x + r.y) + (r.z + r.w);
            sn = (r.x * nz.x + r.y * nz.y) + (r.z * nz.z + r.w * nz.w);
        }
        sb = sr_wave_sum(sb);
        sn = sr_wave_sum(sn);
        if (lane == 0)
            reinterpret_cast<float2*>(partial)[(row * chunks + blockIdx.x) * 4 + wave] = make_float2(sb, sn);
    }
    if (live) {
        *reinterpret_cast<float4*>(gamap + b * inner + p) = ga;
        *reinterpret_cast<float4*>(gsmap + b * inner + p) = gs;
    }
}

// out[i] = sum over g of part[g * plane + i]; blockIdx.y selects one of two (out, part) pairs, so the two map planes of the
// StyledMapConv tail are finished by ONE launch.  A workgroup covers 64 float4 of the plane; its four waves take a quarter
// of the groups each (ascending) and the four partial sums are added in wave order: a fixed order, and the 64 groups of a
// small map (aff_cgroup) are 16 loads deep instead of 64 (one lane walking 64 planes: 16 us for a 32^2 map).
__global__ __launch_bounds__(EB) void k_plane_sum(float* __restrict__ out0, const float* __restrict__ part0,
                                                  float* __restrict__ out1, const float* __restrict__ part1,
                                                  int64_t plane, int groups) {
    __shared__ float4 s_q[3][64];
    float* out = blockIdx.y ? out1 : out0;
    const float* part = blockIdx.y ? part1 : part0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = ((int64_t)blockIdx.x * 64 + lane) * 4;
    const bool live = i < plane;
    const int per = (groups + 3) / 4;
    const int g0 = wave * per, g1 = min(groups, g0 + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
#pragma unroll 4
        for (int g = g0; g < g1; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(part + g * plane + i);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (wave > 0) s_q[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float4 v = s_q[k][lane];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(out + i) = acc;
    }
}

// channel groups so that the launch has a few thousand workgroups
inline int aff_cgroup(int64_t n, int64_t c, int64_t inner) {
    const int64_t wgs = sr_ceil_div(inner, APIX) * n;
    int64_t groups = sr_ceil_div(2048, wgs);
    // k_plane_sum walks the group planes serially: 16, or up to 64 where even 16 groups leave the launch under one
    // workgroup per CU (a 64^2 map at batch 1 — the inversion loop — was 64 workgroups walking 32 channels each)
    const int64_t cap = wgs * 16 < SR_NUM_CU ? 64 : 16;
    if (groups > cap) groups = cap;
    if (groups > c) groups = c;
    if (groups < 1) groups = 1;
    return (int)sr_ceil_div(c, groups);
}

}  // namespace

extern "C" int sr_noise_bias_act_affine(float* y, const float* x, const float* amap, const float* smap,
                                        int64_t map_bstride, const float* noise, const float* noise_w,
                                        const float* bias, float alpha, float scale, int64_t n, int64_t c,
                                        int64_t inner, int64_t noise_bstride, sr_stream_t stream) {
    if (n < 0 || c < 0 || inner < 0) return SR_EINVAL;
    if (n * c * inner == 0) return SR_OK;
    if (!y || !x || !amap || !smap || (noise && !noise_w) || n > 65535) return SR_EINVAL;
    if (!vec_ok(inner, y, x, noise, amap) || ((uintptr_t)smap & 15) || map_bstride % 4 != 0 ||
        (noise && noise_bstride % 4 != 0))
        return SR_EINVAL;
    const int cg = aff_cgroup(n, c, inner);
    hipLaunchKernelGGL(k_nba_aff_fwd, dim3((unsigned)sr_ceil_div(inner, APIX), (unsigned)n, (unsigned)sr_ceil_div(c, cg)),
                       dim3(EB), 0, sr_stream(stream), y, x, amap, smap, map_bstride, noise, noise_w, bias, alpha, scale,
                       (int)c, inner, noise_bstride, cg);
    return sr_launch_status();
}

extern "C" int64_t sr_noise_bias_act_affine_bwd_scratch_floats(int64_t n, int64_t c, int64_t inner) {
    if (n <= 0 || c <= 0 || inner <= 0) return 2;
    const int64_t groups = sr_ceil_div(c, aff_cgroup(n, c, inner));
    return 2 * n * c * sr_ceil_div(inner, APIX) * 4 + ((c + 2 + 3) / 4) * 4 + 2 * groups * n * inner;
}

extern "C" int sr_noise_bias_act_affine_bwd(float* gx, float* gamap, float* gsmap, float* gbias, float* gnoise_w,
                                            const float* gy, const float* out, const float* x, const float* amap,
                                            int64_t map_bstride, const float* noise, float alpha, float scale,
                                            int64_t n, int64_t c, int64_t inner, int64_t noise_bstride,
                                            float* scratch, sr_stream_t stream) {
    if (n < 0 || c < 0 || inner < 0) return SR_EINVAL;
    if (n * c * inner == 0) return SR_OK;
    if (!gx || !gamap || !gsmap || !gy || !out || !x || !amap || !scratch || n > 65535) return SR_EINVAL;
    if (!vec_ok(inner, gx, gy, out, x) || !vec_ok(inner, gamap, gsmap, amap, noise) || map_bstride % 4 != 0 ||
        (noise && noise_bstride % 4 != 0))
        return SR_EINVAL;
    const int chunks = (int)sr_ceil_div(inner, APIX);
    hipStream_t st = sr_stream(stream);
    const int cg = aff_cgroup(n, c, inner);
    const int groups = (int)sr_ceil_div(c, cg);
    const int64_t plane = n * inner;
    float* chan_nw = scratch + 2 * n * c * (int64_t)chunks * 4;
    float* ga_part = groups > 1 ? chan_nw + ((c + 2 + 3) / 4) * 4 : gamap;
    float* gs_part = groups > 1 ? ga_part + groups * plane : gsmap;
    hipLaunchKernelGGL(k_nba_aff_bwd, dim3((unsigned)chunks, (unsigned)n, (unsigned)groups), dim3(EB), 0, st, gx,
                       ga_part, gs_part, scratch, gy, out, x, amap, map_bstride, noise, alpha, scale, (int)c, inner,
                       noise_bstride, chunks, cg, plane);
    if (groups > 1) {
        const unsigned g1 = (unsigned)sr_ceil_div(plane, 64 * 4);
        hipLaunchKernelGGL(k_plane_sum, dim3(g1, 2), dim3(EB), 0, st, gamap, ga_part, gsmap, gs_part, plane, groups);
    }
    if (gbias) {                                            // (NULL: frozen bias / noise strength)
        hipLaunchKernelGGL(k_nba_finish, dim3((unsigned)c), dim3(64), 0, st, gbias, chan_nw, scratch, n, (int)c,
                           chunks * 4);
        if (noise && gnoise_w)
            hipLaunchKernelGGL(k_nba_finish2, dim3(1), dim3(64), 0, st, gnoise_w, chan_nw, (int)c);
    }
    return sr_launch_status();
}

// ---- second-order pass of the StyledMapConv tail (path-length regulariser, reference train.py:118-134) --------------
// First order (k_nba_aff_bwd): r = gy * m (m = LeakyReLU slope of the output sign, times gain);
//   gx = r * a,  ga[b,p] = sum_c r * x,  gs[b,p] = sum_c r,  gb[c] = sum_{b,p} r,  gnw = sum r * noise.
// Given the cotangents (Gx, Ga, Gs, Gb, Gnw) of those outputs, the gradients w.r.t. the first-order INPUTS are
//   d gy = m * (Gx * a + Ga * x + Gs + Gb[c] + Gnw * noise),   d x = r * Ga,   d a[b,p] = sum_c r * Gx
// (the mask is piecewise constant: nothing flows to the activation output, the shift plane, bias or noise weight).
// One pass over the activation (reads gy, out, x, Gx; writes d gy, d x) with the per-pixel channel sum kept in
// registers, channel groups like k_nba_aff_bwd.  Replaces ~45 tensor-algebra launches per layer of the recorded
// backward and its differentiation.
namespace {

__global__ __launch_bounds__(EB) void k_nba_aff_bwd2(float* __restrict__ d_gy, float* __restrict__ d_x,
                                                     float* __restrict__ d_amap, const float* __restrict__ Gx,
                                                     const float* __restrict__ Gmap, int64_t gmap_bstride,
                                                     const float* __restrict__ Gb, const float* __restrict__ Gnw,
                                                     const float* __restrict__ gy, const float* __restrict__ out,
                                                     const float* __restrict__ x, const float* __restrict__ amap,
                                                     int64_t map_bstride, const float* __restrict__ noise,
                                                     float alpha, float scale, int c, int64_t inner,
                                                     int64_t noise_bstride, int cgroup, int64_t plane) {
    const int64_t b = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * APIX + threadIdx.x * 4;
    if (p >= inner) return;
    const int c0 = blockIdx.z * cgroup, c1 = (c0 + cgroup < c) ? c0 + cgroup : c;
    d_amap += (int64_t)blockIdx.z * plane;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 a = *reinterpret_cast<const float4*>(amap + b * map_bstride + p);
    const float4 Ga = Gmap ? *reinterpret_cast<const float4*>(Gmap + b * gmap_bstride + p) : zero;
    const float4 Gs = Gmap ? *reinterpret_cast<const float4*>(Gmap + b * gmap_bstride + inner + p) : zero;
    float4 base = Gs;                    // the channel-independent part of T: Gs + Gnw * noise
    if (noise && Gnw) {
        const float4 nz = *reinterpret_cast<const float4*>(noise + b * noise_bstride + p);
        const float w = Gnw[0];
        base.x += w * nz.x; base.y += w * nz.y; base.z += w * nz.z; base.w += w * nz.w;
    }
    float4 da = zero;
    for (int ch = c0; ch < c1; ++ch) {
        const int64_t o = (b * c + ch) * inner + p;
        const float4 g = *reinterpret_cast<const float4*>(gy + o);
        const float4 yo = *reinterpret_cast<const float4*>(out + o);
        const float4 xv = *reinterpret_cast<const float4*>(x + o);
        const float4 gxv = Gx ? *reinterpret_cast<const float4*>(Gx + o) : zero;
        const float gb = Gb ? Gb[ch] : 0.0f;
        float4 m, r, t;
        m.x = ((yo.x > 0.0f) ? 1.0f : alpha) * scale;
        m.y = ((yo.y > 0.0f) ? 1.0f : alpha) * scale;
        m.z = ((yo.z > 0.0f) ? 1.0f : alpha) * scale;
        m.w = ((yo.w > 0.0f) ? 1.0f : alpha) * scale;
        r.x = g.x * m.x; r.y = g.y * m.y; r.z = g.z * m.z; r.w = g.w * m.w;
        t.x = ((gxv.x * a.x + Ga.x * xv.x) + base.x) + gb;
        t.y = ((gxv.y * a.y + Ga.y * xv.y) + base.y) + gb;
        t.z = ((gxv.z * a.z + Ga.z * xv.z) + base.z) + gb;
        t.w = ((gxv.w * a.w + Ga.w * xv.w) + base.w) + gb;
        *reinterpret_cast<float4*>(d_gy + o) = make_float4(m.x * t.x, m.y * t.y, m.z * t.z, m.w * t.w);
        *reinterpret_cast<float4*>(d_x + o) = make_float4(r.x * Ga.x, r.y * Ga.y, r.z * Ga.z, r.w * Ga.w);
        da.x += r.x * gxv.x; da.y += r.y * gxv.y; da.z += r.z * gxv.z; da.w += r.w * gxv.w;
    }
    *reinterpret_cast<float4*>(d_amap + b * inner + p) = da;
}

// out[b * out_bstride + i] = sum over g of part[g * plane + b * inner + i]; four waves take a quarter of the groups each
// (ascending), partial sums added in wave order — see k_plane_sum
__global__ __launch_bounds__(EB) void k_plane_sum_strided(float* __restrict__ out, int64_t out_bstride,
                                                          const float* __restrict__ part, int64_t plane,
                                                          int64_t inner, int groups) {
    __shared__ float4 s_q[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = ((int64_t)blockIdx.x * 64 + lane) * 4;
    const int64_t b = blockIdx.y;
    const bool live = i < inner;
    const int per = (groups + 3) / 4;
    const int g0 = wave * per, g1 = min(groups, g0 + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
#pragma unroll 4
        for (int g = g0; g < g1; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(part + g * plane + b * inner + i);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (wave > 0) s_q[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float4 v = s_q[k][lane];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(out + b * out_bstride + i) = acc;
    }
}

}  // namespace

extern "C" int64_t sr_noise_bias_act_affine_bwd2_scratch_floats(int64_t n, int64_t c, int64_t inner) {
    if (n <= 0 || c <= 0 || inner <= 0) return 4;
    const int64_t groups = sr_ceil_div(c, aff_cgroup(n, c, inner));
    return groups * n * inner + 4;
}

extern "C" int sr_noise_bias_act_affine_bwd2(float* d_gy, float* d_x, float* d_amap, int64_t d_amap_bstride,
                                             const float* Gx, const float* Gmap, int64_t gmap_bstride,
                                             const float* Gb, const float* Gnw, const float* gy, const float* out,
                                             const float* x, const float* amap, int64_t map_bstride,
                                             const float* noise, float alpha, float scale, int64_t n, int64_t c,
                                             int64_t inner, int64_t noise_bstride, float* scratch,
                                             sr_stream_t stream) {
    if (n < 0 || c < 0 || inner < 0) return SR_EINVAL;
    if (n * c * inner == 0) return SR_OK;
    if (!d_gy || !d_x || !d_amap || !gy || !out || !x || !amap || !scratch || n > 65535) return SR_EINVAL;
    if (!vec_ok(inner, d_gy, d_x, gy, out) || !vec_ok(inner, x, amap, noise, Gx) || !vec_ok(inner, d_amap, Gmap, 0, 0) ||
        map_bstride % 4 != 0 || d_amap_bstride % 4 != 0 || (Gmap && gmap_bstride % 4 != 0) ||
        (noise && noise_bstride % 4 != 0))
        return SR_EINVAL;
    const int chunks = (int)sr_ceil_div(inner, APIX);
    hipStream_t st = sr_stream(stream);
    const int cg = aff_cgroup(n, c, inner);
    const int groups = (int)sr_ceil_div(c, cg);
    const int64_t plane = n * inner;
    hipLaunchKernelGGL(k_nba_aff_bwd2, dim3((unsigned)chunks, (unsigned)n, (unsigned)groups), dim3(EB), 0, st, d_gy,
                       d_x, scratch, Gx, Gmap, gmap_bstride, Gb, Gnw, gy, out, x, amap, map_bstride, noise, alpha, scale,
                       (int)c, inner, noise_bstride, cg, plane);
    hipLaunchKernelGGL(k_plane_sum_strided, dim3((unsigned)sr_ceil_div(inner, 64 * 4), (unsigned)n), dim3(EB), 0, st,
                       d_amap, d_amap_bstride, scratch, plane, inner, groups);
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
    const bool vec = vec_ok(inner, a, b, out_scaled, nullptr);
    hipStream_t st = sr_stream(stream);
    const int chunks = (int)sr_ceil_div(inner > 0 ? inner : 1, ECHUNK);
    const dim3 grid(chunks, (unsigned)rows);
    // one chunk per row (maps up to 64^2): the row's single partial sum IS its dot product — written in place, no
    // finish launch (the finish kernel would add it to zero)
    float* partial = chunks == 1 ? dots : scratch;
    if (out_scaled) {
        if (vec) hipLaunchKernelGGL((k_rowdot<true, true>), grid, dim3(EB), 0, st, partial, out_scaled, a, b, scale, inner, chunks, (const float*)nullptr);
        else hipLaunchKernelGGL((k_rowdot<true, false>), grid, dim3(EB), 0, st, partial, out_scaled, a, b, scale, inner, chunks, (const float*)nullptr);
    } else {
        if (vec) hipLaunchKernelGGL((k_rowdot<false, true>), grid, dim3(EB), 0, st, partial, out_scaled, a, b, scale, inner, chunks, (const float*)nullptr);
        else hipLaunchKernelGGL((k_rowdot<false, false>), grid, dim3(EB), 0, st, partial, out_scaled, a, b, scale, inner, chunks, (const float*)nullptr);
    }
    if (chunks > 1)
        hipLaunchKernelGGL(k_rowdot_finish, dim3((unsigned)rows), dim3(64), 0, st, dots, scratch, chunks, (const float*)nullptr);
    return sr_launch_status();
}

// dots[r] = (sum_p a[r,p] * b[r,p]) / rdiv[r]: the demodulation gradient of a modulated convolution,
// sum_p g * y0 / d (op.conv.ConvFn.backward), without a separate division launch.  Same sums in the same order as
// sr_rowdot followed by an IEEE division.
extern "C" int sr_rowdot_div(float* dots, const float* a, const float* b, const float* rdiv, int64_t rows,
                             int64_t inner, float* scratch, sr_stream_t stream) {
    if (rows < 0 || inner < 0) return SR_EINVAL;
    if (rows == 0) return SR_OK;
    if (!dots || !a || !b || !rdiv || !scratch || rows > 65535) return SR_EINVAL;
    const bool vec = vec_ok(inner, a, b, nullptr, nullptr);
    hipStream_t st = sr_stream(stream);
    const int chunks = (int)sr_ceil_div(inner > 0 ? inner : 1, ECHUNK);
    const dim3 grid(chunks, (unsigned)rows);
    float* partial = chunks == 1 ? dots : scratch;
    const float* in_place = chunks == 1 ? rdiv : nullptr;
    if (vec) hipLaunchKernelGGL((k_rowdot<false, true>), grid, dim3(EB), 0, st, partial, (float*)nullptr, a, b, (const float*)nullptr, inner, chunks, in_place);
    else hipLaunchKernelGGL((k_rowdot<false, false>), grid, dim3(EB), 0, st, partial, (float*)nullptr, a, b, (const float*)nullptr, inner, chunks, in_place);
    if (chunks > 1) hipLaunchKernelGGL(k_rowdot_finish, dim3((unsigned)rows), dim3(64), 0, st, dots, scratch, chunks, rdiv);
    return sr_launch_status();
}

// Backward of the row-dot node (dots = sum_p a*b, out = b*s) in one pass — it runs in the second-order sweep of the
// path-length regulariser, where the node sits in the recorded backward of every modulated convolution:
//   ga = gd[r] * b,   gb = gd[r] * a + go * s[r],   gs[r] = sum_p go * b        (gd / go / s may be absent)
// Replaces five element-wise launches and a reduction over full activation tensors.
namespace {

template <bool VEC>
__global__ __launch_bounds__(EB) void k_rowdot_bwd(float* __restrict__ partial, float* __restrict__ ga,
                                                   float* __restrict__ gb, const float* __restrict__ a,
                                                   const float* __restrict__ b, const float* __restrict__ gd,
                                                   const float* __restrict__ go, const float* __restrict__ s,
                                                   int64_t inner, int chunks) {
    __shared__ float lds8[8];
    const int64_t row = blockIdx.y;
    const int64_t off = (int64_t)blockIdx.x * ECHUNK;
    const int64_t remain = inner - off;
    const int n = (int)(remain < ECHUNK ? remain : ECHUNK);
    const float g = gd ? gd[row] : 0.0f;
    const float sc = s ? s[row] : 1.0f;
    const int64_t base = row * inner + off;
    float acc = 0.0f, dummy = 0.0f;
    if (VEC) {
        for (int i = threadIdx.x; i < n / 4; i += EB) {
            const float4 x = reinterpret_cast<const float4*>(a + base)[i], y = reinterpret_cast<const float4*>(b + base)[i];
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (go) o = reinterpret_cast<const float4*>(go + base)[i];
            if (ga) reinterpret_cast<float4*>(ga + base)[i] = make_float4(g * y.x, g * y.y, g * y.z, g * y.w);
            if (gb)
                reinterpret_cast<float4*>(gb + base)[i] =
                    make_float4(g * x.x + o.x * sc, g * x.y + o.y * sc, g * x.z + o.z * sc, g * x.w + o.w * sc);
            acc += (o.x * y.x + o.y * y.y) + (o.z * y.z + o.w * y.w);
        }
    } else {
#pragma unroll 4
        for (int i = threadIdx.x; i < n; i += EB) {
            const float x = a[base + i], y = b[base + i];
            const float o = go ? go[base + i] : 0.0f;
            if (ga) ga[base + i] = g * y;
            if (gb) gb[base + i] = g * x + o * sc;
            acc += o * y;
        }
    }
    if (partial) {
        block_sum2(acc, dummy, lds8);
        if (threadIdx.x == 0) partial[row * chunks + blockIdx.x] = acc;
    }
}

}  // namespace

extern "C" int sr_rowdot_bwd(float* ga, float* gb, float* gs, const float* a, const float* b, const float* gd,
                             const float* go, const float* scale, int64_t rows, int64_t inner, float* scratch,
                             sr_stream_t stream) {
    if (rows < 0 || inner < 0) return SR_EINVAL;
    if (rows == 0 || inner == 0) return SR_OK;
    if (!a || !b || rows > 65535 || (gs && (!go || !scratch))) return SR_EINVAL;
    const bool vec = vec_ok(inner, a, b, ga, gb) && vec_ok(inner, go, nullptr, nullptr, nullptr);
    hipStream_t st = sr_stream(stream);
    const int chunks = (int)sr_ceil_div(inner, ECHUNK);
    const dim3 grid(chunks, (unsigned)rows);
    float* partial = gs ? (chunks == 1 ? gs : scratch) : nullptr;         // (one chunk: in place, see sr_rowdot)
    if (vec) hipLaunchKernelGGL((k_rowdot_bwd<true>), grid, dim3(EB), 0, st, partial, ga, gb, a, b, gd, go, scale, inner, chunks);
    else hipLaunchKernelGGL((k_rowdot_bwd<false>), grid, dim3(EB), 0, st, partial, ga, gb, a, b, gd, go, scale, inner, chunks);
    if (gs && chunks > 1) hipLaunchKernelGGL(k_rowdot_finish, dim3((unsigned)rows), dim3(64), 0, st, gs, scratch, chunks);
    return sr_launch_status();
}

// ------------------------------------------------------------------------------------------------
// 1x1 modulated convolution with <= 4 output channels (ToRGB, reference model.py:56-69).  A 128-wide
// MFMA tile would be 97 % idle here; the layer is one streaming pass over the activation:
//   fwd  out[b,j,p]  = sum_c ws[b,j,c] * x[b,c,p] (+ bias[j])        ws = W[j,c] * style[b,c]
//   dx   dx[b,c,p]   = sum_j ws[b,j,c] * g[b,j,p]
//   dw   dws[b,j,c]  = sum_p g[b,j,p] * x[b,c,p]
// The three maps are each other's derivatives (bilinear), so gradients of any order stay on them.
namespace {

constexpr int SC_MAXN = 4;

template <int N, int NW>
__global__ __launch_bounds__(64 * NW) void k_smallconv_fwd(float* __restrict__ out, const float* __restrict__ x,
                                                           const float* __restrict__ ws,
                                                           const float* __restrict__ bias, int C, int64_t hw4) {
    // A workgroup covers 64 float4 (256 pixels) of one sample; its NW waves split the channel
    // loop (short serial chains) and are summed through LDS in a fixed order.  Weights are wave-uniform -> scalar
    // loads.  NW = 4 where the map alone gives >= 1024 workgroups; NW = 16 below that (a 64^2 map at batch 16 is 256
    // workgroups: four waves each walking 128 channels were latency-bound at 2.6 TB/s).
    __shared__ float4 s_part[NW - 1][64][N];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* __restrict__ wsb = ws + (int64_t)b * N * C;
    const int64_t p4 = (int64_t)blockIdx.x * 64 + lane;
    const bool ok = p4 < hw4;
    const float4* xs = reinterpret_cast<const float4*>(x) + (int64_t)b * C * hw4 + (ok ? p4 : 0);
    float4 acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c_per = (C + NW - 1) / NW;
    const int c_lo = __builtin_amdgcn_readfirstlane(wave * c_per);
    const int c_hi = min(C, c_lo + c_per);
#pragma unroll 8
    for (int c = c_lo; c < c_hi; ++c) {
        const float4 v = xs[(int64_t)c * hw4];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float w = wsb[j * C + c];
            acc[j].x += w * v.x; acc[j].y += w * v.y; acc[j].z += w * v.z; acc[j].w += w * v.w;
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < N; ++j) s_part[wave - 1][lane][j] = acc[j];
    }
    __syncthreads();
    if (wave == 0 && ok) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float4 r = acc[j];
#pragma unroll
            for (int k = 0; k < NW - 1; ++k) {
                const float4 t = s_part[k][lane][j];
                r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
            }
            const float bb = bias ? bias[j] : 0.0f;
            r.x += bb; r.y += bb; r.z += bb; r.w += bb;
            reinterpret_cast<float4*>(out)[((int64_t)b * N + j) * hw4 + p4] = r;
        }
    }
}

template <int N>
__global__ __launch_bounds__(256) void k_smallconv_dx(float* __restrict__ dx, const float* __restrict__ g,
                                                      const float* __restrict__ ws, int C, int64_t hw4,
                                                      const float* __restrict__ addend, int cgroup) {
    extern __shared__ float s_ws[];
    const int b = blockIdx.y;
    const int c0 = blockIdx.z * cgroup, c1 = (c0 + cgroup < C) ? c0 + cgroup : C;
    for (int i = threadIdx.x; i < N * C; i += 256) s_ws[i] = ws[(int64_t)b * N * C + i];
    __syncthreads();
    const int64_t p4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p4 >= hw4) return;
    float4 gv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) gv[j] = reinterpret_cast<const float4*>(g)[((int64_t)b * N + j) * hw4 + p4];
    float4* dst = reinterpret_cast<float4*>(dx) + (int64_t)b * C * hw4 + p4;
    if (addend) {
        // the other gradient of a feature map that also feeds the next convolution (SmallConvFork): one pass instead
        // of this store + autograd's read-read-write addition.  Eight addend loads in flight per lane: a loop that
        // waits for one load per channel is latency-bound on the small maps
        const float4* add = reinterpret_cast<const float4*>(addend) + (int64_t)b * C * hw4 + p4;
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
            float4 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = add[(int64_t)(c + u) * hw4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float w = s_ws[j * C + c + u];
                    r.x += w * gv[j].x; r.y += w * gv[j].y; r.z += w * gv[j].z; r.w += w * gv[j].w;
                }
                r.x = a[u].x + r.x; r.y = a[u].y + r.y; r.z = a[u].z + r.z; r.w = a[u].w + r.w;
                dst[(int64_t)(c + u) * hw4] = r;
            }
        }
        for (; c < c1; ++c) {
            const float4 a = add[(int64_t)c * hw4];
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float w = s_ws[j * C + c];
                r.x += w * gv[j].x; r.y += w * gv[j].y; r.z += w * gv[j].z; r.w += w * gv[j].w;
            }
            r.x = a.x + r.x; r.y = a.y + r.y; r.z = a.z + r.z; r.w = a.w + r.w;
            dst[(int64_t)c * hw4] = r;
        }
        return;
    }
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float w = s_ws[j * C + c];
            r.x += w * gv[j].x; r.y += w * gv[j].y; r.z += w * gv[j].z; r.w += w * gv[j].w;
        }
        dst[(int64_t)c * hw4] = r;
    }
}

// grid = (chunks, B*C/CH): N partial dot products of CH x-row chunks with the N gradient rows.  The
// gradient chunk is loaded once per workgroup and reused for the CH channels (with one channel per
// workgroup the g rows were re-read C times through L2: 4x the bytes of x).
template <int N, int CH>
__global__ __launch_bounds__(256) void k_smallconv_dw(float* __restrict__ partial, const float* __restrict__ g,
                                                      const float* __restrict__ x, int C, int64_t hw,
                                                      int chunks, float* __restrict__ dws, float* __restrict__ gsum) {
    __shared__ float lds[CH][4 * SC_MAXN];
    __shared__ float lds_g[4 * SC_MAXN];
    const int64_t row0 = (int64_t)blockIdx.y * CH;   // b * C + c0, CH consecutive channels of one sample
    const int64_t b = row0 / C;
    const int64_t off = (int64_t)blockIdx.x * ECHUNK;
    const int64_t remain = hw - off;
    const int n4 = (int)((remain < ECHUNK ? remain : ECHUNK) / 4);
    float acc[CH][N];
#pragma unroll
    for (int k = 0; k < CH; ++k)
#pragma unroll
        for (int j = 0; j < N; ++j) acc[k][j] = 0.0f;
    // the bias gradient sum_{b, p} g[b, j, p] rides along (ToRGB: reference model.py:57-69 adds a [1, 3, 1, 1] bias): the
    // workgroups of a sample's FIRST channel group also sum the gradient chunk they hold anyway (was a 22 us ATen
    // reduction with three output elements per layer)
    const bool sum_g = gsum != nullptr && row0 == b * C;
    float ag[N];
#pragma unroll
    for (int j = 0; j < N; ++j) ag[j] = 0.0f;
#pragma unroll 2
    for (int i = threadIdx.x; i < n4; i += 256) {
        float4 gg[N];
#pragma unroll
        for (int j = 0; j < N; ++j) gg[j] = reinterpret_cast<const float4*>(g + (b * N + j) * hw + off)[i];
        if (sum_g) {
#pragma unroll
            for (int j = 0; j < N; ++j) ag[j] += (gg[j].x + gg[j].y) + (gg[j].z + gg[j].w);
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const float4 v = reinterpret_cast<const float4*>(x + (row0 + k) * hw + off)[i];
#pragma unroll
            for (int j = 0; j < N; ++j)
                acc[k][j] += (v.x * gg[j].x + v.y * gg[j].y) + (v.z * gg[j].z + v.w * gg[j].w);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < CH; ++k)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float r = sr_wave_sum(acc[k][j]);
            if (lane == 0) lds[k][j * 4 + wave] = r;
        }
    if (sum_g) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float r = sr_wave_sum(ag[j]);
            if (lane == 0) lds_g[j * 4 + wave] = r;
        }
    }
    __syncthreads();
    if (sum_g && threadIdx.x >= 64 && threadIdx.x < 64 + N) {
        const int j = threadIdx.x - 64;
        gsum[(b * chunks + blockIdx.x) * N + j] = (lds_g[j * 4] + lds_g[j * 4 + 1]) + (lds_g[j * 4 + 2] + lds_g[j * 4 + 3]);
    }
    if (threadIdx.x < CH * N) {
        const int k = threadIdx.x / N, j = threadIdx.x % N;
        const float r = (lds[k][j * 4] + lds[k][j * 4 + 1]) + (lds[k][j * 4 + 2] + lds[k][j * 4 + 3]);
        // one chunk per row (maps up to 64^2): this IS the result — stored in the output layout, no finish launch
        if (dws) dws[(b * N + j) * C + (row0 + k - b * C)] = r;
        else partial[((row0 + k) * chunks + blockIdx.x) * N + j] = r;
    }
}

// dws[b][j][c] = sum over chunks of partial[(b*C + c)][chunk][j]
__global__ __launch_bounds__(64) void k_smallconv_dw_finish(float* __restrict__ dws,
                                                            const float* __restrict__ partial, int C, int N,
                                                            int chunks, int64_t rows) {
    const int64_t row = blockIdx.x;
    if (row >= rows) return;
    const int64_t b = row / C, c = row % C;
    for (int j = 0; j < N; ++j) {
        float acc = 0.0f;
        for (int i = threadIdx.x; i < chunks; i += 64) acc += partial[(row * chunks + i) * N + j];
        acc = sr_wave_sum(acc);
        if (threadIdx.x == 0) dws[(b * N + j) * C + c] = acc;
    }
}

// gb[j] = sum over (sample, chunk) of gsum[(b * chunks + chunk) * N + j]: lane i takes entries i, i + 64, ..., then the
// wave's fixed-order sum
__global__ __launch_bounds__(64) void k_smallconv_gb_finish(float* __restrict__ gb, const float* __restrict__ gsum, int N,
                                                            int entries) {
    for (int j = 0; j < N; ++j) {
        float acc = 0.0f;
        for (int i = threadIdx.x; i < entries; i += 64) acc += gsum[(int64_t)i * N + j];
        acc = sr_wave_sum(acc);
        if (threadIdx.x == 0) gb[j] = acc;
    }
}

inline bool smallconv_ok(int64_t B, int64_t C, int64_t N, int64_t hw, const void* a, const void* b2) {
    return B > 0 && B <= 65535 && C > 0 && C <= 4096 && N >= 1 && N <= SC_MAXN && hw % 4 == 0 &&
           (((uintptr_t)a | (uintptr_t)b2) & 15) == 0;
}

}  // namespace

#define SR_SMALLCONV_DISPATCH(KERNEL, ...)                                   \
    switch (N) {                                                             \
        case 1: hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); break;           \
        case 2: hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); break;           \
        case 3: hipLaunchKernelGGL(KERNEL<3>, __VA_ARGS__); break;           \
        default: hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__); break;          \
    }

extern "C" int sr_smallconv_fwd(float* out, const float* x, const float* ws, const float* bias, int64_t B,
                                int64_t C, int64_t N, int64_t hw, sr_stream_t stream) {
    if (!out || !x || !ws || !smallconv_ok(B, C, N, hw, out, x)) return SR_EINVAL;
    const int64_t hw4 = hw / 4;
    const dim3 grid((unsigned)sr_ceil_div(hw4, 64), (unsigned)B);
    hipStream_t st = sr_stream(stream);
    const bool wide = sr_ceil_div(hw4, 64) * B < 1024 && C >= 64;
#define SR_SMALLCONV_FWD(NN)                                                                                          \
    if (wide) hipLaunchKernelGGL((k_smallconv_fwd<NN, 16>), grid, dim3(1024), 0, st, out, x, ws, bias, (int)C, hw4);  \
    else hipLaunchKernelGGL((k_smallconv_fwd<NN, 4>), grid, dim3(256), 0, st, out, x, ws, bias, (int)C, hw4);
    switch (N) {
        case 1: SR_SMALLCONV_FWD(1) break;
        case 2: SR_SMALLCONV_FWD(2) break;
        case 3: SR_SMALLCONV_FWD(3) break;
        default: SR_SMALLCONV_FWD(4) break;
    }
#undef SR_SMALLCONV_FWD
    return sr_launch_status();
}

extern "C" int sr_smallconv_dx_add(float* dx, const float* g, const float* ws, const float* addend, int64_t B, int64_t C,
                                   int64_t N, int64_t hw, sr_stream_t stream);

extern "C" int sr_smallconv_dx(float* dx, const float* g, const float* ws, int64_t B, int64_t C, int64_t N,
                               int64_t hw, sr_stream_t stream) {
    return sr_smallconv_dx_add(dx, g, ws, nullptr, B, C, N, hw, stream);
}

// dx = addend + W^T g  (addend [B, C, hw] or NULL; may alias dx: every element is read and written by the same lane)
extern "C" int sr_smallconv_dx_add(float* dx, const float* g, const float* ws, const float* addend, int64_t B, int64_t C,
                                   int64_t N, int64_t hw, sr_stream_t stream) {
    if (!dx || !g || !ws || !smallconv_ok(B, C, N, hw, dx, g) || ((uintptr_t)addend & 15)) return SR_EINVAL;
    const int64_t hw4 = hw / 4;
    // channel groups (multiples of 8) so that the launch has ~2048 workgroups: a 64^2 map at batch 16 is 64 pixel blocks
    const int64_t wgs = sr_ceil_div(hw4, 256) * B;
    int64_t groups = sr_ceil_div(2048, wgs);
    if (groups > sr_ceil_div(C, 8)) groups = sr_ceil_div(C, 8);
    if (groups < 1) groups = 1;
    const int cgroup = (int)(sr_ceil_div(sr_ceil_div(C, groups), 8) * 8);
    const dim3 grid((unsigned)sr_ceil_div(hw4, 256), (unsigned)B, (unsigned)sr_ceil_div(C, cgroup));
    const size_t lds = (size_t)N * C * sizeof(float);
    hipStream_t st = sr_stream(stream);
    SR_SMALLCONV_DISPATCH(k_smallconv_dx, grid, dim3(256), lds, st, dx, g, ws, (int)C, hw4, addend, cgroup);
    return sr_launch_status();
}

extern "C" int64_t sr_smallconv_dw_scratch_floats(int64_t B, int64_t C, int64_t N, int64_t hw) {
    if (B <= 0 || C <= 0 || N <= 0 || hw <= 0) return 1;
    // the weight-gradient partials, then the bias-gradient partials of sr_smallconv_dw_bias
    return B * C * sr_ceil_div(hw, ECHUNK) * N + B * sr_ceil_div(hw, ECHUNK) * N + 1;
}

// gb (optional): the bias gradient sum_{b, p} g[b, j, p] [N], summed by the same launch (+ one 64-lane finish)
extern "C" int sr_smallconv_dw_bias(float* dws, float* gb, const float* g, const float* x, int64_t B, int64_t C, int64_t N,
                                    int64_t hw, float* scratch, sr_stream_t stream) {
    if (!dws || !g || !x || !scratch || !smallconv_ok(B, C, N, hw, g, x) || B * C > 65535) return SR_EINVAL;
    const int chunks = (int)sr_ceil_div(hw, ECHUNK);
    hipStream_t st = sr_stream(stream);
    float* direct = chunks == 1 ? dws : nullptr;
    float* gsum = gb ? scratch + B * C * chunks * N : nullptr;
    if (C % 4 == 0) {
        const dim3 grid((unsigned)chunks, (unsigned)(B * C / 4));
        switch (N) {
            case 1: hipLaunchKernelGGL((k_smallconv_dw<1, 4>), grid, dim3(256), 0, st, scratch, g, x, (int)C, hw, chunks, direct, gsum); break;
            case 2: hipLaunchKernelGGL((k_smallconv_dw<2, 4>), grid, dim3(256), 0, st, scratch, g, x, (int)C, hw, chunks, direct, gsum); break;
            case 3: hipLaunchKernelGGL((k_smallconv_dw<3, 4>), grid, dim3(256), 0, st, scratch, g, x, (int)C, hw, chunks, direct, gsum); break;
            default: hipLaunchKernelGGL((k_smallconv_dw<4, 4>), grid, dim3(256), 0, st, scratch, g, x, (int)C, hw, chunks, direct, gsum); break;
        }
    } else {
        const dim3 grid((unsigned)chunks, (unsigned)(B * C));
        switch (N) {
            case 1: hipLaunchKernelGGL((k_smallconv_dw<1, 1>), grid, dim3(256), 0, st, scratch, g, x, (int)C, hw, chunks, direct, gsum); break;
            case 2: hipLaunchKernelGGL((k_smallconv_dw<2, 1>), grid, dim3(256), 0, st, scratch, g, x, (int)C, hw, chunks, direct, gsum); break;
            case 3: hipLaunchKernelGGL((k_smallconv_dw<3, 1>), grid, dim3(256), 0, st, scratch, g, x, (int)C, hw, chunks, direct, gsum); break;
            default: hipLaunchKernelGGL((k_smallconv_dw<4, 1>), grid, dim3(256), 0, st, scratch, g, x, (int)C, hw, chunks, direct, gsum); break;
        }
    }
    if (chunks > 1)
        hipLaunchKernelGGL(k_smallconv_dw_finish, dim3((unsigned)(B * C)), dim3(64), 0, st, dws, scratch, (int)C,
                           (int)N, chunks, B * C);
    if (gb)
        hipLaunchKernelGGL(k_smallconv_gb_finish, dim3(1), dim3(64), 0, st, gb, gsum, (int)N, (int)(B * chunks));
    return sr_launch_status();
}

extern "C" int sr_smallconv_dw(float* dws, const float* g, const float* x, int64_t B, int64_t C, int64_t N,
                               int64_t hw, float* scratch, sr_stream_t stream) {
    return sr_smallconv_dw_bias(dws, nullptr, g, x, B, C, N, hw, scratch, stream);
}

// ------------------------------------------------------------------------------------------------
// Modulated weight rows of the small 1x1 convolution (ToRGB, reference layers.py:293-297 without demodulation):
//   ws[b,j,c] = (scale * w[j,c]) * s[b,c]                                     one launch instead of two tensor products
// and the pull-back of a gradient dws [B,N,C] (what k_smallconv_dw produces) onto both factors:
//   gs[b,c] = sum_j dws[b,j,c] * (scale * w[j,c]);   gw[j,c] = scale * sum_b dws[b,j,c] * s[b,c]
// (fixed order over j and b: deterministic) — one launch instead of the ~6 multiply / reduce launches autograd spends.
namespace {

__global__ __launch_bounds__(256) void k_modrows_fwd(float* __restrict__ ws, const float* __restrict__ w,
                                                     const float* __restrict__ s, float scale, int B, int N, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * N * C) return;
    const int c = (int)(i % C);
    const int j = (int)((i / C) % N);
    const int b = (int)(i / ((int64_t)C * N));
    ws[i] = (w[(int64_t)j * C + c] * scale) * s[(int64_t)b * C + c];
}

// blockIdx.y < B: the style row of sample b;  otherwise weight row j = blockIdx.y - B (batch added in ascending order:
// deterministic).  One lane per (row, channel): the first form walked B x N products in each of C lanes of TWO
// workgroups — 26 us for 100 KB at batch 16.
__global__ __launch_bounds__(256) void k_modrows_bwd(float* __restrict__ gs, float* __restrict__ gw,
                                                     const float* __restrict__ dws, const float* __restrict__ w,
                                                     const float* __restrict__ s, float scale, int B, int N, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r = blockIdx.y;
    if (r < B) {
        if (!gs) return;
        float acc = 0.0f;
        for (int j = 0; j < N; ++j) acc += dws[((int64_t)r * N + j) * C + c] * (w[(int64_t)j * C + c] * scale);
        gs[(int64_t)r * C + c] = acc;
    } else {
        if (!gw) return;
        const int j = r - B;
        float acc = 0.0f;
#pragma unroll 4
        for (int b = 0; b < B; ++b) acc += dws[((int64_t)b * N + j) * C + c] * s[(int64_t)b * C + c];
        gw[(int64_t)j * C + c] = acc * scale;
    }
}

}  // namespace

extern "C" int sr_modrows_fwd(float* ws, const float* w, const float* s, float scale, int64_t B, int64_t N, int64_t C,
                              sr_stream_t stream) {
    if (B == 0 || N == 0 || C == 0) return SR_OK;
    if (!ws || !w || !s || B < 0 || N < 0 || C < 0) return SR_EINVAL;
    if (B > 65535 || N > 65535 || C > (1 << 20)) return SR_ERANGE;
    const int64_t total = B * N * C;
    hipLaunchKernelGGL(k_modrows_fwd, dim3((unsigned)sr_ceil_div(total, 256)), dim3(256), 0, sr_stream(stream), ws, w, s,
                       scale, (int)B, (int)N, (int)C);
    return sr_launch_status();
}

extern "C" int sr_modrows_bwd(float* gs, float* gw, const float* dws, const float* w, const float* s, float scale,
                              int64_t B, int64_t N, int64_t C, sr_stream_t stream) {
    if (N == 0 || C == 0 || (!gs && !gw)) return SR_OK;
    if (!dws || !w || !s || B < 0 || N < 0 || C < 0) return SR_EINVAL;
    if (B > 65535 || N > 65535 || B + N > 65535 || C > (1 << 20)) return SR_ERANGE;
    hipLaunchKernelGGL(k_modrows_bwd, dim3((unsigned)sr_ceil_div(C, 256), (unsigned)(B + N)), dim3(256), 0, sr_stream(stream), gs, gw, dws, w,
                       s, scale, (int)B, (int)N, (int)C);
    return sr_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Adam over one flat parameter buffer (the optimiser of reference train.py:529-536) as ONE pass: p, g, m, v are
// read once and p, m, v written once (28 B per parameter) instead of the ~10 multi-tensor passes of the foreach
// implementation.  Same update as torch.optim.Adam (no weight decay, no amsgrad):
//     m = m + (g - m) * (1 - b1);  v = b2 * v + (1 - b2) * g * g;
//     p = p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// `step` is a device scalar holding t (already incremented by the caller), so the launch is graph-capturable.
// Guarded form (sr_adam_flat_guarded): `guards` are positions of the gradient buffer (the first element of every
// all-reduce bucket).  A device-side wait for a bucket's signal that times out stores NaN there before its all-reduce
// runs (sr_signal_wait_poison), the SUM carries it to every rank, and every workgroup of this kernel on every rank
// then leaves WITHOUT touching p / m / v; `skipped` (pinned host word) is raised for the host's next check.  No rank
// ever applies a half-written gradient.
namespace {

struct AdamGuards {
    int n;
    long long off[16];
};

__global__ __launch_bounds__(EB) void k_adam_flat(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                                  int64_t n, float lr, float b1, float b2, float eps,
                                                  float* __restrict__ step, AdamGuards guards, int* skipped) {
    if (guards.n) {
        bool bad = false;
        for (int i = 0; i < guards.n; ++i) {
            const float x = g[guards.off[i]];
            bad = bad || (x != x);
        }
        if (bad) {
            // a refused step is no step: the caller advanced the counter in front of this launch, take that back.  Every
            // workgroup sees the same `bad` (g is read-only here) and leaves before reading step[0]: no race.
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                step[0] -= 1.0f;
                if (skipped) __hip_atomic_store(skipped, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
    }
    const float t = step[0];
    const float bc1 = 1.0f - powf(b1, t);
    const float bc2s = sqrtf(1.0f - powf(b2, t));
    const float step_size = lr / bc1;
    const float om1 = 1.0f - b1, om2 = 1.0f - b2;
    const int64_t stride = (int64_t)gridDim.x * EB;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n4; i += stride) {
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float4 pv = reinterpret_cast<float4*>(p)[i];
#define SR_ADAM1(c)                                                   \
        mv.c = mv.c + (gv.c - mv.c) * om1;                            \
        vv.c = b2 * vv.c + om2 * gv.c * gv.c;                         \
        pv.c = pv.c - step_size * (mv.c / (sqrtf(vv.c) / bc2s + eps));
        SR_ADAM1(x) SR_ADAM1(y) SR_ADAM1(z) SR_ADAM1(w)
#undef SR_ADAM1
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
        reinterpret_cast<float4*>(p)[i] = pv;
    }
    // tail (n % 4 elements)
    if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) {
        const int64_t i = 4 * n4 + threadIdx.x;
        const float gs = g[i];
        const float ms = m[i] + (gs - m[i]) * om1;
        const float vs = b2 * v[i] + om2 * gs * gs;
        m[i] = ms;
        v[i] = vs;
        p[i] = p[i] - step_size * (ms / (sqrtf(vs) / bc2s + eps));
    }
}

}  // namespace

extern "C" int sr_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                            float beta2, float eps, const float* step, sr_stream_t stream) {
    if (n < 0) return SR_EINVAL;
    if (n == 0) return SR_OK;
    if (!p || !g || !m || !v || !step) return SR_EINVAL;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return SR_EINVAL;
    const int64_t n4 = n / 4;
    AdamGuards none;
    none.n = 0;
    hipLaunchKernelGGL(k_adam_flat, dim3(sr_stream_grid(n4 > 0 ? n4 : 1, EB)), dim3(EB), 0, sr_stream(stream), p, g, m, v,
                       n4, n, lr, beta1, beta2, eps, const_cast<float*>(step), none, (int*)nullptr);
    return sr_launch_status();
}

extern "C" int sr_adam_flat_guarded(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                    float beta2, float eps, float* step, const int64_t* guard_offs, int n_guards,
                                    int32_t* skipped_host, sr_stream_t stream) {
    if (n < 0 || n_guards < 0 || n_guards > 16 || (n_guards > 0 && !guard_offs)) return SR_EINVAL;
    if (n == 0) return SR_OK;
    if (!p || !g || !m || !v || !step) return SR_EINVAL;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return SR_EINVAL;
    AdamGuards gd;
    gd.n = n_guards;
    for (int i = 0; i < n_guards; ++i) {
        if (guard_offs[i] < 0 || guard_offs[i] >= n) return SR_EINVAL;
        gd.off[i] = guard_offs[i];
    }
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(k_adam_flat, dim3(sr_stream_grid(n4 > 0 ? n4 : 1, EB)), dim3(EB), 0, sr_stream(stream), p, g, m, v,
                       n4, n, lr, beta1, beta2, eps, step, gd, skipped_host);
    return sr_launch_status();
}
