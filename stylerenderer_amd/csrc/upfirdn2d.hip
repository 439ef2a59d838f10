// upfirdn2d for gfx950: zero-insert upsample -> pad/crop -> 2-D FIR -> decimate.
//
// Replaces the reference's upfirdn2d_op and its six tiled "modes" + large fallback
// (reference op/upfirdn2d_kernel.cu:79-257; mode table op/upfirdn2d.cpp:50-75).  MI355X-first design:
//   * k_fir4_tile — the shape that carries all the bytes in the generator (Blur after the
//     stride-2 transposed conv and its gradient: up = down = 1, 4x4 taps; reference layers.py:272-275):
//     a 32x64 output tile per 256-thread workgroup, the 35x67 input halo tile staged once in LDS
//     (9.5 KiB, rows padded to 68 floats so every lane's 8-float window is two aligned
//     ds_read_b128), each lane producing a 2x4 output micro-tile from a register sliding window and
//     writing 16-byte stores.  HBM traffic = 4*(N_in + N_out) bytes, the algorithmic minimum.
//   * k_upfirdn_generic — any up/down/kernel/pad (ToRGB skip upsample up=2, its gradient down=2,
//     odd kernels, crops): one output per lane, consecutive lanes on consecutive columns
//     (coalesced), polyphase tap skipping instead of multiplying inserted zeros.
// Taps are accumulated ky-major then kx in fp32 with separate multiply and add
// (-ffp-contract=off), the order the oracle (oracle/ops_np.py) fixes, so HIP == oracle bitwise.
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace {

struct UfdParams {
    int in_h, in_w, out_h, out_w, kh, kw;
    int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
};

__device__ __forceinline__ int floor_div(int a, int b) {
    int q = a / b;
    return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

// ------------------------------------------------------------------------------ generic
__global__ __launch_bounds__(256) void k_upfirdn_generic(float* __restrict__ out,
                                                         const float* __restrict__ x,
                                                         const float* __restrict__ k, UfdParams p,
                                                         int64_t total) {
    extern __shared__ float s_k[];      // flipped kernel
    for (int i = threadIdx.x; i < p.kh * p.kw; i += blockDim.x) {
        const int ky = i / p.kw, kx = i % p.kw;
        s_k[i] = k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t plane_out = (int64_t)p.out_h * p.out_w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t plane = i / plane_out;
        const int rem = (int)(i - plane * plane_out);
        const int oy = rem / p.out_w, ox = rem - oy * p.out_w;
        const float* src = x + plane * (int64_t)p.in_h * p.in_w;
        // tap ky touches upsampled row  oy*down - pad0 + ky ; only multiples of `up` hold data
        const int my = oy * p.down_y - p.pad_y0, mx = ox * p.down_x - p.pad_x0;
        int ky0 = (-my) % p.up_y;   // first ky with (my + ky) % up == 0
        if (ky0 < 0) ky0 += p.up_y;
        int kx0 = (-mx) % p.up_x;
        if (kx0 < 0) kx0 += p.up_x;
        float acc = 0.0f;
        for (int ky = ky0; ky < p.kh; ky += p.up_y) {
            const int iy = (my + ky) / p.up_y;          // exact: divisible by construction
            if (iy < 0 || iy >= p.in_h) continue;
            for (int kx = kx0; kx < p.kw; kx += p.up_x) {
                const int ix = (mx + kx) / p.up_x;
                if (ix < 0 || ix >= p.in_w) continue;
                const float prod = src[(int64_t)iy * p.in_w + ix] * s_k[ky * p.kw + kx];
                acc = acc + prod;
            }
        }
        out[i] = acc;
    }
}

// ------------------------------------------------------------------------------ 4x4, up=down=1
constexpr int T_OH = 32, T_OW = 64, T_K = 4;
constexpr int T_IH = T_OH + T_K - 1;        // 35
constexpr int T_IW = T_OW + T_K - 1;        // 67
constexpr int T_LD = 68;                    // LDS row pitch (floats), multiple of 4
// Aligned staging (ALN): when the input rows are 16-byte aligned (in_w % 4 == 0, aligned base) and the window starts two
// columns left of a tile boundary (pad_x0 == 2: the gradient of the up-sampling layers' blur, the blur in front of the
// discriminator's stride-2 convolutions), the window is widened to start FOUR columns left — 72 columns = 18 aligned
// 16-byte groups per row, 630 vector loads per tile instead of 2 380 scalar ones with their index arithmetic; the FIR then
// reads its 8-float window at column lx + 2 (8-byte aligned: four ds_read_b64).
constexpr int A_LD = 72, A_G = A_LD / 4;    // pitch / 16-byte groups per row of the widened window

typedef const float __attribute__((address_space(4)))* cptr_t;

// The sixteen taps are wave-uniform: scalar loads into SGPRs (constant address space) instead of sixteen vector registers
// per lane — with the row-sliding window below the kernels fit 64 registers (eight waves per SIMD).
#define FIR4_LOAD_TAPS(kf, k)                                   \
    float kf[16];                                               \
    {                                                           \
        const cptr_t kc_ = (cptr_t)(uintptr_t)(k);              \
        _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) kf[i_] = kc_[15 - i_];   /* flipped */ \
    }

// 2 x 4 outputs per lane from rows ly .. ly+4 of the staged tile.  Row t is tap row ky = t of output row 0 and ky = t - 1
// of output row 1: walking t upwards visits every accumulator's taps ky-major then kx — the bit-exact order of the
// oracle (multiply and add separate) — with one 8-float window row live at a time.
template <int LD, int COFF>
__device__ __forceinline__ void fir4_rows(const float* s_in, const float (&kf)[16], int lx, int ly, float (&acc)[2][4]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0f;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        float win[8];
        const float* row = s_in + (ly + t) * LD + lx + COFF;
        if (COFF % 4 == 0) {
            const float4 a = *reinterpret_cast<const float4*>(row);
            const float4 b = *reinterpret_cast<const float4*>(row + 4);
            win[0] = a.x; win[1] = a.y; win[2] = a.z; win[3] = a.w;
            win[4] = b.x; win[5] = b.y; win[6] = b.z; win[7] = b.w;
        } else {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float2 a = *reinterpret_cast<const float2*>(row + 2 * h);
                win[2 * h] = a.x;
                win[2 * h + 1] = a.y;
            }
        }
#ifdef SR_FIR_SEP_EXPERIMENT
        // timing experiment: separable taps, fused multiply-adds (8 per output instead of 16 mul + 16 add)
        float hrow[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float h = win[c] * kf[0];
            h = __builtin_fmaf(win[c + 1], kf[1], h);
            h = __builtin_fmaf(win[c + 2], kf[2], h);
            h = __builtin_fmaf(win[c + 3], kf[3], h);
            hrow[c] = h;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ky = t - r;
            if (ky < 0 || ky > 3) continue;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = ky == 0 ? hrow[c] * kf[4] : __builtin_fmaf(hrow[c], kf[4 + ky], acc[r][c]);
        }
#else
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ky = t - r;
            if (ky < 0 || ky > 3) continue;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float prod = win[c + kx] * kf[ky * 4 + kx];
                    acc[r][c] = acc[r][c] + prod;
                }
        }
#endif
    }
}

__device__ __forceinline__ void fir4_store_row(float* q, int ox, int out_w, const float (&v)[4]) {
    if (ox + 3 < out_w && ((reinterpret_cast<uintptr_t>(q) & 15) == 0)) {
        *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (ox + c < out_w) q[c] = v[c];
    }
}

// NBA: the StyledConv tail (reference model.py:26-32: NoiseInjection + FusedLeakyReLU) applied to the
// blurred value before it is stored — y = lrelu((f + nw * noise[b, p]) + bias[c]) * gain, the same
// operation order as k_nba_fwd — which saves one full read + write of the activation per layer.
struct FirNba {
    const float* noise;       // [B or 1, 1, out_h, out_w] or NULL
    const float* noise_w;     // 1 float (device) when noise != NULL
    const float* bias;        // [C] or NULL
    int64_t noise_bstride;
    int channels;
    float alpha, gain;
};

template <bool NBA, bool ALN>
__global__ __launch_bounds__(256) void k_fir4_tile(float* __restrict__ out,
                                                   const float* __restrict__ x,
                                                   const float* __restrict__ k, int in_h, int in_w,
                                                   int out_h, int out_w, int pad_x0, int pad_y0,
                                                   int tiles_x, int tiles_y, FirNba nba) {
    constexpr int LD = ALN ? A_LD : T_LD;
    __shared__ __attribute__((aligned(16))) float s_in[T_IH * LD];
    int bid = blockIdx.x;
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int64_t plane = bid / tiles_y;
    const int oy0 = ty_i * T_OH, ox0 = tx_i * T_OW;
    const int iy0 = oy0 - pad_y0, ix0 = ox0 - pad_x0;
    const float* src = x + plane * (int64_t)in_h * in_w;
    FIR4_LOAD_TAPS(kf, k)

    if (ALN) {
        // (pad_x0 == 2) window columns ox0 - 4 .. ox0 + 67 as 18 aligned groups per row; a group is inside or outside
        constexpr int NG = T_IH * A_G, ST = (NG + 255) / 256;
        float4 stage[ST];
#pragma unroll
        for (int j = 0; j < ST; ++j) {
            const int i = threadIdx.x + 256 * j;
            const int r = i / A_G, g = i - r * A_G;
            const int gy = iy0 + r, gx = ox0 - 4 + 4 * g;
            const bool ok = i < NG && gy >= 0 && gy < in_h && gx >= 0 && gx < in_w;
            const float4 v = *reinterpret_cast<const float4*>(src + (ok ? (int64_t)gy * in_w + gx : 0));
            stage[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < ST; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < NG) reinterpret_cast<float4*>(s_in)[i] = stage[j];
        }
    } else {
        // stage the halo tile: consecutive lanes read consecutive columns of one row (coalesced).  All ten
        // loads of a lane are issued before the first LDS write (unconditional loads from clamped addresses,
        // masked afterwards): one memory round trip per tile instead of ten dependent ones.
        constexpr int T_STAGE = (T_IH * T_LD + 255) / 256;
        float stage[T_STAGE];
#pragma unroll
        for (int j = 0; j < T_STAGE; ++j) {
            const int i = threadIdx.x + 256 * j;
            const int r = i / T_LD, c = i - r * T_LD;
            const int gy = iy0 + r, gx = ix0 + c;
            const bool ok = i < T_IH * T_LD && c < T_IW && gy >= 0 && gy < in_h && gx >= 0 && gx < in_w;
            const float v = src[ok ? (int64_t)gy * in_w + gx : 0];
            stage[j] = ok ? v : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < T_STAGE; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < T_IH * T_LD) s_in[i] = stage[j];
        }
    }
    __syncthreads();

    const int lx = (threadIdx.x & 15) * 4;       // 4 output columns per lane
    const int ly = (threadIdx.x >> 4) * 2;       // 2 output rows per lane
    float acc[2][4];
    fir4_rows<LD, ALN ? 2 : 0>(s_in, kf, lx, ly, acc);

    float* dst = out + plane * (int64_t)out_h * out_w;
    float nw = 0.0f, bb = 0.0f;
    const float* nz = nullptr;
    if (NBA) {
        const int64_t b = plane / nba.channels;
        if (nba.noise) {
            nw = nba.noise_w[0];
            nz = nba.noise + b * nba.noise_bstride;
        }
        if (nba.bias) bb = nba.bias[plane - b * nba.channels];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int oy = oy0 + ly + r;
        if (oy >= out_h) continue;
        const int ox = ox0 + lx;
        if (NBA) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = acc[r][c];
                if (nz && ox + c < out_w) v = v + nw * nz[(int64_t)oy * out_w + ox + c];
                v += bb;
                acc[r][c] = ((v > 0.0f) ? v : v * nba.alpha) * nba.gain;
            }
        }
        fir4_store_row(dst + (int64_t)oy * out_w + ox, ox, out_w, acc[r]);
    }
}

// ------------------------------------------------------------------------------ backward of blur + tail, one pass
// Gradient of  y = lrelu((blur(x) + w * noise) + bias) * gain  (k_fir4_tile<true>) w.r.t. x, with the reductions of
// the tail's backward, in ONE pass over the two full-size tensors it needs (gy, y) — the two-kernel form
// (k_nba_bwd: read gy, y, write gpre;  k_fir4_tile<false>: read gpre, write gx) moves five tensors, this one three.
// The halo tile of gpre = lrelu'(y) * gy * gain is computed while it is staged (same expression as k_nba_bwd, so gx
// is bit-identical to the two-kernel form); every gpre element belongs to exactly one tile's "owned" 32 x 64 corner
// of its 35 x 67 window, where it enters the three sums
//     sum gpre (bias gradient), sum gpre * noise (noise strength), sum gpre * y0 (demodulation row-dot; y0 = the
//     pre-activation value of the forward pass rebuilt from y, see k_nba_bwd<true>)
// as one partial triple per workgroup; the finish kernels of csrc/fused_elem.hip add them in a fixed order.
struct FirNbaBwd {
    const float* fw;          // forward output y [planes, in_h, in_w]
    const float* noise;       // [B or 1, 1, in_h, in_w] or NULL
    const float* noise_w;     // 1 float (device) when noise != NULL
    const float* bias;        // [C] or NULL
    float* partial;           // [planes * tiles] pairs (sum gpre, sum gpre * noise)
    float* dot_partial;       // [planes * tiles]
    int64_t noise_bstride;
    int channels;
    float alpha, gain, inv_pos, inv_neg;
};

struct FirSums {
    float b, n, d;
};

__device__ __forceinline__ float fir4_gpre(float g, float o, float nzv, bool ok, bool owned, const FirNbaBwd& q, float nw,
                                           float bb, FirSums& s) {
    const float v = ok ? ((o > 0.0f) ? g : g * q.alpha) * q.gain : 0.0f;
    if (owned) {
        const float y0 = ((o > 0.0f) ? o * q.inv_pos : o * q.inv_neg) - nw * nzv - bb;
        s.b += v;
        s.n += v * nzv;
        s.d += v * y0;
    }
    return v;
}

template <bool ALN>
__global__ __launch_bounds__(256) void k_fir4_nba_bwd(float* __restrict__ out, const float* __restrict__ gy,
                                                      const float* __restrict__ k, int in_h, int in_w, int out_h,
                                                      int out_w, int pad_x0, int pad_y0, int tiles_x, int tiles_y,
                                                      FirNbaBwd q) {
    constexpr int LD = ALN ? A_LD : T_LD;
    __shared__ __attribute__((aligned(16))) float s_in[T_IH * LD];
    __shared__ float s_red[12];
    int bid = blockIdx.x;
    const int tile = bid % (tiles_x * tiles_y);
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int64_t plane = bid / tiles_y;
    const int oy0 = ty_i * T_OH, ox0 = tx_i * T_OW;
    const int iy0 = oy0 - pad_y0, ix0 = ox0 - pad_x0;
    const float* src = gy + plane * (int64_t)in_h * in_w;
    const float* fws = q.fw + plane * (int64_t)in_h * in_w;
    const int64_t b = plane / q.channels;
    const float* nz = q.noise ? q.noise + b * q.noise_bstride : nullptr;
    const float nw = q.noise ? q.noise_w[0] : 0.0f;
    const float bb = q.bias ? q.bias[plane - b * q.channels] : 0.0f;
    FIR4_LOAD_TAPS(kf, k)

    FirSums sums{0.0f, 0.0f, 0.0f};
    if (ALN) {
        // window columns ox0 - 4 .. ox0 + 67; owned = rows 0..31, window columns 2..65 (image columns ox0 - 2 .. ox0 + 61)
        constexpr int NG = T_IH * A_G, ST = (NG + 255) / 256;
        float4 sg[ST], so[ST], sn[ST];
#pragma unroll
        for (int j = 0; j < ST; ++j) {
            const int i = threadIdx.x + 256 * j;
            const int r = i / A_G, g = i - r * A_G;
            const int y = iy0 + r, x = ox0 - 4 + 4 * g;
            const bool ok = i < NG && y >= 0 && y < in_h && x >= 0 && x < in_w;
            const int64_t o = ok ? (int64_t)y * in_w + x : 0;
            sg[j] = *reinterpret_cast<const float4*>(src + o);
            so[j] = *reinterpret_cast<const float4*>(fws + o);
            sn[j] = nz ? *reinterpret_cast<const float4*>(nz + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < ST; ++j) {
            const int i = threadIdx.x + 256 * j;
            const int r = i / A_G, g = i - r * A_G;
            const int y = iy0 + r, x = ox0 - 4 + 4 * g;
            const bool ok = i < NG && y >= 0 && y < in_h && x >= 0 && x < in_w;
            const bool own_r = ok && r < T_OH;
            const int c0 = 4 * g;
            float4 v;
            v.x = fir4_gpre(sg[j].x, so[j].x, sn[j].x, ok, own_r && c0 >= 2 && c0 < 66, q, nw, bb, sums);
            v.y = fir4_gpre(sg[j].y, so[j].y, sn[j].y, ok, own_r && c0 + 1 >= 2 && c0 + 1 < 66, q, nw, bb, sums);
            v.z = fir4_gpre(sg[j].z, so[j].z, sn[j].z, ok, own_r && c0 + 2 >= 2 && c0 + 2 < 66, q, nw, bb, sums);
            v.w = fir4_gpre(sg[j].w, so[j].w, sn[j].w, ok, own_r && c0 + 3 >= 2 && c0 + 3 < 66, q, nw, bb, sums);
            if (i < NG) reinterpret_cast<float4*>(s_in)[i] = v;
        }
    } else {
        constexpr int T_STAGE = (T_IH * T_LD + 255) / 256;
        float sg[T_STAGE], so[T_STAGE], sn[T_STAGE];
#pragma unroll
        for (int j = 0; j < T_STAGE; ++j) {
            const int i = threadIdx.x + 256 * j;
            const int r = i / T_LD, c = i - r * T_LD;
            const int y = iy0 + r, x = ix0 + c;
            const bool ok = i < T_IH * T_LD && c < T_IW && y >= 0 && y < in_h && x >= 0 && x < in_w;
            const int64_t o = ok ? (int64_t)y * in_w + x : 0;
            sg[j] = src[o];
            so[j] = fws[o];
            sn[j] = nz ? nz[o] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < T_STAGE; ++j) {
            const int i = threadIdx.x + 256 * j;
            const int r = i / T_LD, c = i - r * T_LD;
            const int y = iy0 + r, x = ix0 + c;
            const bool ok = i < T_IH * T_LD && c < T_IW && y >= 0 && y < in_h && x >= 0 && x < in_w;
            const float v = fir4_gpre(sg[j], so[j], sn[j], ok, ok && r < T_OH && c < T_OW, q, nw, bb, sums);
            if (i < T_IH * T_LD) s_in[i] = v;
        }
    }
    sums.b = sr_wave_sum(sums.b);
    sums.n = sr_wave_sum(sums.n);
    sums.d = sr_wave_sum(sums.d);
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) { s_red[wave] = sums.b; s_red[4 + wave] = sums.n; s_red[8 + wave] = sums.d; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t slot = plane * (int64_t)(tiles_x * tiles_y) + tile;
        q.partial[slot * 2] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        q.partial[slot * 2 + 1] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
        q.dot_partial[slot] = (s_red[8] + s_red[9]) + (s_red[10] + s_red[11]);
    }

    const int lx = (threadIdx.x & 15) * 4, ly = (threadIdx.x >> 4) * 2;
    float acc[2][4];
    fir4_rows<LD, ALN ? 2 : 0>(s_in, kf, lx, ly, acc);
    float* dst = out + plane * (int64_t)out_h * out_w;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int oy = oy0 + ly + r;
        if (oy >= out_h) continue;
        const int ox = ox0 + lx;
        fir4_store_row(dst + (int64_t)oy * out_w + ox, ox, out_w, acc[r]);
    }
}

// ------------------------------------------------------------------------------ 4x4 taps, up = 2 or down = 2
// The two resampling shapes of the networks: down = 2 (the blur in front of ResBlock.skip's 1x1 stride-2
// convolution evaluated only at the pixels that convolution reads, and the gradient of the ToRGB skip
// up-sampling) and up = 2 (the ToRGB skip up-sampling and the gradient of the former).  Same scheme as
// k_fir4_tile: the input window of a 32 x 32 output tile is staged once in LDS with coalesced row loads, each
// lane produces four consecutive outputs of one row.  Taps are visited ky-major then kx, multiply and add
// separate (-ffp-contract=off), inserted zeros are skipped (not multiplied): the order of k_upfirdn_generic and
// of the numpy oracle, so all three agree bit for bit.
constexpr int R_OH = 32, R_OW = 32;

template <int UP, int DOWN>
__global__ __launch_bounds__(256) void k_fir4_resample(float* __restrict__ out, const float* __restrict__ x,
                                                       const float* __restrict__ k, int in_h, int in_w,
                                                       int out_h, int out_w, int pad_x0, int pad_y0,
                                                       int tiles_x, int tiles_y,
                                                       const float* __restrict__ addend) {
    static_assert((UP == 1 && DOWN == 2) || (UP == 2 && DOWN == 1), "one of the two resampling shapes");
    // input rows / columns a tile can touch: (R - 1) * DOWN + 4 taps, every UP-th of them holds data
    constexpr int SPAN_H = ((R_OH - 1) * DOWN + 4 + UP - 1) / UP + 1;
    constexpr int SPAN_W = ((R_OW - 1) * DOWN + 4 + UP - 1) / UP + 1;
    constexpr int LDW = SPAN_W | 1;                                   // odd pitch: rows on different banks
    __shared__ float s_in[SPAN_H * LDW];
    __shared__ float s_k[16];
    int bid = blockIdx.x;
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int64_t plane = bid / tiles_y;
    const int oy0 = ty_i * R_OH, ox0 = tx_i * R_OW;
    // first upsampled coordinate of the tile, and the first input row / column at or after it
    const int my0 = oy0 * DOWN - pad_y0, mx0 = ox0 * DOWN - pad_x0;
    const int iy_lo = floor_div(my0 + UP - 1, UP), ix_lo = floor_div(mx0 + UP - 1, UP);
    const float* src = x + plane * (int64_t)in_h * in_w;
    if (threadIdx.x < 16) {
        const int ky = threadIdx.x >> 2, kx = threadIdx.x & 3;
        s_k[threadIdx.x] = k[(3 - ky) * 4 + (3 - kx)];
    }
    for (int i = threadIdx.x; i < SPAN_H * SPAN_W; i += 256) {
        const int r = i / SPAN_W, c = i - r * SPAN_W;
        const int gy = iy_lo + r, gx = ix_lo + c;
        const bool ok = gy >= 0 && gy < in_h && gx >= 0 && gx < in_w;
        s_in[r * LDW + c] = ok ? src[(int64_t)gy * in_w + gx] : 0.0f;
    }
    __syncthreads();
    float kf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = s_k[i];
    const int ly = threadIdx.x >> 3, lx = (threadIdx.x & 7) * 4;
    const int oy = oy0 + ly;
    if (oy >= out_h) return;
    const int my = oy * DOWN - pad_y0;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (UP == 2) {
        // Same taps in the same order as the loop below, without its 64 parity tests: of the four tap rows the two with
        // (my + ky) even hold data (ky = ky0, ky0 + 2), and since ox0 + lx is a multiple of four the column parity is the
        // same for every lane — output c reads the taps kx = ((c + PX) & 1), + 2 at compile-time offsets from one base
        // column.  Rows / columns outside the map are zeros in the staged window: adding their 0 * tap leaves the sum as
        // skipping them does.
        const int ky0 = my & 1;
        const int mx = ox0 + lx - pad_x0;
        const int px = mx & 1;
        const int cbase = ((mx - px) >> 1) - ix_lo;
        auto rows = [&](auto pxc) {
            constexpr int PX = decltype(pxc)::value;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ky = ky0 + 2 * j;
                const float* row = s_in + (((my + ky) >> 1) - iy_lo) * LDW + cbase;
                const float* kr = s_k + ky * 4;
                const float k0 = kr[0], k1 = kr[1], k2 = kr[2], k3 = kr[3];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int kx = ((c + PX) & 1) + 2 * i;
                        const float kv = kx == 0 ? k0 : (kx == 1 ? k1 : (kx == 2 ? k2 : k3));
                        const float prod = row[(PX + c + kx) >> 1] * kv;
                        acc[c] = acc[c] + prod;
                    }
            }
        };
        if (px) rows(std::integral_constant<int, 1>{});
        else rows(std::integral_constant<int, 0>{});
    } else
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        const int m = my + ky;
        if (UP == 2 && (m & 1)) continue;                               // an inserted zero row
        const int iy = (UP == 2 ? (m >> 1) : m);                        // arithmetic shift: exact for even m
        if (iy < 0 || iy >= in_h) continue;
        const float* row = s_in + (iy - iy_lo) * LDW;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int mxc = (ox0 + lx + c) * DOWN - pad_x0 + kx;
                if (UP == 2 && (mxc & 1)) continue;
                const int ix = (UP == 2 ? (mxc >> 1) : mxc);
                if (ix < 0 || ix >= in_w) continue;
                const float prod = row[ix - ix_lo] * kf[ky * 4 + kx];
                acc[c] = acc[c] + prod;
            }
    }
    const int64_t o = plane * (int64_t)out_h * out_w + (int64_t)oy * out_w + ox0 + lx;
    float* q = out + o;
    // optional second operand of a following addition (ToRGB: rgb + upsample(skip), reference model.py:66-68)
    // rows of 4-float groups on 16-byte addresses (every map of the networks): one 16-byte access per lane instead of
    // four 4-byte ones at a 16-byte stride — the 256^2 gradient of the discriminator's skip branch is 400 MB of traffic
    const bool vec = (out_w & 3) == 0 && ox0 + lx + 3 < out_w && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                     (!addend || (reinterpret_cast<uintptr_t>(addend) & 15) == 0);
    if (vec) {
        float4 r = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (addend) {
            const float4 a = *reinterpret_cast<const float4*>(addend + o);
            r.x = r.x + a.x; r.y = r.y + a.y; r.z = r.z + a.z; r.w = r.w + a.w;
        }
        *reinterpret_cast<float4*>(q) = r;
        return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (ox0 + lx + c < out_w) q[c] = addend ? acc[c] + addend[o + c] : acc[c];
}

// aligned staging of k_fir4_tile / k_fir4_nba_bwd (see A_LD); SR_FIR_ALIGNED=0 keeps the scalar staging (A/B)
inline bool fir4_aligned(const float* x, int in_w, int pad_x0) {
    static const bool enabled = [] {
        const char* e = std::getenv("SR_FIR_ALIGNED");
        return !(e && e[0] == '0');
    }();
    return enabled && pad_x0 == 2 && in_w % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
}

}  // namespace

extern "C" int sr_upfirdn2d(float* out, const float* x, const float* k, int64_t major, int in_h,
                            int in_w, int out_h, int out_w, int kh, int kw, int up_x, int up_y,
                            int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                            sr_stream_t stream) {
    if (major < 0 || in_h < 0 || in_w < 0 || kh <= 0 || kw <= 0 || up_x <= 0 || up_y <= 0 ||
        down_x <= 0 || down_y <= 0)
        return SR_EINVAL;
    // floor division like Python's // (reference op/upfirdn2d.py:103-104)
    auto fdiv = [](int a, int b) { int q = a / b; return (a % b != 0 && (a < 0)) ? q - 1 : q; };
    const int eh = fdiv(in_h * up_y + pad_y0 + pad_y1 - kh, down_y) + 1;
    const int ew = fdiv(in_w * up_x + pad_x0 + pad_x1 - kw, down_x) + 1;
    if (eh != out_h || ew != out_w) return SR_EINVAL;
    if (major == 0 || out_h <= 0 || out_w <= 0) return SR_OK;
    if (!out || !x || !k) return SR_EINVAL;
    hipStream_t st = sr_stream(stream);
    if (up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kh == 4 && kw == 4) {
        const int tiles_x = (out_w + T_OW - 1) / T_OW, tiles_y = (out_h + T_OH - 1) / T_OH;
        const int64_t blocks = (int64_t)tiles_x * tiles_y * major;
        if (blocks < 0x7FFFFFFFLL) {
            if (fir4_aligned(x, in_w, pad_x0))
                hipLaunchKernelGGL((k_fir4_tile<false, true>), dim3((unsigned)blocks), dim3(256), 0, st, out, x, k, in_h,
                                   in_w, out_h, out_w, pad_x0, pad_y0, tiles_x, tiles_y, FirNba{});
            else
                hipLaunchKernelGGL((k_fir4_tile<false, false>), dim3((unsigned)blocks), dim3(256), 0, st, out, x, k, in_h,
                                   in_w, out_h, out_w, pad_x0, pad_y0, tiles_x, tiles_y, FirNba{});
            return sr_launch_status();
        }
    }
    if (kh == 4 && kw == 4 && up_x == up_y && down_x == down_y &&
        ((up_x == 1 && down_x == 2) || (up_x == 2 && down_x == 1))) {
        const int tiles_x = (out_w + R_OW - 1) / R_OW, tiles_y = (out_h + R_OH - 1) / R_OH;
        const int64_t blocks = (int64_t)tiles_x * tiles_y * major;
        if (blocks < 0x7FFFFFFFLL) {
            if (down_x == 2)
                hipLaunchKernelGGL((k_fir4_resample<1, 2>), dim3((unsigned)blocks), dim3(256), 0, st, out, x, k, in_h,
                                   in_w, out_h, out_w, pad_x0, pad_y0, tiles_x, tiles_y, (const float*)nullptr);
            else
                hipLaunchKernelGGL((k_fir4_resample<2, 1>), dim3((unsigned)blocks), dim3(256), 0, st, out, x, k, in_h,
                                   in_w, out_h, out_w, pad_x0, pad_y0, tiles_x, tiles_y, (const float*)nullptr);
            return sr_launch_status();
        }
    }
    UfdParams p{in_h, in_w, out_h, out_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0};
    const int64_t total = major * (int64_t)out_h * out_w;
    hipLaunchKernelGGL(k_upfirdn_generic, dim3(sr_stream_grid(total, 256)), dim3(256),
                       (size_t)kh * kw * sizeof(float), st, out, x, k, p, total);
    return sr_launch_status();
}

extern "C" int sr_upsample2_add(float* out, const float* x, const float* k, const float* addend, int64_t major,
                                int in_h, int in_w, int out_h, int out_w, int pad0, int pad1, sr_stream_t stream) {
    if (major < 0 || in_h < 0 || in_w < 0) return SR_EINVAL;
    if (in_h * 2 + pad0 + pad1 - 4 + 1 != out_h || in_w * 2 + pad0 + pad1 - 4 + 1 != out_w) return SR_EINVAL;
    if (major == 0 || out_h <= 0 || out_w <= 0) return SR_OK;
    if (!out || !x || !k || !addend) return SR_EINVAL;
    const int tiles_x = (out_w + R_OW - 1) / R_OW, tiles_y = (out_h + R_OH - 1) / R_OH;
    const int64_t blocks = (int64_t)tiles_x * tiles_y * major;
    if (blocks >= 0x7FFFFFFFLL) return SR_ERANGE;
    hipLaunchKernelGGL((k_fir4_resample<2, 1>), dim3((unsigned)blocks), dim3(256), 0, sr_stream(stream), out, x, k, in_h,
                       in_w, out_h, out_w, pad0, pad0, tiles_x, tiles_y, addend);
    return sr_launch_status();
}

extern "C" int sr_blur_noise_bias_act(float* y, const float* x, const float* k, const float* noise,
                                      const float* noise_w, const float* bias, float alpha, float gain, int64_t n,
                                      int64_t c, int in_h, int in_w, int out_h, int out_w, int pad0, int pad1,
                                      int64_t noise_bstride, sr_stream_t stream) {
    if (n < 0 || c < 0 || in_h < 0 || in_w < 0) return SR_EINVAL;
    if (in_h + pad0 + pad1 - 4 + 1 != out_h || in_w + pad0 + pad1 - 4 + 1 != out_w) return SR_EINVAL;
    if (n * c == 0 || out_h <= 0 || out_w <= 0) return SR_OK;
    if (!y || !x || !k || (noise && !noise_w)) return SR_EINVAL;
    const int tiles_x = (out_w + T_OW - 1) / T_OW, tiles_y = (out_h + T_OH - 1) / T_OH;
    const int64_t blocks = (int64_t)tiles_x * tiles_y * n * c;
    if (blocks >= 0x7FFFFFFFLL) return SR_ERANGE;
    const FirNba nba{noise, noise_w, bias, noise_bstride, (int)c, alpha, gain};
    if (fir4_aligned(x, in_w, pad0))
        hipLaunchKernelGGL((k_fir4_tile<true, true>), dim3((unsigned)blocks), dim3(256), 0, sr_stream(stream), y, x, k,
                           in_h, in_w, out_h, out_w, pad0, pad0, tiles_x, tiles_y, nba);
    else
        hipLaunchKernelGGL((k_fir4_tile<true, false>), dim3((unsigned)blocks), dim3(256), 0, sr_stream(stream), y, x, k,
                           in_h, in_w, out_h, out_w, pad0, pad0, tiles_x, tiles_y, nba);
    return sr_launch_status();
}

// finish kernels of csrc/fused_elem.hip (bias / noise-strength / row-dot partials -> results, fixed order)
int sr_nba_finish_launch(float* gbias, float* gnoise_w, float* rowdot, const float* partial, const float* dot_partial,
                         float* chan_nw, int64_t n, int64_t c, int chunks, bool has_noise, hipStream_t st);

extern "C" int64_t sr_blur_nba_bwd_scratch_floats(int64_t n, int64_t c, int out_h, int out_w) {
    if (n <= 0 || c <= 0 || out_h <= 0 || out_w <= 0) return 3;
    const int64_t tiles = (int64_t)((out_w + T_OW - 1) / T_OW) * ((out_h + T_OH - 1) / T_OH);
    return 3 * n * c * tiles + c + 2;
}

// gx [n, c, out_h, out_w] = blur^T( lrelu'(y) * gy * gain ) with `k` the FORWARD blur kernel and pad0 / pad1 the
// forward's padding (the gradient pads by 3 - pad0 in front), plus gbias [c], gnoise_w [1] (both optional: NULL = the
// parameters are frozen) and rowdot [n * c] (see FirNbaBwd).  gy, y: [n, c, in_h, in_w], the forward's OUTPUT extent.
extern "C" int sr_blur_nba_bwd(float* gx, float* gbias, float* gnoise_w, float* rowdot, const float* gy, const float* y,
                               const float* k_flipped, const float* noise, const float* noise_w, const float* bias,
                               float alpha, float gain, int64_t n, int64_t c, int in_h, int in_w, int out_h, int out_w,
                               int pad0, int64_t noise_bstride, float* scratch, sr_stream_t stream) {
    if (n < 0 || c < 0 || in_h < 0 || in_w < 0) return SR_EINVAL;
    // forward: in = out + 2 * pad - 3 with pad0 == pad1 (the up-sampling layers: 257 -> 256, pad 1)
    if (out_h + 2 * pad0 - 3 != in_h || out_w + 2 * pad0 - 3 != in_w || pad0 < 0 || pad0 > 3) return SR_EINVAL;
    if (n * c == 0 || in_h == 0 || in_w == 0) return SR_OK;
    if (!gx || !rowdot || !gy || !y || !k_flipped || !scratch || (noise && !noise_w) || gain == 0.0f || alpha == 0.0f)
        return SR_EINVAL;
    const int tiles_x = (out_w + T_OW - 1) / T_OW, tiles_y = (out_h + T_OH - 1) / T_OH;
    const int chunks = tiles_x * tiles_y;
    const int64_t blocks = (int64_t)chunks * n * c;
    if (blocks >= 0x7FFFFFFFLL) return SR_ERANGE;
    // every gpre element must lie in some tile's owned corner: the tiles start at -gpad and cover chunks * 32 / 64
    const int gpad = 3 - pad0;
    if (tiles_y * T_OH - gpad < in_h || tiles_x * T_OW - gpad < in_w) return SR_EINVAL;
    hipStream_t st = sr_stream(stream);
    float* dot_partial = scratch + 2 * n * c * (int64_t)chunks;
    float* chan_nw = dot_partial + n * c * (int64_t)chunks;
    const FirNbaBwd q{y, noise, noise_w, bias, scratch, chunks == 1 ? rowdot : dot_partial, noise_bstride, (int)c, alpha,
                      gain, 1.0f / gain, 1.0f / (alpha * gain)};
    const bool aln = fir4_aligned(gy, in_w, gpad) && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
                     (!noise || ((reinterpret_cast<uintptr_t>(noise) & 15) == 0 && noise_bstride % 4 == 0));
    if (aln)
        hipLaunchKernelGGL(k_fir4_nba_bwd<true>, dim3((unsigned)blocks), dim3(256), 0, st, gx, gy, k_flipped, in_h, in_w,
                           out_h, out_w, gpad, gpad, tiles_x, tiles_y, q);
    else
        hipLaunchKernelGGL(k_fir4_nba_bwd<false>, dim3((unsigned)blocks), dim3(256), 0, st, gx, gy, k_flipped, in_h, in_w,
                           out_h, out_w, gpad, gpad, tiles_x, tiles_y, q);
    const int rc = sr_launch_status();
    if (rc != SR_OK) return rc;
    return sr_nba_finish_launch(gbias, gnoise_w, chunks == 1 ? nullptr : rowdot, scratch, dot_partial, chan_nw, n, c,
                                chunks, noise != nullptr, st);
}
