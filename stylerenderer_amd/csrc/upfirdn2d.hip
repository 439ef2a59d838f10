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

template <bool NBA>
__global__ __launch_bounds__(256) void k_fir4_tile(float* __restrict__ out,
                                                   const float* __restrict__ x,
                                                   const float* __restrict__ k, int in_h, int in_w,
                                                   int out_h, int out_w, int pad_x0, int pad_y0,
                                                   int tiles_x, int tiles_y, FirNba nba) {
    __shared__ __attribute__((aligned(16))) float s_in[T_IH * T_LD];
    __shared__ float s_k[16];
    int bid = blockIdx.x;
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int64_t plane = bid / tiles_y;
    const int oy0 = ty_i * T_OH, ox0 = tx_i * T_OW;
    const int iy0 = oy0 - pad_y0, ix0 = ox0 - pad_x0;
    const float* src = x + plane * (int64_t)in_h * in_w;

    if (threadIdx.x < 16) {
        const int ky = threadIdx.x >> 2, kx = threadIdx.x & 3;
        s_k[threadIdx.x] = k[(3 - ky) * 4 + (3 - kx)];
    }
    // stage the halo tile: consecutive lanes read consecutive columns of one row (coalesced).  All ten
    // loads of a lane are issued before the first LDS write (unconditional loads from clamped addresses,
    // masked afterwards): one memory round trip per tile instead of ten dependent ones.
    constexpr int T_STAGE = (T_IH * T_LD + 255) / 256;
    float stage[T_STAGE];
#pragma unroll
    for (int k = 0; k < T_STAGE; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int r = i / T_LD, c = i - r * T_LD;
        const int gy = iy0 + r, gx = ix0 + c;
        const bool ok = i < T_IH * T_LD && c < T_IW && gy >= 0 && gy < in_h && gx >= 0 && gx < in_w;
        const float v = src[ok ? (int64_t)gy * in_w + gx : 0];
        stage[k] = ok ? v : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < T_STAGE; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < T_IH * T_LD) s_in[i] = stage[k];
    }
    __syncthreads();

    float kf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = s_k[i];

    const int lx = (threadIdx.x & 15) * 4;       // 4 output columns per lane
    const int ly = (threadIdx.x >> 4) * 2;       // 2 output rows per lane
    float acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0f;

    // rows ly .. ly+4 of the tile feed the two output rows; for bit-exact ky-major order each
    // output row walks its own four input rows top to bottom.
    float win[5][8];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const float4 a = *reinterpret_cast<const float4*>(&s_in[(ly + r) * T_LD + lx]);
        const float4 b = *reinterpret_cast<const float4*>(&s_in[(ly + r) * T_LD + lx + 4]);
        win[r][0] = a.x; win[r][1] = a.y; win[r][2] = a.z; win[r][3] = a.w;
        win[r][4] = b.x; win[r][5] = b.y; win[r][6] = b.z; win[r][7] = b.w;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float prod = win[r + ky][c + kx] * kf[ky * 4 + kx];
                    acc[r][c] = acc[r][c] + prod;
                }

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
        float* q = dst + (int64_t)oy * out_w + ox;
        if (ox + 3 < out_w && ((reinterpret_cast<uintptr_t>(q) & 15) == 0)) {
            *reinterpret_cast<float4*>(q) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (ox + c < out_w) q[c] = acc[r][c];
        }
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
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (ox0 + lx + c < out_w) q[c] = addend ? acc[c] + addend[o + c] : acc[c];
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
            hipLaunchKernelGGL(k_fir4_tile<false>, dim3((unsigned)blocks), dim3(256), 0, st, out, x, k, in_h, in_w,
                               out_h, out_w, pad_x0, pad_y0, tiles_x, tiles_y, FirNba{});
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
    hipLaunchKernelGGL(k_fir4_tile<true>, dim3((unsigned)blocks), dim3(256), 0, sr_stream(stream), y, x, k, in_h,
                       in_w, out_h, out_w, pad0, pad0, tiles_x, tiles_y, nba);
    return sr_launch_status();
}
