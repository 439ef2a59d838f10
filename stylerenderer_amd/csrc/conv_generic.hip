// Generic-geometry convolution on the device: any kernel extent, stride and zero padding (dilation 1, one group).
//
// The networks of the benchmarked path use four geometries (3x3 s1 p1, 3x3 s2 p0, 1x1 s1, 1x1 s2 and the transposed
// 3x3 s2), which csrc/conv_mfma.hip / conv_wino.hip run on the matrix cores.  The reference's layers accept ANY
// geometry (EqualConv2d: reference layers.py:204-221; ModulatedConv2d with an arbitrary kernel_size, also up- and
// down-sampling: layers.py:259-323): those calls land here instead of on MIOpen, so that a device tensor never leaves
// this library's kernels.  Three plain direct kernels in the reference's own weight layout [N, C, KH, KW] — they are the
// complete set under differentiation (each one's two gradients are the other two), which is how
// op/conv_generic.py makes them differentiable to any order:
//     k_gconv_fwd     y[b,n,oy,ox]   = sum_{c,ky,kx} w[n,c,ky,kx] * x[b,c,oy*sy-py+ky, ox*sx-px+kx]
//     k_gconv_dgrad   dx[b,c,iy,ix]  = sum_{n,ky,kx} w[n,c,ky,kx] * g[b,n,oy,ox]      (oy*sy-py+ky == iy, ...)
//     k_gconv_wgrad   dw[n,c,ky,kx]  = sum_{b,oy,ox} g[b,n,oy,ox] * x[b,c,oy*sy-py+ky, ox*sx-px+kx]
// Bound: vector ALU / L2 (one fma per operand pair, no data reuse in registers beyond the weight broadcast).  They are
// the correctness path of shapes no network of SURVEY §8 produces, not a tuned one: fixed summation order
// (deterministic), fp32 fma chain, |err| <= 2e-6 * sum|a*b| vs float64 like the direct MFMA kernels.
#include "common.h"

namespace {

struct GConv {
    int B, C, N, IH, IW, OH, OW, KH, KW, sy, sx, py, px;
};

// grid (pixel blocks, (b * N + n) folded over y and z): the weights of a workgroup are wave-uniform (scalar loads), lanes
// walk output columns
__global__ __launch_bounds__(256) void k_gconv_fwd(float* __restrict__ y, const float* __restrict__ x,
                                                   const float* __restrict__ w, const float* __restrict__ bias, GConv p) {
    const int64_t bn64 = (int64_t)blockIdx.z * gridDim.y + blockIdx.y;
    if (bn64 >= (int64_t)p.B * p.N) return;
    const int bn = (int)bn64, b = bn / p.N, n = bn - b * p.N;
    const int64_t opix = (int64_t)p.OH * p.OW;
    const float* wn = w + (int64_t)n * p.C * p.KH * p.KW;
    const float* xb = x + (int64_t)b * p.C * p.IH * p.IW;
    const float bb = bias ? bias[n] : 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < opix; i += (int64_t)gridDim.x * blockDim.x) {
        const int oy = (int)(i / p.OW), ox = (int)(i - (int64_t)oy * p.OW);
        const int iy0 = oy * p.sy - p.py, ix0 = ox * p.sx - p.px;
        float acc = 0.0f;
        for (int c = 0; c < p.C; ++c) {
            const float* xc = xb + (int64_t)c * p.IH * p.IW;
            const float* wc = wn + (int64_t)c * p.KH * p.KW;
            for (int ky = 0; ky < p.KH; ++ky) {
                const int iy = iy0 + ky;
                if (iy < 0 || iy >= p.IH) continue;
                for (int kx = 0; kx < p.KW; ++kx) {
                    const int ix = ix0 + kx;
                    if (ix < 0 || ix >= p.IW) continue;
                    acc = fmaf(wc[ky * p.KW + kx], xc[(int64_t)iy * p.IW + ix], acc);
                }
            }
        }
        y[(int64_t)bn * opix + i] = acc + bb;
    }
}

// grid (pixel blocks, b * C + c): input pixel (iy, ix) receives tap (ky, kx) of output (oy, ox) when
// oy * sy - py + ky == iy  and  ox * sx - px + kx == ix
__global__ __launch_bounds__(256) void k_gconv_dgrad(float* __restrict__ dx, const float* __restrict__ g,
                                                     const float* __restrict__ w, GConv p) {
    const int64_t bc64 = (int64_t)blockIdx.z * gridDim.y + blockIdx.y;
    if (bc64 >= (int64_t)p.B * p.C) return;
    const int bc = (int)bc64, b = bc / p.C, c = bc - b * p.C;
    const int64_t ipix = (int64_t)p.IH * p.IW, opix = (int64_t)p.OH * p.OW;
    const int taps = p.KH * p.KW;
    const float* gb = g + (int64_t)b * p.N * opix;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ipix; i += (int64_t)gridDim.x * blockDim.x) {
        const int iy = (int)(i / p.IW), ix = (int)(i - (int64_t)iy * p.IW);
        float acc = 0.0f;
        for (int n = 0; n < p.N; ++n) {
            const float* gn = gb + (int64_t)n * opix;
            const float* wn = w + ((int64_t)n * p.C + c) * taps;
            for (int ky = 0; ky < p.KH; ++ky) {
                const int ty = iy + p.py - ky;
                if (ty < 0 || ty % p.sy != 0) continue;
                const int oy = ty / p.sy;
                if (oy >= p.OH) continue;
                for (int kx = 0; kx < p.KW; ++kx) {
                    const int tx = ix + p.px - kx;
                    if (tx < 0 || tx % p.sx != 0) continue;
                    const int ox = tx / p.sx;
                    if (ox >= p.OW) continue;
                    acc = fmaf(wn[ky * p.KW + kx], gn[(int64_t)oy * p.OW + ox], acc);
                }
            }
        }
        dx[(int64_t)bc * ipix + i] = acc;
    }
}

// one workgroup per weight element: lanes walk (b, oy, ox) with a fixed stride, then a fixed-order tree
__global__ __launch_bounds__(256) void k_gconv_wgrad(float* __restrict__ dw, const float* __restrict__ x,
                                                     const float* __restrict__ g, GConv p) {
    __shared__ float red[4];
    int e = blockIdx.x;
    const int kx = e % p.KW;
    e /= p.KW;
    const int ky = e % p.KH;
    e /= p.KH;
    const int c = e % p.C, n = e / p.C;
    const int64_t opix = (int64_t)p.OH * p.OW, total = (int64_t)p.B * opix;
    float acc = 0.0f;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const int b = (int)(i / opix);
        const int64_t q = i - (int64_t)b * opix;
        const int oy = (int)(q / p.OW), ox = (int)(q - (int64_t)oy * p.OW);
        const int iy = oy * p.sy - p.py + ky, ix = ox * p.sx - p.px + kx;
        if (iy < 0 || iy >= p.IH || ix < 0 || ix >= p.IW) continue;
        acc = fmaf(g[((int64_t)b * p.N + n) * opix + q], x[(((int64_t)b * p.C + c) * p.IH + iy) * p.IW + ix], acc);
    }
    acc = sr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) dw[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

bool geometry_ok(const GConv& p) {
    if (p.B < 0 || p.C <= 0 || p.N <= 0 || p.IH <= 0 || p.IW <= 0 || p.KH <= 0 || p.KW <= 0 || p.sy <= 0 || p.sx <= 0 ||
        p.py < 0 || p.px < 0)
        return false;
    if (p.IH + 2 * p.py < p.KH || p.IW + 2 * p.px < p.KW) return false;
    return p.OH == (p.IH + 2 * p.py - p.KH) / p.sy + 1 && p.OW == (p.IW + 2 * p.px - p.KW) / p.sx + 1;
}

GConv make(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW, int64_t OH, int64_t OW, int kh, int kw, int sy, int sx,
           int py, int px) {
    return GConv{(int)B, (int)C, (int)N, (int)IH, (int)IW, (int)OH, (int)OW, kh, kw, sy, sx, py, px};
}

bool fits(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW, int64_t OH, int64_t OW, int kh, int kw) {
    return B * C < (1LL << 31) && B * N < (1LL << 31) && IH * IW < (1LL << 31) && OH * OW < (1LL << 31) && C < (1 << 24) &&
           N < (1 << 24) && N * C * kh * kw < (1LL << 31);
}

// rows = b * channels of the output: folded over grid.y (<= 65535) and grid.z
dim3 row_grid(int64_t pix, int64_t rows) {
    const int64_t gx = sr_ceil_div(pix, 256) < 1024 ? sr_ceil_div(pix, 256) : 1024;
    const int64_t gy = rows < 65535 ? rows : 65535;
    return dim3((unsigned)gx, (unsigned)gy, (unsigned)sr_ceil_div(rows, gy));
}

}  // namespace

extern "C" int sr_conv2d_generic(float* y, const float* x, const float* w, const float* bias, int64_t B, int64_t C, int64_t N,
                                 int64_t IH, int64_t IW, int64_t OH, int64_t OW, int kh, int kw, int sy, int sx, int py,
                                 int px, sr_stream_t stream) {
    if (!fits(B, C, N, IH, IW, OH, OW, kh, kw)) return SR_ERANGE;
    const GConv p = make(B, C, N, IH, IW, OH, OW, kh, kw, sy, sx, py, px);
    if (!geometry_ok(p)) return SR_EINVAL;
    if (B == 0) return SR_OK;
    if (!y || !x || !w) return SR_EINVAL;
    const int64_t opix = OH * OW;
    const dim3 grid = row_grid(opix, B * N);
    hipLaunchKernelGGL(k_gconv_fwd, grid, dim3(256), 0, sr_stream(stream), y, x, w, bias, p);
    return sr_launch_status();
}

extern "C" int sr_conv2d_generic_dgrad(float* dx, const float* g, const float* w, int64_t B, int64_t C, int64_t N, int64_t IH,
                                       int64_t IW, int64_t OH, int64_t OW, int kh, int kw, int sy, int sx, int py, int px,
                                       sr_stream_t stream) {
    if (!fits(B, C, N, IH, IW, OH, OW, kh, kw)) return SR_ERANGE;
    const GConv p = make(B, C, N, IH, IW, OH, OW, kh, kw, sy, sx, py, px);
    // the data gradient is also the TRANSPOSED convolution as a forward operator: any (IH, IW) whose convolution has
    // the extent (OH, OW) is legal (a strided convolution maps up to `stride` input extents onto one output extent)
    if (!geometry_ok(p)) return SR_EINVAL;
    if (B == 0) return SR_OK;
    if (!dx || !g || !w) return SR_EINVAL;
    const int64_t ipix = IH * IW;
    const dim3 grid = row_grid(ipix, B * C);
    hipLaunchKernelGGL(k_gconv_dgrad, grid, dim3(256), 0, sr_stream(stream), dx, g, w, p);
    return sr_launch_status();
}

extern "C" int sr_conv2d_generic_wgrad(float* dw, const float* x, const float* g, int64_t B, int64_t C, int64_t N, int64_t IH,
                                       int64_t IW, int64_t OH, int64_t OW, int kh, int kw, int sy, int sx, int py, int px,
                                       sr_stream_t stream) {
    if (!fits(B, C, N, IH, IW, OH, OW, kh, kw)) return SR_ERANGE;
    const GConv p = make(B, C, N, IH, IW, OH, OW, kh, kw, sy, sx, py, px);
    if (!geometry_ok(p)) return SR_EINVAL;
    if (!dw || (B > 0 && (!x || !g))) return SR_EINVAL;
    hipLaunchKernelGGL(k_gconv_wgrad, dim3((unsigned)(N * C * kh * kw)), dim3(256), 0, sr_stream(stream), dw, x, g, p);
    return sr_launch_status();
}
