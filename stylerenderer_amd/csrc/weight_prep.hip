// Weight preparation for the MFMA convolutions (C ABI: sr_weight_prep, sr_weight_prep_bwd,
// sr_weight_adjoint).
//
// The reference scales and modulates a [B, Co, Ci, k, k] weight tensor with half a dozen ATen
// passes per layer (layers.py:293-300: scale * weight * style, pow, sum, rsqrt, views).  Here the
// shared weight is touched once per step:
//   k_wprep      W[Co,Ci,kk] -> Wt[kk,Ci,ld] = scale*W (tap-major, Cout contiguous, 16-byte row
//                pitch: the layout k_conv_mfma streams) and Wsq[Ci,Co] = sum_taps (scale*W)^2 (the
//                demodulation matrix: rsqrt(style^2 @ Wsq + eps) is the per-sample output scale)
//   k_wprep_bwd  dW = scale*dWt^T + 2*scale^2 * W * dWsq   (one pass, both cotangents)
//   k_wadjoint   Wt[kk,C,ldn] -> WtA[kk',N,ldc]: channel transpose (+ tap reversal for the
//                stride-1 correlation) = the weights of the data-gradient convolution
// All three are small (<= 9.4 MB at 512x512x3x3) HBM passes; the point is launch count.
#include "common.h"

namespace {

template <int KK>
__global__ __launch_bounds__(256) void k_wprep(float* __restrict__ wt, float* __restrict__ wsq,
                                               const float* __restrict__ w, float scale, int Co, int Ci,
                                               int ld) {
    const int co = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ci = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ci >= Ci || co >= ld) return;
    float v[KK];
    float sq = 0.0f;
    if (co < Co) {
        const float* src = w + ((int64_t)co * Ci + ci) * KK;
#pragma unroll
        for (int t = 0; t < KK; ++t) {
            v[t] = src[t] * scale;
            sq += v[t] * v[t];
        }
    } else {
#pragma unroll
        for (int t = 0; t < KK; ++t) v[t] = 0.0f;
    }
#pragma unroll
    for (int t = 0; t < KK; ++t) wt[((int64_t)t * Ci + ci) * ld + co] = v[t];
    if (wsq && co < Co) wsq[(int64_t)ci * Co + co] = sq;
}

// One workgroup = 16 output channels x 16 input channels.  gwt / gwsq are read with co fastest (64-byte
// runs), w and gw are [co][ci][KK] with ci*KK fastest (16 * KK contiguous floats per co): both sides go
// through an LDS tile so that neither is a 36-byte-strided access.
template <int KK>
__global__ __launch_bounds__(256) void k_wprep_bwd(float* __restrict__ gw, const float* __restrict__ gwt,
                                                   const float* __restrict__ gwsq,
                                                   const float* __restrict__ w, float scale, int Co,
                                                   int Ci, int ldg) {
    __shared__ float tile[16][16 * KK + 1];              // [co][ci * KK + t]
    const int co0 = blockIdx.x * 16, ci0 = blockIdx.y * 16;
    const int row_floats = 16 * KK;
    // (1) w tile -> LDS, row-contiguous
    if (gwsq) {
        for (int e = threadIdx.x; e < 16 * row_floats; e += 256) {
            const int r = e / row_floats, q = e % row_floats;
            const int co = co0 + r, ci = ci0 + q / KK;
            tile[r][q] = (co < Co && ci < Ci) ? w[((int64_t)co * Ci + ci0) * KK + q] : 0.0f;
        }
        __syncthreads();
    }
    // (2) thread (co fastest, ci): combine the two cotangents, result back into the tile
    const int lco = threadIdx.x & 15, lci = threadIdx.x >> 4;
    const int co = co0 + lco, ci = ci0 + lci;
    const bool ok = co < Co && ci < Ci;
    const float q2 = (gwsq && ok) ? 2.0f * scale * scale * gwsq[(int64_t)ci * Co + co] : 0.0f;
#pragma unroll
    for (int t = 0; t < KK; ++t) {
        float g = (gwt && ok) ? scale * gwt[((int64_t)t * Ci + ci) * ldg + co] : 0.0f;
        if (gwsq) g += q2 * tile[lco][lci * KK + t];
        tile[lco][lci * KK + t] = g;
    }
    __syncthreads();
    // (3) tile -> gw, row-contiguous
    for (int e = threadIdx.x; e < 16 * row_floats; e += 256) {
        const int r = e / row_floats, q = e % row_floats;
        const int oco = co0 + r, oci = ci0 + q / KK;
        if (oco < Co && oci < Ci) gw[((int64_t)oco * Ci + ci0) * KK + q] = tile[r][q];
    }
}

// out[tA][n][c] = in[t][c][n]; 32x32 tiles through LDS so both sides are coalesced
__global__ __launch_bounds__(256) void k_wadjoint(float* __restrict__ out, const float* __restrict__ in,
                                                  int C, int N, int ldn, int ldc, int KK, int flip) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int tA = flip ? KK - 1 - t : t;
    const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, n = n0 + tx;
        tile[r][tx] = (c < C && n < N) ? in[((int64_t)t * C + c) * ldn + n] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, c = c0 + tx;
        if (n < N && c < ldc) out[((int64_t)tA * N + n) * ldc + c] = tile[tx][r];
    }
}


// ---- batched forms: every layer of a network in ONE launch ---------------------------------------------------------
// A training iteration at the reference's per-GPU batch prepares ~230 weights in its forward passes (k_wprep +
// k_wadjoint per convolution and phase), each a 5 us launch on a tensor of a few hundred KB: launch count, not bytes.
// The batched kernels take the per-layer pointers and shapes BY VALUE in the kernel argument (<= SR_WB_MAX items, 2.3 KB; no
// device-side table to keep alive, graph capture bakes them in) and map a flat workgroup index to (item, tile).
constexpr int SR_WB_MAX = 48;

struct WPrepBatch {
    float* wt[SR_WB_MAX];
    float* wsq[SR_WB_MAX];
    const float* w[SR_WB_MAX];
    float scale[SR_WB_MAX];
    int co[SR_WB_MAX], ci[SR_WB_MAX], ld[SR_WB_MAX], kk[SR_WB_MAX];
    int first[SR_WB_MAX + 1];          // first flat workgroup of item i; first[n] = total
    int n;
};

__device__ __forceinline__ int wb_item(const int* first, int n, int bid) {
    int it = 0;
    while (it + 1 < n && bid >= first[it + 1]) ++it;
    return it;
}

template <int KK>
__device__ __forceinline__ void wprep_tile(float* __restrict__ wt, float* __restrict__ wsq, const float* __restrict__ w,
                                           float scale, int Co, int Ci, int ld, int bx, int by) {
    const int co = bx * 64 + (threadIdx.x & 63);
    const int ci = by * 4 + (threadIdx.x >> 6);
    if (ci >= Ci || co >= ld) return;
    float v[KK];
    float sq = 0.0f;
    if (co < Co) {
        const float* src = w + ((int64_t)co * Ci + ci) * KK;
#pragma unroll
        for (int t = 0; t < KK; ++t) {
            v[t] = src[t] * scale;
            sq += v[t] * v[t];
        }
    } else {
#pragma unroll
        for (int t = 0; t < KK; ++t) v[t] = 0.0f;
    }
#pragma unroll
    for (int t = 0; t < KK; ++t) wt[((int64_t)t * Ci + ci) * ld + co] = v[t];
    if (wsq && co < Co) wsq[(int64_t)ci * Co + co] = sq;
}

__global__ __launch_bounds__(256) void k_wprep_batch(const WPrepBatch b) {
    const int it = wb_item(b.first, b.n, blockIdx.x);
    const int local = blockIdx.x - b.first[it];
    const int gx = (b.ld[it] + 63) / 64;
    if (b.kk[it] == 9) wprep_tile<9>(b.wt[it], b.wsq[it], b.w[it], b.scale[it], b.co[it], b.ci[it], b.ld[it], local % gx, local / gx);
    else wprep_tile<1>(b.wt[it], b.wsq[it], b.w[it], b.scale[it], b.co[it], b.ci[it], b.ld[it], local % gx, local / gx);
}

struct WAdjBatch {
    float* out[SR_WB_MAX];
    const float* in[SR_WB_MAX];
    int C[SR_WB_MAX], N[SR_WB_MAX], ldn[SR_WB_MAX], ldc[SR_WB_MAX], KK[SR_WB_MAX], flip[SR_WB_MAX];
    int first[SR_WB_MAX + 1];
    int n;
};

__global__ __launch_bounds__(256) void k_wadjoint_batch(const WAdjBatch b) {
    __shared__ float tile[32][33];
    const int it = wb_item(b.first, b.n, blockIdx.x);
    int local = blockIdx.x - b.first[it];
    const int C = b.C[it], N = b.N[it], ldn = b.ldn[it], ldc = b.ldc[it], KK = b.KK[it];
    const int gx = (ldc + 31) / 32, gy = (N + 31) / 32;
    const int bx = local % gx;
    local /= gx;
    const int by = local % gy, t = local / gy;
    const int tA = b.flip[it] ? KK - 1 - t : t;
    const float* __restrict__ in = b.in[it];
    float* __restrict__ out = b.out[it];
    const int c0 = bx * 32, n0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, n = n0 + tx;
        tile[r][tx] = (c < C && n < N) ? in[((int64_t)t * C + c) * ldn + n] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, c = c0 + tx;
        if (n < N && c < ldc) out[((int64_t)tA * N + n) * ldc + c] = tile[tx][r];
    }
}

}  // namespace

extern "C" int sr_weight_prep(float* wt, float* wsq, const float* w, float scale, int64_t Co, int64_t Ci,
                              int ksize, int64_t ld, sr_stream_t stream) {
    if (Co <= 0 || Ci <= 0 || !wt || !w || ld < Co || (ld & 3)) return SR_EINVAL;
    if (Co > (1 << 20) || Ci > (1 << 20)) return SR_ERANGE;
    const dim3 grid((unsigned)sr_ceil_div(ld, 64), (unsigned)sr_ceil_div(Ci, 4));
    hipStream_t st = sr_stream(stream);
    if (ksize == 3)
        hipLaunchKernelGGL(k_wprep<9>, grid, dim3(256), 0, st, wt, wsq, w, scale, (int)Co, (int)Ci, (int)ld);
    else if (ksize == 1)
        hipLaunchKernelGGL(k_wprep<1>, grid, dim3(256), 0, st, wt, wsq, w, scale, (int)Co, (int)Ci, (int)ld);
    else
        return SR_EINVAL;
    return sr_launch_status();
}

extern "C" int sr_weight_prep_bwd(float* gw, const float* gwt, const float* gwsq, const float* w,
                                  float scale, int64_t Co, int64_t Ci, int ksize, int64_t ldg,
                                  sr_stream_t stream) {
    if (Co <= 0 || Ci <= 0 || !gw || (!gwt && !gwsq) || (gwsq && !w) || (gwt && ldg < Co)) return SR_EINVAL;
    if (Co > (1 << 20) || Ci > (1 << 20)) return SR_ERANGE;
    const dim3 grid((unsigned)sr_ceil_div(Co, 16), (unsigned)sr_ceil_div(Ci, 16));
    hipStream_t st = sr_stream(stream);
    if (ksize == 3)
        hipLaunchKernelGGL(k_wprep_bwd<9>, grid, dim3(256), 0, st, gw, gwt, gwsq, w, scale, (int)Co, (int)Ci,
                           (int)ldg);
    else if (ksize == 1)
        hipLaunchKernelGGL(k_wprep_bwd<1>, grid, dim3(256), 0, st, gw, gwt, gwsq, w, scale, (int)Co, (int)Ci,
                           (int)ldg);
    else
        return SR_EINVAL;
    return sr_launch_status();
}

extern "C" int sr_weight_adjoint(float* out, const float* in, int64_t taps, int64_t C, int64_t N, int64_t ldn,
                                 int64_t ldc, int flip, sr_stream_t stream) {
    if (taps <= 0 || C <= 0 || N <= 0 || !out || !in || ldn < N || ldc < C) return SR_EINVAL;
    if (taps > 65535 || sr_ceil_div(N, 32) > 65535) return SR_ERANGE;
    const dim3 grid((unsigned)sr_ceil_div(ldc, 32), (unsigned)sr_ceil_div(N, 32), (unsigned)taps);
    hipLaunchKernelGGL(k_wadjoint, grid, dim3(256), 0, sr_stream(stream), out, in, (int)C, (int)N, (int)ldn,
                       (int)ldc, (int)taps, flip);
    return sr_launch_status();
}

// ---- batched entry points: arrays of n per-layer arguments (host memory), chunked into launches of <= 48 layers ----
extern "C" int sr_weight_prep_batch(int n, float* const* wt, float* const* wsq, const float* const* w,
                                    const float* scale, const int64_t* Co, const int64_t* Ci, const int* ksize,
                                    const int64_t* ld, sr_stream_t stream) {
    if (n < 0 || (n > 0 && (!wt || !wsq || !w || !scale || !Co || !Ci || !ksize || !ld))) return SR_EINVAL;
    hipStream_t st = sr_stream(stream);
    for (int base = 0; base < n; base += SR_WB_MAX) {
        WPrepBatch b;
        b.n = n - base < SR_WB_MAX ? n - base : SR_WB_MAX;
        int64_t total = 0;
        for (int i = 0; i < b.n; ++i) {
            const int g = base + i;
            if (Co[g] <= 0 || Ci[g] <= 0 || !wt[g] || !w[g] || ld[g] < Co[g] || (ld[g] & 3) || (ksize[g] != 1 && ksize[g] != 3))
                return SR_EINVAL;
            if (Co[g] > (1 << 20) || Ci[g] > (1 << 20)) return SR_ERANGE;
            b.wt[i] = wt[g]; b.wsq[i] = wsq[g]; b.w[i] = w[g]; b.scale[i] = scale[g];
            b.co[i] = (int)Co[g]; b.ci[i] = (int)Ci[g]; b.ld[i] = (int)ld[g]; b.kk[i] = ksize[g] * ksize[g];
            b.first[i] = (int)total;
            total += sr_ceil_div(ld[g], 64) * sr_ceil_div(Ci[g], 4);
            if (total > 0x7FFFFFFFLL) return SR_ERANGE;
        }
        b.first[b.n] = (int)total;
        if (total > 0) hipLaunchKernelGGL(k_wprep_batch, dim3((unsigned)total), dim3(256), 0, st, b);
    }
    return sr_launch_status();
}

extern "C" int sr_weight_adjoint_batch(int n, float* const* out, const float* const* in, const int64_t* taps,
                                       const int64_t* C, const int64_t* N, const int64_t* ldn, const int64_t* ldc,
                                       const int* flip, sr_stream_t stream) {
    if (n < 0 || (n > 0 && (!out || !in || !taps || !C || !N || !ldn || !ldc || !flip))) return SR_EINVAL;
    hipStream_t st = sr_stream(stream);
    for (int base = 0; base < n; base += SR_WB_MAX) {
        WAdjBatch b;
        b.n = n - base < SR_WB_MAX ? n - base : SR_WB_MAX;
        int64_t total = 0;
        for (int i = 0; i < b.n; ++i) {
            const int g = base + i;
            if (taps[g] <= 0 || C[g] <= 0 || N[g] <= 0 || !out[g] || !in[g] || ldn[g] < N[g] || ldc[g] < C[g]) return SR_EINVAL;
            if (taps[g] > 65535 || C[g] > (1 << 20) || N[g] > (1 << 20)) return SR_ERANGE;
            b.out[i] = out[g]; b.in[i] = in[g];
            b.C[i] = (int)C[g]; b.N[i] = (int)N[g]; b.ldn[i] = (int)ldn[g]; b.ldc[i] = (int)ldc[g];
            b.KK[i] = (int)taps[g]; b.flip[i] = flip[g];
            b.first[i] = (int)total;
            total += sr_ceil_div(ldc[g], 32) * sr_ceil_div(N[g], 32) * taps[g];
            if (total > 0x7FFFFFFFLL) return SR_ERANGE;
        }
        b.first[b.n] = (int)total;
        if (total > 0) hipLaunchKernelGGL(k_wadjoint_batch, dim3((unsigned)total), dim3(256), 0, st, b);
    }
    return sr_launch_status();
}
