// 1x1 stride-1 convolution as a plain GEMM on the fp32 matrix cores (round 6).
//
//   out[b, n, p] = oscale[b, n] * sum_c wt[c, n] * (iscale[b, c] * x[b, c, p]) + obias[n]        p = pixel, contiguous
//
// The discriminator's skip convolutions and their data gradients (reference model.py:296-336: ResBlock.skip, 128 -> 256
// ... 512 -> 512 channels on 128^2 ... 16^2 maps) ran on the 1x1 instantiation of k_conv_mfma, whose machinery is built
// for windows: a 32 x 4 pixel patch, 16 channels and ONE tap of K per barrier chain — 75-103 TFLOP/s on these shapes
// (scripts/bench_conv1x1_gemm.py), this kernel 94-116 with bit-identical results (same K order).  Without a window both operands are already K-major with the MFMA's lane
// dimension contiguous (weights [c][n], activations [c][p]): a 128 (n) x 128 (p) tile per workgroup, 16 channels per
// chunk through a double-buffered LDS pair (16-byte global loads -> 16-byte LDS writes), each of the four waves owns
// 64 x 64 = 2 x 2 accumulator tiles of v_mfma_f32_32x32x2_f32: four ds_read_b32 feed four MFMAs per k-step.
// Rows of the accumulator tile are output channels, its 32 lanes consecutive pixels: 128-byte store segments.
//
// Eligible: C % 16 == 0, N % 128 == 0, pixels % 128 == 0, 16-byte aligned operands, enough tiles to fill the chip
// (anything else stays on k_conv_mfma).  SR_CONV1X1_GEMM=0 disables.
#include "common.h"
#include "conv1x1_gemm.h"

#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TN = 128, TP = 128, KC = 16;

struct GemmParams {
    const float* x;
    const float* wt;
    const float* iscale;
    const float* oscale;
    const float* obias;
    const float* addend;    // optional [B, N, P]: out = conv + addend (ResBlock: conv2(conv1(x)) + skip(x), reference model.py)
    float* out;
    int C, N, ldw;
    int64_t P;
    int tiles_p, tiles_n;
};

__global__ __launch_bounds__(256, 2) void k_conv1x1_gemm(const GemmParams q) {
#if __HIP_DEVICE_COMPILE__
    __shared__ float s_a[2][KC][TN];      // weights  [k][n]
    __shared__ float s_b[2][KC][TP];      // activations [k][p]
    int bid = blockIdx.x;
    const int tp = bid % q.tiles_p;
    bid /= q.tiles_p;
    const int tn = bid % q.tiles_n;
    const int b = bid / q.tiles_n;
    const int n0 = tn * TN;
    const int64_t p0 = (int64_t)tp * TP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wn = (wave >> 1) * 64, wp = (wave & 1) * 64;        // the wave's 64 x 64 corner of the tile

    // staging: 16 rows x 128 floats = 512 float4 per operand and chunk, two per lane
    const int r0 = tid >> 5, c4 = (tid & 31) * 4;                 // rows r0 and r0 + 8
    const float* wsrc = q.wt + (int64_t)r0 * q.ldw + n0 + c4;
    const float* xsrc = q.x + ((int64_t)b * q.C + r0) * q.P + p0 + c4;
    const float* isc = q.iscale ? q.iscale + (int64_t)b * q.C : nullptr;
    const int64_t wstep8 = (int64_t)8 * q.ldw, xstep8 = (int64_t)8 * q.P;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra0, ra1, rb0, rb1;
    auto fetch = [&](int kc) {
        const float* w = wsrc + (int64_t)kc * KC * q.ldw;
        const float* xx = xsrc + (int64_t)kc * KC * q.P;
        ra0 = *reinterpret_cast<const float4*>(w);
        ra1 = *reinterpret_cast<const float4*>(w + wstep8);
        rb0 = *reinterpret_cast<const float4*>(xx);
        rb1 = *reinterpret_cast<const float4*>(xx + xstep8);
        if (isc) {
            const float s0 = isc[kc * KC + r0], s1 = isc[kc * KC + r0 + 8];
            rb0.x *= s0; rb0.y *= s0; rb0.z *= s0; rb0.w *= s0;
            rb1.x *= s1; rb1.y *= s1; rb1.z *= s1; rb1.w *= s1;
        }
    };
    auto stash = [&](int buf) {
        *reinterpret_cast<float4*>(&s_a[buf][r0][c4]) = ra0;
        *reinterpret_cast<float4*>(&s_a[buf][r0 + 8][c4]) = ra1;
        *reinterpret_cast<float4*>(&s_b[buf][r0][c4]) = rb0;
        *reinterpret_cast<float4*>(&s_b[buf][r0 + 8][c4]) = rb1;
    };

    const int nchunks = q.C / KC;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int kc = 0; kc < nchunks; ++kc) {
        const int buf = kc & 1;
        const bool more = kc + 1 < nchunks;
        if (more) fetch(kc + 1);                                   // in flight under this chunk's MFMAs
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            const int k = 2 * kk + half;
            const float a0 = s_a[buf][k][wn + l31], a1 = s_a[buf][k][wn + 32 + l31];
            const float b0 = s_b[buf][k][wp + l31], b1 = s_b[buf][k][wp + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
    }

    // epilogue: rows = output channels (register r -> row (r & 3) + 8 * (r >> 2) + 4 * half), lanes = pixels
    float* dst = q.out + ((int64_t)b * q.N + n0 + wn) * q.P + p0 + wp + l31;
    const float* osc = q.oscale ? q.oscale + (int64_t)b * q.N + n0 + wn : nullptr;
    const float* bia = q.obias ? q.obias + n0 + wn : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float s = osc ? osc[row] : 1.0f;
            const float t = bia ? bia[row] : 0.0f;
            float* d = dst + (int64_t)row * q.P;
            float v0 = acc[i][0][r] * s + t, v1 = acc[i][1][r] * s + t;
            if (q.addend) {
                const float* a = q.addend + (d - q.out);
                v0 += a[0];
                v1 += a[32];
            }
            d[0] = v0;
            d[32] = v1;
        }
#endif
}

// ---- tap-split transposed convolution on the same tile ------------------------------------------------------------------
// The stride-2 transposed 3x3 convolution of a small map is nine shifted 1x1 convolutions over the common (IH+1) x (IW+1)
// phase grid (conv_mfma.hip, TAP9).  On 32 x 4 / 16 x 8 pixel PATCHES a 17 x 17 grid fills 45 % of its tiles (32-wide rows
// for 17 columns, 20 rows for 17): 43 TFLOP/s for the 16^2 -> 32^2 layer at batch 4.  Here the pixel dimension of a tile is
// 128 consecutive points of the FLATTENED (sample, grid point) index — only the last tile of a launch is partial — and the
// shifted, border-clipped window is a per-lane base pointer + mask computed once (a lane stages ONE pixel column of the
// tile for all chunks).  Raw sums go to partial[slice * 9 + tap][b][n][grid point]: the layout k_convt_tap_reduce adds up.
struct TapParams {
    const float* x;
    const float* wt;        // [9][C][ldw]
    const float* iscale;
    float* partial;
    int B, C, N, ldw, IH, IW, GW, region;
    int tiles_q, tiles_n, c_per_slice;
    int64_t total_q;        // B * region
};

__global__ __launch_bounds__(256, 2) void k_convt_taps_gemm(const TapParams q) {
#if __HIP_DEVICE_COMPILE__
    __shared__ float s_a[2][KC][TN];
    __shared__ float s_b[2][KC][TP];
    int bid = blockIdx.x;
    const int tap = bid % 9;                    // the taps of a tile are neighbours: one input patch in L2
    bid /= 9;
    const int tn = bid % q.tiles_n;
    bid /= q.tiles_n;
    const int tq = bid % q.tiles_q;
    const int slice = bid / q.tiles_q;
    const int c_beg = slice * q.c_per_slice;
    const int c_end = min(q.C, c_beg + q.c_per_slice);
    const int n0 = tn * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wn = (wave >> 1) * 64, wp = (wave & 1) * 64;
    const int ky = tap / 3, kx = tap - 3 * ky;

    // weights: rows r0, r0 + 8 of the chunk, 16 bytes per lane
    const int r0 = tid >> 5, c4 = (tid & 31) * 4;
    const float* wsrc = q.wt + ((int64_t)tap * q.C + c_beg + r0) * q.ldw + n0 + c4;
    const int64_t wstep8 = (int64_t)8 * q.ldw;
    // activations: this lane's pixel column, rows rb0 + 2 i
    const int col = tid & 127, rb0 = tid >> 7;
    const int64_t qi = (int64_t)tq * TP + col;
    const int plane = q.IH * q.IW;
    bool ok = qi < q.total_q;
    const int bq = ok ? (int)(qi / q.region) : 0;
    const int g = ok ? (int)(qi - (int64_t)bq * q.region) : 0;
    const int yy = g / q.GW - (ky >> 1), xx = g % q.GW - (kx >> 1);
    ok = ok && yy >= 0 && yy < q.IH && xx >= 0 && xx < q.IW;
    const float* xsrc = q.x + ((int64_t)bq * q.C + c_beg + rb0) * plane + (ok ? yy * q.IW + xx : 0);
    const float* isc = q.iscale ? q.iscale + (int64_t)bq * q.C + c_beg + rb0 : nullptr;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra0, ra1;
    float rb[8];
    auto fetch = [&](int kc) {
        const float* w = wsrc + (int64_t)kc * KC * q.ldw;
        ra0 = *reinterpret_cast<const float4*>(w);
        ra1 = *reinterpret_cast<const float4*>(w + wstep8);
        const float* xx_ = xsrc + (int64_t)kc * KC * plane;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = ok ? xx_[(int64_t)(2 * i) * plane] : 0.0f;
            if (isc) v *= isc[kc * KC + 2 * i];
            rb[i] = v;
        }
    };
    auto stash = [&](int buf) {
        *reinterpret_cast<float4*>(&s_a[buf][r0][c4]) = ra0;
        *reinterpret_cast<float4*>(&s_a[buf][r0 + 8][c4]) = ra1;
#pragma unroll
        for (int i = 0; i < 8; ++i) s_b[buf][rb0 + 2 * i][col] = rb[i];
    };

    const int nchunks = (c_end - c_beg) / KC;
    if (nchunks > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    for (int kc = 0; kc < nchunks; ++kc) {
        const int buf = kc & 1;
        const bool more = kc + 1 < nchunks;
        if (more) fetch(kc + 1);
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            const int k = 2 * kk + half;
            const float a0 = s_a[buf][k][wn + l31], a1 = s_a[buf][k][wn + 32 + l31];
            const float b0 = s_b[buf][k][wp + l31], b1 = s_b[buf][k][wp + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
    }

    // raw sums: partial[((slice * 9 + tap) * B + b) * N + n][grid point]
    float* base = q.partial + ((int64_t)slice * 9 + tap) * q.B * q.N * q.region;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t qo = (int64_t)tq * TP + wp + j * 32 + l31;
        if (qo >= q.total_q) continue;
        const int bo = (int)(qo / q.region);
        const int go = (int)(qo - (int64_t)bo * q.region);
        float* d = base + ((int64_t)bo * q.N + n0 + wn) * q.region + go;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                d[(int64_t)row * q.region] = acc[i][j][r];
            }
    }
#endif
}

// SR_CONV1X1_GEMM: "0" off, "force" also below the tile count that fills the chip (tests)
int mode() {
    const char* e = std::getenv("SR_CONV1X1_GEMM");
    if (e && e[0] == '0') return 0;
    if (e && e[0] == 'f') return 2;
    return 1;
}

}  // namespace

bool sr_conv1x1_gemm_eligible(int64_t B, int64_t C, int64_t N, int64_t ldw, int64_t P, const void* in, const void* wt,
                              const void* out) {
    const int m = mode();
    if (m == 0 || B <= 0) return false;
    if (C % KC != 0 || N % TN != 0 || P % TP != 0 || ldw % 4 != 0) return false;
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(wt) | reinterpret_cast<uintptr_t>(out)) & 15) != 0)
        return false;
    const int64_t tiles = (P / TP) * (N / TN) * B;
    // fewer tiles than one per CU: the split-K slices of k_conv_mfma fill the chip better (512 -> 512 at 32^2: 47.5
    // against 54.6 us with 256 tiles at batch 8, 45.2 against 42.4 with 128 at batch 4)
    return (m == 2 || tiles >= SR_NUM_CU) && tiles < (1LL << 31);
}

int sr_conv1x1_gemm_launch(float* out, const float* in, const float* wt, int64_t ldw, const float* iscale,
                           const float* oscale, const float* obias, int64_t B, int64_t C, int64_t N, int64_t P,
                           hipStream_t st, const float* addend) {
    GemmParams q;
    q.x = in; q.wt = wt; q.iscale = iscale; q.oscale = oscale; q.obias = obias; q.out = out; q.addend = addend;
    q.C = (int)C; q.N = (int)N; q.ldw = (int)ldw; q.P = P;
    q.tiles_p = (int)(P / TP); q.tiles_n = (int)(N / TN);
    const int64_t tiles = (int64_t)q.tiles_p * q.tiles_n * B;
    hipLaunchKernelGGL(k_conv1x1_gemm, dim3((unsigned)tiles), dim3(256), 0, st, q);
    return sr_launch_status();
}

bool sr_convt_taps_gemm_eligible(int64_t B, int64_t C, int64_t N, int64_t IW, int64_t ldw, int c_per_slice,
                                 const void* wt) {
    const char* e = std::getenv("SR_CONVT_TAPS_GEMM");
    if (e && e[0] == '0') return false;
    // maps up to 32 wide: their (2^k + 1)-wide grids fill half of a 32-wide patch row; from 64 up the patch form's
    // aligned 16-byte DMA wins (batch 1, scripts/bench_convt_small.py: 0.169 against 0.180 ms at 64^2, 0.172 / 0.188 at 128^2)
    if (IW > 32 && !(e && e[0] == 'f')) return false;
    return B > 0 && C % KC == 0 && c_per_slice % KC == 0 && N % TN == 0 && ldw % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(wt) & 15) == 0;
}

int sr_convt_taps_gemm_launch(float* partial, const float* in, const float* wt, int64_t ldw, const float* iscale, int64_t B,
                              int64_t C, int64_t N, int64_t IH, int64_t IW, int ks, int c_per_slice, hipStream_t st) {
    TapParams q;
    q.x = in; q.wt = wt; q.iscale = iscale; q.partial = partial;
    q.B = (int)B; q.C = (int)C; q.N = (int)N; q.ldw = (int)ldw; q.IH = (int)IH; q.IW = (int)IW;
    q.GW = (int)IW + 1; q.region = (int)((IH + 1) * (IW + 1));
    q.total_q = B * q.region;
    q.tiles_q = (int)((q.total_q + TP - 1) / TP); q.tiles_n = (int)(N / TN); q.c_per_slice = c_per_slice;
    const int64_t blocks = (int64_t)q.tiles_q * q.tiles_n * 9 * ks;
    if (blocks > 0x7FFFFFFFLL) return SR_ERANGE;
    hipLaunchKernelGGL(k_convt_taps_gemm, dim3((unsigned)blocks), dim3(256), 0, st, q);
    return sr_launch_status();
}

// C ABI: the fused form  out = oscale * conv1x1(in, wt) + addend  (no window, stride 1).  _supported says whether the
// GEMM-shaped kernel takes the call (shape / alignment / tile count, SR_CONV1X1_GEMM); otherwise the caller adds separately.
extern "C" int sr_conv1x1_add_supported(int64_t B, int64_t C, int64_t N, int64_t wt_ld, int64_t P, const float* in,
                                        const float* wt, const float* out, const float* addend) {
    return sr_conv1x1_gemm_eligible(B, C, N, wt_ld, P, in, wt, out) && (reinterpret_cast<uintptr_t>(addend) & 15) == 0 ? 1 : 0;
}

extern "C" int sr_conv1x1_add(float* out, const float* in, const float* wt, const float* oscale, const float* addend,
                              int64_t B, int64_t C, int64_t N, int64_t wt_ld, int64_t P, sr_stream_t stream) {
    if (!out || !in || !wt || !addend) return SR_EINVAL;
    if (!sr_conv1x1_add_supported(B, C, N, wt_ld, P, in, wt, out, addend)) return SR_EINVAL;
    if (B * C * P >= (1LL << 31)) return SR_ERANGE;
    return sr_conv1x1_gemm_launch(out, in, wt, wt_ld, nullptr, oscale, nullptr, B, C, N, P, sr_stream(stream), addend);
}
