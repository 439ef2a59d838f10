// Implicit-GEMM convolution on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// This is the contraction of the reference's ModulatedConv2d (reference layers.py:293-323), which the
// reference runs as F.conv2d / F.conv_transpose2d with groups = batch over per-sample weight copies
// ([B*Cout, Cin, k, k]: 151 MB per 512x512 layer at B = 16).  Here the modulation is moved onto the
// operands instead (mathematically identical):
//
//   out[b, n, oy, ox] = oscale[b, n] * sum_{tap, c} Wt[tap][c][n] * (iscale[b, c] * in[b, c, iy, ix])  (+ obias[n])
//
// with ONE shared weight tensor Wt [taps][C][N], the style s[b, c] applied to the activation tile
// while it is staged into LDS, and the demodulation d[b, n] applied in the epilogue.  The same
// kernel is the data-gradient (swap iscale/oscale, flipped + transposed weights) and serves plain
// EqualConv2d layers (no scales, optional bias).
//
// Mapping (one 256-thread workgroup = 4 waves):
//   GEMM  D[n][pixel] = sum_k A[n][k] * Bm[k][pixel],  A = weights, Bm = input window; k = (tap, c).
//   tile  128 output channels x 128 pixels (a PB x PH x PW patch of the output grid), wave = 64 x 64
//         = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator VGPRs).
//   K     channel chunks of 8: the input HALO patch of the chunk is staged ONCE in LDS and all
//         taps read it at shifted addresses (9x less staging than im2col); the chunk's weights
//         [taps][8][128] sit next to it.  Operand fetch is one conflict-free ds_read_b32 per MFMA
//         operand (lanes 0-31 = 32 consecutive pixels / channels, lanes 32-63 = the next k).
//   pipe  next chunk's global loads are issued into registers before the MFMA block of the
//         current chunk and written to LDS after it (async-stage split), 2-3 workgroups per CU.
//   out   lane (l & 31) owns a pixel, registers own channels: every store instruction writes
//         32 consecutive pixels of one channel row per half-wave (128 B segments).
// Variants by template: input stride 1 / 2, tap window (3x3, 2x2, 2x1, 1x2, 1x1) — the stride-2
// transposed convolution of the upsampling layers is run as its four output phases — and the patch
// shape for 4x4 ... 256x256 feature maps.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// channels per K chunk: 8 (4 for the stride-2 3x3 window, whose halo patch is 4x larger)
constexpr int BN = 128;    // output channels per workgroup
constexpr int BM = 128;    // pixels per workgroup

struct ConvParams {
    const float* in;
    const float* wt;
    const float* iscale;
    const float* oscale;
    const float* obias;
    float* out;
    int B, C, N;
    int IH, IW;          // input extent
    int GH, GW;          // output grid computed by this launch (phase space)
    int OH, OW;          // full output extent
    int osy, osx, ooy, oox;   // output coordinate = grid * os + oo
    int dy0, dx0;        // input coordinate = grid * IS + d0 + tap
    int wmap[9];         // weight slab of window position (ty, tx)
    int tiles_x, tiles_y, tiles_b, tiles_n;
};

template <int IS, int TY, int TX, int PW, int PH, int PB>
struct Geo {
    static constexpr int KC = (IS == 2 && TY * TX == 9) ? 4 : 8;
    static constexpr int EH = (PH - 1) * IS + TY;
    static constexpr int EW = (PW - 1) * IS + TX;
    static constexpr int EWP = EW + ((EW % 2 == 0) ? 1 : 0);     // odd row pitch
    static constexpr int PLANE = PB * EH * EWP;                  // one channel of the chunk
    static constexpr int IN_ELEMS = KC * PB * EH * EW;
    static constexpr int IN_ITERS = (IN_ELEMS + 255) / 256;
    static constexpr int NT = TY * TX;
    static constexpr int LDS_IN = KC * PLANE;
    static constexpr int LDS_W = NT * KC * BN;
};

template <int IS, int TY, int TX, int PW, int PH, int PB>
__global__ __launch_bounds__(256, 2) void k_conv_mfma(const ConvParams p) {
    using G = Geo<IS, TY, TX, PW, PH, PB>;
    static_assert(PW * PH * PB == BM, "patch must hold 128 pixels");
    __shared__ __attribute__((aligned(16))) float s_w[G::LDS_W];
    __shared__ float s_in[G::LDS_IN];

    // ---- tile decode; workgroups that share an input patch (different n tiles) and neighbouring
    // patches are numbered consecutively and kept on one XCD (bid % 8 is the XCD): chunked remap.
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg / SR_NUM_XCD, r = nwg % SR_NUM_XCD, xcd = bid % SR_NUM_XCD;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / SR_NUM_XCD;
    }
    const int n_t = bid % p.tiles_n;
    bid /= p.tiles_n;
    const int tx_i = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty_i = bid % p.tiles_y;
    const int tb_i = bid / p.tiles_y;
    const int n0 = n_t * BN, gy0 = ty_i * PH, gx0 = tx_i * PW, b0 = tb_i * PB;
    const int iy0 = gy0 * IS + p.dy0, ix0 = gx0 * IS + p.dx0;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wco = wave & 1, wpx = wave >> 1;

    // ---- per-lane LDS offsets of the MFMA operands
    int a_off[2], b_off[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        a_off[t] = half * BN + wco * 64 + t * 32 + l31;
        const int m = wpx * 64 + t * 32 + l31;
        const int px = m % PW, py = (m / PW) % PH, pb = m / (PW * PH);
        b_off[t] = half * G::PLANE + (pb * G::EH + py * IS) * G::EWP + px * IS;
    }

    // ---- staging descriptors (what this thread loads every chunk)
    const int w_c = tid >> 5, w_n4 = (tid & 31) * 4;           // weights: channel of chunk, column
    const bool w_vec = (p.N % 4 == 0) && (n0 + w_n4 + 3 < p.N);
    // per staged element only two registers are kept (global offset, LDS offset); the chunk
    // channel and the sample are re-derived from the element number (constant divisors).
    int in_goff[G::IN_ITERS];    // offset inside one channel plane, -1 = outside the image / idle
    int in_lds[G::IN_ITERS];     // LDS offset, -1 = idle slot
#pragma unroll
    for (int it = 0; it < G::IN_ITERS; ++it) {
        const int e = tid + it * 256;
        const int col = e % G::EW, r = (e / G::EW) % G::EH, pb = (e / (G::EW * G::EH)) % PB;
        const int c = e / (G::EW * G::EH * PB);
        const int gy = iy0 + r, gx = ix0 + col, b = b0 + pb;
        const bool live = e < G::IN_ELEMS;
        const bool inside = live && gy >= 0 && gy < p.IH && gx >= 0 && gx < p.IW && b < p.B;
        in_goff[it] = inside ? gy * p.IW + gx : -1;
        in_lds[it] = live ? c * G::PLANE + (pb * G::EH + r) * G::EWP + col : -1;
    }
    const int64_t plane_in = (int64_t)p.IH * p.IW;

    float4 w_reg[G::NT];
    float in_reg[G::IN_ITERS];

    auto fetch = [&](int c0) {
        // weights: one float4 per tap per thread (KC channels x 32 float4 columns <= 256 threads)
#pragma unroll
        for (int t = 0; t < G::NT; ++t) {
            const int c = c0 + w_c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (w_c < G::KC && c < p.C) {
                const float* src = p.wt + ((int64_t)p.wmap[t] * p.C + c) * p.N + n0 + w_n4;
                if (w_vec) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (n0 + w_n4 < p.N) v.x = src[0];
                    if (n0 + w_n4 + 1 < p.N) v.y = src[1];
                    if (n0 + w_n4 + 2 < p.N) v.z = src[2];
                    if (n0 + w_n4 + 3 < p.N) v.w = src[3];
                }
            }
            w_reg[t] = v;
        }
#pragma unroll
        for (int it = 0; it < G::IN_ITERS; ++it) {
            float v = 0.0f;
            const int e = tid + it * 256;
            const int c = c0 + e / (G::EW * G::EH * PB);
            const int b = b0 + (e / (G::EW * G::EH)) % PB;
            if (in_goff[it] >= 0 && c < p.C) {
                v = p.in[((int64_t)b * p.C + c) * plane_in + in_goff[it]];
                if (p.iscale) v *= p.iscale[(int64_t)b * p.C + c];
            }
            in_reg[it] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int t = 0; t < G::NT; ++t)
            if (w_c < G::KC) *reinterpret_cast<float4*>(&s_w[(t * G::KC + w_c) * BN + w_n4]) = w_reg[t];
#pragma unroll
        for (int it = 0; it < G::IN_ITERS; ++it)
            if (in_lds[it] >= 0) s_in[in_lds[it]] = in_reg[it];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    fetch(0);
    for (int c0 = 0; c0 < p.C; c0 += G::KC) {
        __syncthreads();            // everyone finished reading the previous chunk
        commit();
        __syncthreads();
        if (c0 + G::KC < p.C) fetch(c0 + G::KC);     // in flight while the matrix cores run
#pragma unroll
        for (int ty = 0; ty < TY; ++ty)
#pragma unroll
            for (int tx = 0; tx < TX; ++tx) {
                const int w_tap = (ty * TX + tx) * G::KC * BN;
                const int i_tap = ty * G::EWP + tx;
#pragma unroll
                for (int cp = 0; cp < G::KC / 2; ++cp) {
                    const float a0 = s_w[w_tap + (2 * cp) * BN + a_off[0]];
                    const float a1 = s_w[w_tap + (2 * cp) * BN + a_off[1]];
                    const float x0 = s_in[(2 * cp) * G::PLANE + i_tap + b_off[0]];
                    const float x1 = s_in[(2 * cp) * G::PLANE + i_tap + b_off[1]];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, x0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, x1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x1, acc[1][1], 0, 0, 0);
                }
            }
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: column (pixel) = lane & 31,
    // row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int64_t plane_out = (int64_t)p.OH * p.OW;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int m = wpx * 64 + pt * 32 + l31;
        const int px = m % PW, py = (m / PW) % PH, pb = m / (PW * PH);
        const int gy = gy0 + py, gx = gx0 + px, b = b0 + pb;
        if (gy >= p.GH || gx >= p.GW || b >= p.B) continue;
        const int64_t pix = (int64_t)(gy * p.osy + p.ooy) * p.OW + (gx * p.osx + p.oox);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wco * 64 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (n < p.N) {
                    float v = acc[ct][pt][r];
                    if (p.oscale) v *= p.oscale[(int64_t)b * p.N + n];
                    if (p.obias) v += p.obias[n];
                    p.out[((int64_t)b * p.N + n) * plane_out + pix] = v;
                }
            }
    }
}

template <int IS, int TY, int TX>
int launch_by_patch(ConvParams& p, hipStream_t st) {
    // patch shape from the grid width: 32x4, 16x8, 8x8x2, 4x4x8
    int pw, ph, pb;
    if (p.GW > 16) { pw = 32; ph = 4; pb = 1; }
    else if (p.GW > 8) { pw = 16; ph = 8; pb = 1; }
    else if (p.GW > 4) { pw = 8; ph = 8; pb = 2; }
    else { pw = 4; ph = 4; pb = 8; }
    p.tiles_x = (p.GW + pw - 1) / pw;
    p.tiles_y = (p.GH + ph - 1) / ph;
    p.tiles_b = (p.B + pb - 1) / pb;
    p.tiles_n = (p.N + BN - 1) / BN;
    const int64_t blocks = (int64_t)p.tiles_x * p.tiles_y * p.tiles_b * p.tiles_n;
    if (blocks <= 0) return SR_OK;
    if (blocks > 0x7FFFFFFFLL) return SR_ERANGE;
    const dim3 grid((unsigned)blocks), block(256);
    if (pw == 32) hipLaunchKernelGGL((k_conv_mfma<IS, TY, TX, 32, 4, 1>), grid, block, 0, st, p);
    else if (pw == 16) hipLaunchKernelGGL((k_conv_mfma<IS, TY, TX, 16, 8, 1>), grid, block, 0, st, p);
    else if (pw == 8) hipLaunchKernelGGL((k_conv_mfma<IS, TY, TX, 8, 8, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((k_conv_mfma<IS, TY, TX, 4, 4, 8>), grid, block, 0, st, p);
    return sr_launch_status();
}

}  // namespace

extern "C" int sr_conv2d_mfma(float* out, const float* in, const float* wt, const float* iscale,
                              const float* oscale, const float* obias, int64_t B, int64_t C,
                              int64_t N, int64_t IH, int64_t IW, int64_t OH, int64_t OW, int ksize,
                              int stride, int pad, int transposed, sr_stream_t stream) {
    if (B < 0 || C <= 0 || N <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return SR_EINVAL;
    if (B == 0) return SR_OK;
    if (!out || !in || !wt) return SR_EINVAL;
    if (IH * IW >= (1LL << 31) || OH * OW >= (1LL << 31)) return SR_ERANGE;
    hipStream_t st = sr_stream(stream);
    ConvParams p;
    p.in = in; p.wt = wt; p.iscale = iscale; p.oscale = oscale; p.obias = obias; p.out = out;
    p.B = (int)B; p.C = (int)C; p.N = (int)N;
    p.IH = (int)IH; p.IW = (int)IW; p.OH = (int)OH; p.OW = (int)OW;
    for (int i = 0; i < 9; ++i) p.wmap[i] = 0;
    if (!transposed) {
        if (OH != (IH + 2 * pad - ksize) / stride + 1 || OW != (IW + 2 * pad - ksize) / stride + 1)
            return SR_EINVAL;
        p.GH = p.OH; p.GW = p.OW;
        p.osy = p.osx = 1; p.ooy = p.oox = 0;
        p.dy0 = p.dx0 = -pad;
        for (int i = 0; i < ksize * ksize; ++i) p.wmap[i] = i;
        if (ksize == 3 && stride == 1) return launch_by_patch<1, 3, 3>(p, st);
        if (ksize == 3 && stride == 2) return launch_by_patch<2, 3, 3>(p, st);
        if (ksize == 1 && stride == 1) return launch_by_patch<1, 1, 1>(p, st);
        if (ksize == 1 && stride == 2) return launch_by_patch<2, 1, 1>(p, st);
        return SR_EINVAL;
    }
    // transposed 3x3 stride 2, no padding: out[2y + ky, 2x + kx] += in[y, x] * W[ky][kx].
    // Output phase (py, px) of grid point (j, i) = output (2j + py, 2i + px); window position ty
    // reads input row j + ty - (TY - 1) and pairs with ky = py + 2 * (TY - 1 - ty).
    if (ksize != 3 || stride != 2 || pad != 0 || OH != 2 * IH + 1 || OW != 2 * IW + 1) return SR_EINVAL;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const int TYp = py == 0 ? 2 : 1, TXp = px == 0 ? 2 : 1;
            p.GH = py == 0 ? p.IH + 1 : p.IH;
            p.GW = px == 0 ? p.IW + 1 : p.IW;
            p.osy = p.osx = 2; p.ooy = py; p.oox = px;
            p.dy0 = -(TYp - 1); p.dx0 = -(TXp - 1);
            for (int ty = 0; ty < TYp; ++ty)
                for (int tx = 0; tx < TXp; ++tx) {
                    const int ky = py + 2 * (TYp - 1 - ty), kx = px + 2 * (TXp - 1 - tx);
                    p.wmap[ty * TXp + tx] = ky * 3 + kx;
                }
            int rc;
            if (py == 0 && px == 0) rc = launch_by_patch<1, 2, 2>(p, st);
            else if (py == 0) rc = launch_by_patch<1, 2, 1>(p, st);
            else if (px == 0) rc = launch_by_patch<1, 1, 2>(p, st);
            else rc = launch_by_patch<1, 1, 1>(p, st);
            if (rc != SR_OK) return rc;
        }
    return SR_OK;
}
