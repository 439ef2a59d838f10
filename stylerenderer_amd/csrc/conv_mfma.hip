// Implicit-GEMM convolution on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// This is the contraction of the reference's ModulatedConv2d (reference layers.py:293-323), which the
// reference runs as F.conv2d / F.conv_transpose2d with groups = batch over per-sample weight copies
// ([B*Cout, Cin, k, k]: 151 MB per 512x512 layer at B = 16).  Here the modulation is moved onto the
// operands instead (mathematically identical):
//
//   out[b, n, oy, ox] = oscale[b, n] * sum_{tap, c} Wt[tap][c][n] * (iscale[b, c] * in[b, c, iy, ix])  (+ obias[n])
//
// with ONE shared weight tensor Wt [taps][C][N], the style s[b, c] multiplied onto the MFMA operand
// right after it is read from LDS, and the demodulation d[b, n] applied in the epilogue.  The same
// kernel is the data-gradient (swap iscale/oscale, flipped + transposed weights) and serves plain
// EqualConv2d layers (no scales, optional bias).
//
// Mapping (one 256-thread workgroup = 4 waves):
//   GEMM  D[n][pixel] = sum_k A[n][k] * Bm[k][pixel],  A = weights, Bm = input window; k = (tap, c).
//   tile  128 output channels x 128 pixels (a PB x PH x PW patch of the output grid), wave = 64 x 64
//         = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator VGPRs).
//   K     channel chunks of KC: the input HALO patch of the chunk is staged ONCE in LDS and every
//         tap reads it at a shifted address (9x less staging than im2col); the chunk's weights
//         [taps][KC][128] sit next to it.  Operand fetch is one conflict-free ds_read_b32 per MFMA
//         operand (lanes 0-31 = 32 consecutive pixels / channels, lanes 32-63 = the next k).
//   pipe  LDS is double buffered and filled by LDS-DMA (global_load_lds, 16 B/lane for weights and,
//         when the map is >= 32 wide and 16-byte aligned, for the halo patch too (Geo<.., V4>: aligned
//         window with LEAD extra columns); 4 B/lane otherwise): no staging VGPRs, no
//         LDS-write phase, no load result is consumed by VALU, so nothing waits on memory in front of
//         the MFMA block; chunk i+1 streams in while chunk i is on the matrix cores; ONE barrier
//         per chunk.  Border / channel-tail elements are sourced from a zero line, the style from a
//         ones line when absent — every lane always issues its DMA (branch-free).
//   out   lane (l & 31) owns a pixel, registers own channels: every store instruction writes
//         32 consecutive pixels of one channel row per half-wave (128 B segments).
// Variants by template: input stride 1 / 2, tap window (3x3, 2x2, 2x1, 1x2, 1x1) — the stride-2
// transposed convolution of the upsampling layers is run as its four output phases (interior +
// border strips) — and the patch shape for 4x4 ... 256x256 feature maps; small maps split K.
// The stride-1 3x3 case of maps >= 32 wide is normally taken by the Winograd kernel (conv_wino.hip);
// this file then serves the strided / transposed / 1x1 / small-map launches and SR_WINOGRAD=0.
#include "common.h"
#include "conv1x1_gemm.h"
#include "conv_s2_bf16x3.h"
#include "conv_wgrad_bf16x3.h"
#include "conv_wino.h"

#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

constexpr int BN = 128;    // output channels per workgroup
constexpr int BM = 128;    // pixels per workgroup

// DMA sources for masked lanes
__device__ __attribute__((aligned(16))) const float g_zero_line[4] = {0.f, 0.f, 0.f, 0.f};
__device__ __attribute__((aligned(16))) const float g_ones_line[4] = {1.f, 1.f, 1.f, 1.f};

struct ConvParams {
    const float* in;
    const float* wt;
    const float* iscale;
    const float* oscale;
    const float* obias;
    float* out;
    int B, C, N, ldw;    // ldw: row pitch of wt in floats (multiple of 4, >= N)
    int IH, IW;          // input extent
    int GH, GW;          // output grid (phase space): rows [gy_base, GH), columns [gx_base, GW)
    int gy_base, gx_base;
    int OH, OW;          // full output extent
    int osy, osx, ooy, oox;   // output coordinate = grid * os + oo
    int dy0, dx0;        // input coordinate = grid * IS + d0 + tap
    int wmap[9];         // weight slab of window position (ty, tx)
    int tiles_x, tiles_y, tiles_b, tiles_n;
    // split-K for small feature maps (few tiles, long serial K loop): slice s handles channels
    // [s * c_per_slice, (s+1) * c_per_slice) and writes raw sums to partial[s] (layout of `out`);
    // k_conv_reduce adds the slices in order and applies oscale / bias.  ks == 1: direct epilogue.
    float* partial;
    int64_t partial_floats;
    int ks, c_per_slice;
    // TAP9 launches: the taps this launch walks are tap_first + i * tap_step, i < tap_count (all nine: 0, 1, 9; the
    // border strips behind the fused kernel: row 2*IH = taps 6, 7, 8; column 2*IW = taps 2, 5, 8)
    int tap_first, tap_step, tap_count;
};

// V4: the halo window starts at a 16-byte aligned column (LEAD extra columns on the left), rows
// are padded to a multiple of 4 floats, and the whole image moves with 16-byte DMAs (a 4-byte DMA
// costs the same issue slot for a quarter of the data: measured ~3x fewer DMA instructions per
// chunk).  Needs IW % 4 == 0, a 16-byte aligned input and stride 1; otherwise the 4-byte form.
// V4 with IS == 2 ("rotating lead", ROT): the stride-2 data gradient of the up-sampling layers reads (2^k+1)-wide
// maps whose rows are NOT 16-byte aligned — but with IW = 1 and IH*IW = 1 (mod 4) the alignment of a window row is
// (channel + row + column) mod 4, so every LDS row is filled from the aligned address at or below its window start
// (lead = 0..3 floats, different per row and channel) and the operand fetch adds that lead back: its value for
// k-step (channel pair cp, tap row ty) is (L0 + 2 cp + ty) mod 4 with a per-lane constant L0 — four precomputed
// offsets, selected at compile time in the unrolled loop.  Same 16-byte DMA count as the aligned stride-1 form.
template <int IS, int TY, int TX, int PW, int PH, int PB, bool V4, bool TAP9 = false>
struct Geo {
    static constexpr bool ROT = V4 && IS == 2;
    static constexpr int NT = TY * TX;
    // channels per K chunk: ~32-36 k-steps of MFMA work per barrier whatever the window
    static constexpr int KC = NT >= 9 ? 4 : (NT >= 4 ? 8 : 16);
    static constexpr int EH = (PH - 1) * IS + TY;
    static constexpr int EW = (PW - 1) * IS + TX;
    // TAP9 with 16-byte DMAs: every tap's window starts FOUR columns left of the tile (aligned); the operand fetch adds
    // 4 + dx0 (dx0 = 0 or -1: the tap's input shift) — one staging layout for all nine taps
    static constexpr int LEAD = (V4 && !ROT) ? (TAP9 ? 4 : (TX > 1 ? 3 : 0)) : 0;
    static constexpr int EWP = ROT ? (EW + 3 + 3) / 4 * 4
                                   : (V4 ? (LEAD + EW + 3) / 4 * 4 : EW + ((EW % 2 == 0) ? 1 : 0));
    static constexpr int PLANE = PB * EH * EWP;                  // one channel of the chunk
    static constexpr int W_FLOATS = NT * KC * BN;                // [tap][c][128]
    static constexpr int W_INSTR = W_FLOATS / 256;               // 1 KiB (2 rows) per wave instruction
    // image instructions, then style instructions (V4) / one mixed 4-byte stream (!V4)
    static constexpr int IMG_INSTR = V4 ? (KC * PLANE + 255) / 256 : 0;
    static constexpr int ST_INSTR = V4 ? (KC * PB + 63) / 64 : 0;
    static constexpr int IN_FLOATS = KC * PLANE + KC * PB;       // halo planes, then the style row(s)
    static constexpr int IN_INSTR = V4 ? IMG_INSTR + ST_INSTR : (IN_FLOATS + 63) / 64;
    static constexpr int S_BASE = V4 ? W_FLOATS + IMG_INSTR * 256 : W_FLOATS + KC * PLANE;
    static constexpr int BUF = V4 ? S_BASE + ST_INSTR * 64 : W_FLOATS + IN_INSTR * 64;
    static constexpr int W_PER_WAVE = (W_INSTR + 3) / 4;
    static constexpr int IN_PER_WAVE = (IN_INSTR + 3) / 4;
    static constexpr int LDS_BYTES = 2 * BUF * 4;
};

// FAST: C is a multiple of the chunk size (no channel tail): every DMA source is then
//   pointer(chunk 0) + chunk_channel * stride   with a chunk-invariant per-lane pointer and stride
// (image lanes: one plane; style lanes: one float; border / absent-style lanes: 0 from the zero / ones
// line), i.e. one v_mad_u64_u32 per DMA instruction instead of a dozen selects and scalar branches.  Every
// instruction issued on the SIMD costs the matrix pipe a slot.
// TAP9 (1x1 window only): the stride-2 TRANSPOSED 3x3 convolution of a small map as nine shifted 1x1 convolutions in ONE
// launch.  out[2j + py, 2i + px] = sum over taps (ky, kx) = (py, px) mod 2 of in[j - (ky >> 1), i - (kx >> 1)] * W[ky][kx]:
// tap (ky, kx) is a 1x1 convolution over the phase grid with the input shifted by (ky >> 1, kx >> 1).  blockIdx
// enumerates (tap, tile, K slice); every workgroup walks the same K = channels-of-its-slice, so the launch is balanced
// whatever the phase (the four per-phase launches carry 4 / 2 / 2 / 1 taps) and has 9x the workgroups of one phase —
// what a 4^2 .. 32^2 map at batch 1 .. 4 needs to fill 256 CUs.  Each workgroup stores its raw sums to
// partial[slice * 9 + tap] (common (IH+1) x (IW+1) grid); k_convt_tap_reduce adds the taps of each output's phase and
// the slices in a fixed order and applies scale / bias.  Replaces 8 .. 18 launches per layer (four phases x {interior, two
// border strips} + their split-K reductions) by two.
template <int IS, int TY, int TX, int PW, int PH, int PB, bool V4, bool FAST, bool TAP9 = false>
__global__ __launch_bounds__(256) void k_conv_mfma(const ConvParams p) {
    using G = Geo<IS, TY, TX, PW, PH, PB, V4, TAP9>;
    static_assert(PW * PH * PB == BM, "patch must hold 128 pixels");
    static_assert(!TAP9 || (IS == 1 && TY == 1 && TX == 1), "tap-split mode is a shifted 1x1 convolution");
    extern __shared__ __attribute__((aligned(16))) float smem[];     // the ONLY LDS object

    // ---- tile decode; workgroups that share an input patch (different n tiles) and neighbouring
    // patches are numbered consecutively and kept on one XCD (bid % 8 is the XCD): chunked remap.
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg / SR_NUM_XCD, r = nwg % SR_NUM_XCD, xcd = bid % SR_NUM_XCD;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / SR_NUM_XCD;
    }
    // geometry of this workgroup: the launch's, or (TAP9) its tap's
    int tap = 0, g_dy0 = p.dy0, g_dx0 = p.dx0, g_gh = p.GH, g_gw = p.GW, g_ooy = p.ooy, g_oox = p.oox, g_slab = p.wmap[0];
    int tap_local = 0;
    if (TAP9) {
        tap_local = bid % p.tap_count;      // the taps of a tile are neighbours: one input patch in L2
        bid /= p.tap_count;
        tap = p.tap_first + tap_local * p.tap_step;
        const int ky = tap / 3, kx = tap - 3 * ky;
        g_dy0 = -(ky >> 1); g_dx0 = -(kx >> 1);
        g_ooy = ky & 1; g_oox = kx & 1;
        // the phase's extent, clipped to the launch's region [gy_base, GH) x [gx_base, GW)
        g_gh = min(g_ooy ? p.IH : p.IH + 1, p.GH); g_gw = min(g_oox ? p.IW : p.IW + 1, p.GW);
        g_slab = tap;
    }
    const int n_t = bid % p.tiles_n;
    bid /= p.tiles_n;
    const int tx_i = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty_i = bid % p.tiles_y;
    bid /= p.tiles_y;
    const int tb_i = bid % p.tiles_b;
    const int slice = bid / p.tiles_b;
    const int c_beg = slice * p.c_per_slice;
    const int c_end = min(p.C, c_beg + p.c_per_slice);
    const int n0 = n_t * BN, gy0 = p.gy_base + ty_i * PH, gx0 = p.gx_base + tx_i * PW, b0 = tb_i * PB;
    const int iy0 = gy0 * IS + g_dy0, ix0 = gx0 * IS + g_dx0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wco = wave & 1, wpx = wave >> 1;

    // ---- per-lane LDS offsets of the MFMA operands (inside one buffer)
    int a_off[2], b_off[2], s_off[2];
    int b_rot[2][4];               // ROT: b_off plus the lead of k-step class (2 cp + ty) mod 4
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        a_off[t] = half * BN + wco * 64 + t * 32 + l31;
        const int m = wpx * 64 + t * 32 + l31;
        const int px = m % PW, py = (m / PW) % PH, pb = m / (PW * PH);
        b_off[t] = G::W_FLOATS + half * G::PLANE + (pb * G::EH + py * IS) * G::EWP + px * IS + G::LEAD +
                   ((TAP9 && V4) ? g_dx0 : 0);
        s_off[t] = G::S_BASE + pb * G::KC + half;                        // style of (sample, channel)
        const int l0 = half + iy0 + py * IS + ix0;
#pragma unroll
        for (int k = 0; k < 4; ++k) b_rot[t][k] = b_off[t] + ((l0 + k) & 3);
    }

    // ---- DMA descriptors (chunk-invariant part), a few registers per lane
    // weights: instruction j moves LDS floats [256 j, 256 j + 256) = rows (2j, 2j+1) of [tap*KC + c][128]
    int w_src[G::W_PER_WAVE];      // element offset into wt for c0 = 0
    int w_cl[G::W_PER_WAVE];       // chunk-local channel (for the channel-tail mask), -1 = no instr
#pragma unroll
    for (int i = 0; i < G::W_PER_WAVE; ++i) {
        const int j = wave + 4 * i;
        const int row = 2 * j + half;
        const int t = row / G::KC, c = row % G::KC;
        const int col = min(n0 + l31 * 4, p.ldw - 4);
        w_cl[i] = (j < G::W_INSTR) ? c : -1;
        int slab = 0;       // p.wmap[t] via selects (a runtime index would spill the kernarg struct)
#pragma unroll
        for (int k = 0; k < G::NT; ++k)
            if (t == k) slab = p.wmap[k];
        if (TAP9) slab = g_slab;
        w_src[i] = (slab * p.C + c) * p.ldw + col;
    }
    // input: !V4: instruction j moves LDS floats [64 j, 64 j + 64) of the halo image, then the
    // style rows;  V4: instruction j < IMG_INSTR moves floats [256 j, 256 j + 256) (4 per lane, never
    // straddling a row or the image border), the remaining ones move the style rows.
    int i_src[G::IN_PER_WAVE];     // element offset for (b0, c0) = (0, 0); -1 zero line, -2 style
    int i_cl[G::IN_PER_WAVE];      // chunk-local channel | sample << 8
#pragma unroll
    for (int i = 0; i < G::IN_PER_WAVE; ++i) {
        const int j = wave + 4 * i;
        int src = -1, c = 0, pb = 0;
        if (V4) {
            const int f = (j * 64 + lane) * 4;
            if (j < G::IMG_INSTR && f < G::KC * G::PLANE) {
                c = f / G::PLANE;
                const int q = f % G::PLANE;
                const int cola = q % G::EWP, r = (q / G::EWP) % G::EH;
                pb = q / (G::EWP * G::EH);
                const int gy = iy0 + r;
                int gx = ((TAP9 && V4) ? gx0 : ix0) - G::LEAD + cola;
                bool ok = gy >= 0 && gy < p.IH && b0 + pb < p.B;
                if (G::ROT) {
                    // aligned group at or below the window start of this (channel, row); it may run into the
                    // neighbouring row — those columns are never fetched as operands
                    gx = ix0 - ((c * (p.IH * p.IW) + gy * p.IW + ix0) & 3) + cola;
                } else {
                    ok = ok && gx >= 0 && gx + 3 < p.IW;
                }
                if (ok) src = (pb * p.C + c) * p.IH * p.IW + gy * p.IW + gx;
            } else if (j >= G::IMG_INSTR && j < G::IN_INSTR) {
                const int e = (j - G::IMG_INSTR) * 64 + lane;
                if (e < G::KC * PB) {
                    pb = e / G::KC;
                    c = e % G::KC;
                    src = -2;
                }
            }
        } else {
            const int f = j * 64 + lane;
            if (j < G::IN_INSTR && f < G::KC * G::PLANE) {
                c = f / G::PLANE;
                const int q = f % G::PLANE;
                const int colp = q % G::EWP, r = (q / G::EWP) % G::EH;
                pb = q / (G::EWP * G::EH);
                const int gy = iy0 + r, gx = ix0 + colp;
                if (colp < G::EW && gy >= 0 && gy < p.IH && gx >= 0 && gx < p.IW && b0 + pb < p.B)
                    src = (pb * p.C + c) * p.IH * p.IW + gy * p.IW + gx;
            } else if (j < G::IN_INSTR && f < G::IN_FLOATS) {
                const int e = f - G::KC * G::PLANE;
                pb = e / G::KC;
                c = e % G::KC;
                src = -2;
            }
        }
        i_src[i] = src;
        i_cl[i] = c | (pb << 8);
    }
    const float* in_base = p.in + (int64_t)b0 * p.C * p.IH * p.IW;
    const int plane_in = p.IH * p.IW;

    const char* f_wp[G::W_PER_WAVE];
    const char* f_ip[G::IN_PER_WAVE];
    unsigned f_is[G::IN_PER_WAVE];
    if (FAST) {
#pragma unroll
        for (int i = 0; i < G::W_PER_WAVE; ++i) f_wp[i] = reinterpret_cast<const char*>(p.wt + w_src[i]);
#pragma unroll
        for (int i = 0; i < G::IN_PER_WAVE; ++i) {
            const int c = i_cl[i] & 255, pb = i_cl[i] >> 8;
            const char* ptr = reinterpret_cast<const char*>(g_zero_line);
            unsigned str = 0;
            if (i_src[i] >= 0) {
                ptr = reinterpret_cast<const char*>(in_base + i_src[i]);
                str = (unsigned)plane_in * 4u;
            } else if (i_src[i] == -2) {
                const bool ok = p.iscale && b0 + pb < p.B;
                ptr = ok ? reinterpret_cast<const char*>(p.iscale + (int64_t)(b0 + pb) * p.C + c)
                         : reinterpret_cast<const char*>(g_ones_line);
                str = ok ? 4u : 0u;
            }
            f_ip[i] = ptr;
            f_is[i] = str;
        }
    }
    const unsigned w_chunk_stride = (unsigned)p.ldw * 4u;

    auto dma = [&](int c0, int buf) {
        float* dst = smem + buf * G::BUF;
        if (FAST) {
#pragma unroll
            for (int i = 0; i < G::W_PER_WAVE; ++i) {
                const int j = wave + 4 * i;
                if (j < G::W_INSTR)
                    __builtin_amdgcn_global_load_lds((gptr_t)(f_wp[i] + (uint64_t)(unsigned)c0 * w_chunk_stride),
                                                     (lptr_t)(dst + j * 256), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < G::IN_PER_WAVE; ++i) {
                const int j = wave + 4 * i;
                if (j < G::IN_INSTR) {
                    const char* src = f_ip[i] + (uint64_t)(unsigned)c0 * f_is[i];
                    if (V4 && j < G::IMG_INSTR)
                        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + G::W_FLOATS + j * 256), 16, 0, 0);
                    else if (V4)
                        __builtin_amdgcn_global_load_lds((gptr_t)src,
                                                         (lptr_t)(dst + G::S_BASE + (j - G::IMG_INSTR) * 64), 4, 0, 0);
                    else
                        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + G::W_FLOATS + j * 64), 4, 0, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < G::W_PER_WAVE; ++i) {
            const int j = wave + 4 * i;
            if (j < G::W_INSTR) {       // wave-uniform
                const float* src = (c0 + w_cl[i] < p.C) ? p.wt + w_src[i] + (int64_t)c0 * p.ldw : g_zero_line;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + j * 256), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < G::IN_PER_WAVE; ++i) {
            const int j = wave + 4 * i;
            if (j < G::IN_INSTR) {      // wave-uniform
                const int c = i_cl[i] & 255, pb = i_cl[i] >> 8;
                const bool c_ok = c0 + c < p.C;
                const float* src = g_zero_line;
                if (i_src[i] >= 0 && c_ok) src = in_base + i_src[i] + (int64_t)c0 * plane_in;
                if (i_src[i] == -2)
                    src = (p.iscale && c_ok && b0 + pb < p.B) ? p.iscale + (int64_t)(b0 + pb) * p.C + c0 + c
                                                              : g_ones_line;
                if (V4 && j < G::IMG_INSTR)
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + G::W_FLOATS + j * 256), 16, 0, 0);
                else if (V4)
                    __builtin_amdgcn_global_load_lds((gptr_t)src,
                                                     (lptr_t)(dst + G::S_BASE + (j - G::IMG_INSTR) * 64), 4, 0, 0);
                else
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + G::W_FLOATS + j * 64), 4, 0, 0);
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    dma(c_beg, 0);
    int buf = 0;
    for (int c0 = c_beg; c0 < c_end; c0 += G::KC) {
        // (a) this wave's DMA of chunk c0 has landed (the compiler drains vmcnt before the barrier),
        // (b) every wave is done reading the buffer the next DMA overwrites
        __syncthreads();
        if (c0 + G::KC < c_end) dma(c0 + G::KC, buf ^ 1);
        const float* sb = smem + buf * G::BUF;
#pragma unroll
        for (int cp = 0; cp < G::KC / 2; ++cp) {
            // style of channel (2 cp + half): onto the weight operand when the tile holds one
            // sample, onto the input operand when it holds several
            const float sc0 = sb[s_off[0] + 2 * cp];
            const float sc1 = (PB > 1) ? sb[s_off[1] + 2 * cp] : sc0;
#pragma unroll
            for (int ty = 0; ty < TY; ++ty)
#pragma unroll
                for (int tx = 0; tx < TX; ++tx) {
                    const int w_tap = ((ty * TX + tx) * G::KC + 2 * cp) * BN;
                    const int i_tap = (2 * cp) * G::PLANE + ty * G::EWP + tx;
                    float a0 = sb[w_tap + a_off[0]];
                    float a1 = sb[w_tap + a_off[1]];
                    float x0 = sb[i_tap + (G::ROT ? b_rot[0][(2 * cp + ty) & 3] : b_off[0])];
                    float x1 = sb[i_tap + (G::ROT ? b_rot[1][(2 * cp + ty) & 3] : b_off[1])];
                    if (PB == 1) { a0 *= sc0; a1 *= sc0; }
                    else { x0 *= sc0; x1 *= sc1; }
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, x0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, x1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x1, acc[1][1], 0, 0, 0);
                }
        }
        buf ^= 1;
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: column (pixel) = lane & 31,
    // row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int64_t plane_out = (int64_t)p.OH * p.OW;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int m = wpx * 64 + pt * 32 + l31;
        const int px = m % PW, py = (m / PW) % PH, pb = m / (PW * PH);
        const int gy = gy0 + py, gx = gx0 + px, b = b0 + pb;
        if (gy >= g_gh || gx >= g_gw || b >= p.B) continue;
        const int64_t pix = (int64_t)(gy * p.osy + g_ooy) * p.OW + (gx * p.osx + g_oox);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wco * 64 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (n < p.N) {
                    float v = acc[ct][pt][r];
                    if (TAP9) {
                        // [slice * tap_count + tap][b][n][row][col] over the launch's region
                        const int rw = p.GW - p.gx_base, rh = p.GH - p.gy_base;
                        p.partial[((((int64_t)slice * p.tap_count + tap_local) * p.B + b) * p.N + n) * (rh * rw) +
                                  (gy - p.gy_base) * rw + (gx - p.gx_base)] = v;
                    } else if (p.ks > 1) {
                        // compact slab of THIS launch's region: [slice][b][n][row][col]
                        const int rw = p.GW - p.gx_base, rh = p.GH - p.gy_base;
                        p.partial[(((int64_t)slice * p.B + b) * p.N + n) * (rh * rw) +
                                  (gy - p.gy_base) * rw + (gx - p.gx_base)] = v;
                    } else {
                        if (p.oscale) v *= p.oscale[(int64_t)b * p.N + n];
                        if (p.obias) v += p.obias[n];
                        p.out[((int64_t)b * p.N + n) * plane_out + pix] = v;
                    }
                }
            }
    }
}

template <int IS, int TY, int TX, int PW, int PH, int PB, bool V4, bool FAST, bool TAP9 = false>
int launch_fast(const ConvParams& p, dim3 grid, hipStream_t st) {
    using G = Geo<IS, TY, TX, PW, PH, PB, V4, TAP9>;
    auto kern = k_conv_mfma<IS, TY, TX, PW, PH, PB, V4, FAST, TAP9>;
    static bool configured = false;     // opt in to > 64 KiB of dynamic LDS once per variant
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        configured = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), G::LDS_BYTES, st, p);
    return sr_launch_status();
}

template <int IS, int TY, int TX, int PW, int PH, int PB, bool V4, bool TAP9 = false>
int launch_one(const ConvParams& p, dim3 grid, hipStream_t st) {
    using G = Geo<IS, TY, TX, PW, PH, PB, V4, TAP9>;
    // no channel tail in any K slice -> chunk-invariant DMA descriptors
    if (p.C % G::KC == 0 && p.c_per_slice % G::KC == 0)
        return launch_fast<IS, TY, TX, PW, PH, PB, V4, true, TAP9>(p, grid, st);
    return launch_fast<IS, TY, TX, PW, PH, PB, V4, false, TAP9>(p, grid, st);
}

// ------------------------------------------------------------------------------------------------
// Stride-2 transposed 3x3 convolution, all four output phases in ONE workgroup (interior of the map).
//
// out[2j + py, 2i + px] = sum over the taps (ky, kx) with ky = py (mod 2), kx = px (mod 2) of
// in[j - (ky >> 1), i - (kx >> 1)] * W[ky][kx]: nine (tap -> phase, input shift) pairs over only four
// distinct input shifts.  Run as four launches (above) every phase re-stages the same halo patch and
// the 1- and 2-tap phases are operand-fetch bound (80 TFLOP/s).  Here a wave keeps the 2x2 MFMA tiles
// of ALL four phases (16 accumulator tiles = 256 registers, one wave per SIMD), so per k-step the 36
// MFMAs share 8 input operands and 18 weight operands, and the halo patch is staged once.
// Workgroup = 32x4 input positions (-> 64x8 output pixels) x 128 output channels, K chunks of 4 input
// channels, double-buffered LDS-DMA as in k_conv_mfma.  The last output row / column (grid row H,
// column W) are left to the strip launches of the per-phase kernel.
struct TFused {
    static constexpr int KC = 8;
    static constexpr int PW = 32, PH = 4;
    static constexpr int LEAD = 3;                          // halo columns start at i0 - 4 (aligned); i0 - 1 is column 3
    static constexpr int EWP = 36;                          // LEAD + 33 columns
    static constexpr int EH = PH + 1;
    static constexpr int PLANE = EH * EWP;                  // 180
    static constexpr int W_FLOATS = 9 * KC * BN;            // [tap][c][128] = 9216
    static constexpr int W_INSTR = W_FLOATS / 256;          // 36 = 9 per wave
    static constexpr int I_INSTR = (KC * PLANE / 4 + 63) / 64;   // 6
    static constexpr int I_FLOATS = I_INSTR * 256;          // 1536
    static constexpr int BUF = W_FLOATS + I_FLOATS;         // 10752 floats = 42 KB
    static constexpr int W_PER_WAVE = 9, I_PER_WAVE = 2;    // 36 / 8 instruction slots (surplus -> pad zone)
    static constexpr int PAD = 2 * BUF;
    static constexpr int STY = PAD + 4 * 256;
};

#ifdef CONVT_TIMING
// debug build only (scripts/build_variant.sh): per-workgroup time stamps of wave 0
__device__ long long g_convt_stamps[8 * 16384];
#define CONVT_STAMP(i) do { if (threadIdx.x == 0) { g_convt_stamps[(blockIdx.x & 16383) * 8 + (i)] = wall_clock64(); } } while (0)
#else
#define CONVT_STAMP(i)
#endif

// KS: K slices (round 5).  With few tiles and a long channel loop (the 32^2 x 512-channel layer at batch 4: 128
// workgroups walking 64 chunks each) slice s of p.ks walks the channels [s * c_per_slice, (s + 1) * c_per_slice): the
// same loop over k = 0 .. n - 1 with the bases of the two buffer resources and of the style row moved, raw sums to
// partial[s] in the layout of `out`; k_convt_fused_reduce adds the slices in order and applies scale / bias.
template <bool KS>
__global__ __launch_bounds__(256) void k_convt_fused(const ConvParams p) {
#if __HIP_DEVICE_COMPILE__   // buffer-resource builtins: device pass only (the host pass needs just the stub)
    using G = TFused;
    CONVT_STAMP(0);
#ifdef CONVT_TIMING
    const long long convt_c0 = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sty = smem + G::STY;

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
    bid /= p.tiles_y;
    const int b = KS ? bid % p.B : bid;
    const int slice = KS ? bid / p.B : 0;
    const int c_beg = KS ? slice * p.c_per_slice : 0;          // first channel of this workgroup's K range
    const int c_cnt = KS ? p.c_per_slice : p.C;
    const int n0 = n_t * BN, j0 = ty_i * G::PH, i0 = tx_i * G::PW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wco = wave & 1, wpx = wave >> 1;

    // ---- DMA descriptors: byte offsets inside the weight tensor / this sample for chunk 0; lanes with nothing to
    // fetch (outside the image, surplus instructions) point beyond the buffer and receive zeros.  The chunk part of
    // every address is the scalar soffset of the buffer load: no vector ALU work per DMA instruction.
    constexpr int OOB = 0x7FFFFFF0;
    int w_off[G::W_PER_WAVE];
#pragma unroll
    for (int i = 0; i < G::W_PER_WAVE; ++i) {
        const int j = wave + 4 * i;
        const int row = 2 * j + half;                       // [tap * KC + c]
        const int t = row / G::KC, c = row % G::KC;
        const int col = min(n0 + l31 * 4, p.ldw - 4);
        w_off[i] = j < G::W_INSTR ? ((t * p.C + c) * p.ldw + col) * 4 : OOB;
    }
    int i_off[G::I_PER_WAVE];
#pragma unroll
    for (int i = 0; i < G::I_PER_WAVE; ++i) {
        const int j = wave + 4 * i;
        const int f = (j * 64 + lane) * 4;
        int off = OOB;
        if (j < G::I_INSTR && f < G::KC * G::PLANE) {
            const int c = f / G::PLANE, q = f % G::PLANE;
            const int r = q / G::EWP, cola = q % G::EWP;
            const int gy = j0 - 1 + r, gx = i0 - 4 + cola;
            if (gy >= 0 && gy < p.IH && gx >= 0 && gx + 3 < p.IW) off = ((c * p.IH + gy) * p.IW + gx) * 4;
        }
        i_off[i] = off;
    }
    const int plane_in = p.IH * p.IW;
    const __amdgpu_buffer_rsrc_t r_w = uniform_rsrc(p.wt + (int64_t)c_beg * p.ldw, (9 * p.C - c_beg) * p.ldw * 4);
    const __amdgpu_buffer_rsrc_t r_in = uniform_rsrc(p.in + ((int64_t)b * p.C + c_beg) * plane_in, c_cnt * plane_in * 4);
    const int nchunks = c_cnt / G::KC;

    // DMA instruction i (0 .. 10) of chunk k (clamped to the last chunk: a surplus fetch lands in a free buffer)
    auto dma1 = [&](int k, int i) {
        const int kc = min(k, nchunks - 1), c0 = kc * G::KC;
        float* dst = smem + (k & 1) * G::BUF;
        if (i < G::I_PER_WAVE) {                             // the halo patch first: it may come from HBM
            const int j = wave + 4 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                r_in, (lptr_t)(j < G::I_INSTR ? dst + G::W_FLOATS + j * 256 : smem + G::PAD + wave * 256), 16, i_off[i],
                c0 * plane_in * 4, 0, 0);
        } else {                                             // then the (L2-resident) weights
            const int ii = i - G::I_PER_WAVE, j = wave + 4 * ii;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lptr_t)(j < G::W_INSTR ? dst + j * 256 : smem + G::PAD + wave * 256),
                                                     16, w_off[ii], c0 * p.ldw * 4, 0, 0);
        }
    };
    constexpr int N_DMA = G::W_PER_WAVE + G::I_PER_WAVE;    // 11 per wave and chunk

    // accumulators: [phase py*2+px][channel tile][pixel tile]
    f32x16 acc[4][2][2];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ph][i][j][r] = 0.0f;

    int a_off[2], b_off[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        a_off[t] = half * BN + wco * 64 + t * 32 + l31;
        const int py = wpx * 2 + t;                          // pixel tile t = input row py of the patch
        b_off[t] = G::W_FLOATS + half * G::PLANE + (py + 1) * G::EWP + l31 + G::LEAD + 1;
    }

    // ---- pipeline.  One wave per SIMD: nobody else hides LDS latency or DMA issue, so
    //  * the 26 operands of k-step g + 1 are fetched (and the 8 input operands style-scaled) under the 36 MFMAs
    //    of k-step g — across chunk boundaries too: the chunk barrier sits in front of the LAST k-step of a
    //    chunk, whose operands are already in registers, and the first k-step of the next chunk is fetched
    //    under it (no MFMA ever waits for an LDS round trip behind a barrier);
    //  * the 11 DMA instructions of a chunk are spread over three k-steps, one per 8 MFMAs.
    // Chunk k + 1 is complete at the barrier of chunk k (issued a whole chunk earlier); chunk k + 2 then goes
    // into the buffer of chunk k, which nobody reads any more.
#pragma unroll
    for (int i = 0; i < N_DMA; ++i) dma1(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma1(1, i);
    for (int c = tid; c < c_cnt; c += 256) sty[c] = p.iscale ? p.iscale[(int64_t)b * p.C + c_beg + c] : 1.0f;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();

    float A[2][9][2], X[2][4][2];                        // [set][tap][channel tile], [set][dy*2+dx][pixel tile]
    float sc_next;
    auto load_set = [&](const float* sb, int c0, int cp, int set) {
#pragma unroll
        for (int sh = 0; sh < 4; ++sh)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                X[set][sh][t] = sb[(2 * cp) * G::PLANE + b_off[t] - (sh >> 1) * G::EWP - (sh & 1)];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int t = 0; t < 2; ++t) A[set][tap][t] = sb[(tap * G::KC + 2 * cp) * BN + a_off[t]];
        sc_next = sty[c0 + 2 * cp + half];
    };
    load_set(smem, 0, 0, 0);
#pragma unroll
    for (int sh = 0; sh < 4; ++sh) { X[0][sh][0] *= sc_next; X[0][sh][1] *= sc_next; }
    CONVT_STAMP(1);

    for (int k = 0; k < nchunks; ++k) {
        const float* sb = smem + (k & 1) * G::BUF;
        const float* sbn = smem + ((k + 1) & 1) * G::BUF;
        const int c0 = k * G::KC, c0n = min(k + 1, nchunks - 1) * G::KC;
#pragma unroll
        for (int cp = 0; cp < G::KC / 2; ++cp) {
            const int cur = cp & 1, nxt = cur ^ 1;
            if (cp == G::KC / 2 - 1) {
                // chunk k + 1 landed (every wave drained its own DMAs), all reads of chunk k retired
                __builtin_amdgcn_s_waitcnt(0x0F70);
                __syncthreads();
                load_set(sbn, c0n, 0, nxt);
            } else {
                load_set(sb, c0, cp + 1, nxt);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int tap = ky * 3 + kx, ph = (ky & 1) * 2 + (kx & 1), sh = (ky >> 1) * 2 + (kx >> 1);
                    acc[ph][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[cur][tap][0], X[cur][sh][0], acc[ph][0][0], 0, 0, 0);
                    acc[ph][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[cur][tap][0], X[cur][sh][1], acc[ph][0][1], 0, 0, 0);
                    acc[ph][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[cur][tap][1], X[cur][sh][0], acc[ph][1][0], 0, 0, 0);
                    acc[ph][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[cur][tap][1], X[cur][sh][1], acc[ph][1][1], 0, 0, 0);
                    // style onto the 8 input operands of the next k-step (not its 18 weights): one multiply per
                    // tap from the second tap on (eight MFMAs cover the LDS latency of the fetch above)
                    if (tap >= 1) X[nxt][(tap - 1) >> 1][(tap - 1) & 1] *= sc_next;
                    // DMA: chunk k + 2 starts behind the barrier (4 instructions), the rest of chunk k + 1 follows in
                    // the first two k-steps of the next iteration... seen from this iteration: cp 0 / 1 finish
                    // chunk k + 1, the last k-step starts chunk k + 2
                    if ((tap & 1) == 1) {
                        const int u = tap >> 1;                               // 0 .. 3
                        if (cp == 0) dma1(k + 1, 4 + u);
                        else if (cp == 1 && 8 + u < N_DMA) dma1(k + 1, 8 + u);
                        else if (cp == G::KC / 2 - 1) dma1(k + 2, u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    CONVT_STAMP(2);

    // ---- epilogue: lane = input column i, registers = channels; phases (py, 0) and (py, 1) are the
    // neighbouring output columns 2i, 2i + 1 of output row 2j + py
    const int64_t plane_out = (int64_t)p.OH * p.OW;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int j = j0 + wpx * 2 + pt, i = i0 + l31;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wco * 64 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (n < p.N) {
                    const float os = (!KS && p.oscale) ? p.oscale[(int64_t)b * p.N + n] : 1.0f;
                    const float ob = (!KS && p.obias) ? p.obias[n] : 0.0f;
                    float* o = (KS ? p.partial + (int64_t)slice * p.B * p.N * plane_out : p.out) +
                               ((int64_t)b * p.N + n) * plane_out + (int64_t)(2 * j) * p.OW + 2 * i;
                    o[0] = acc[0][ct][pt][r] * os + ob;
                    o[1] = acc[1][ct][pt][r] * os + ob;
                    o[p.OW] = acc[2][ct][pt][r] * os + ob;
                    o[p.OW + 1] = acc[3][ct][pt][r] * os + ob;
                }
            }
    }
    CONVT_STAMP(3);
#ifdef CONVT_TIMING
    if (threadIdx.x == 0) g_convt_stamps[(blockIdx.x & 16383) * 8 + 5] = clock64() - convt_c0;
#endif
#endif
}

#ifdef CONVT_TIMING
extern "C" int sr_debug_convt_stamps(long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_convt_stamps), (size_t)n * sizeof(long long));
}
#endif

bool convt_fused_eligible(const ConvParams& p) {
    // buffer addressing: byte offsets inside one sample / the weight tensor below 2^31 - 16
    if ((int64_t)p.C * p.IH * p.IW >= (1LL << 29) - 4 || 9LL * p.C * p.ldw >= (1LL << 29) - 4) return false;
    return p.IH % TFused::PH == 0 && p.IW % TFused::PW == 0 && p.C % TFused::KC == 0 && p.C <= 2048 &&
           (reinterpret_cast<uintptr_t>(p.in) & 15) == 0;
}

// interior outputs (rows < 2 IH, columns < 2 IW) = the slices of k_convt_fused<true> added in order, then scale / bias
__global__ __launch_bounds__(256) void k_convt_fused_reduce(const ConvParams p) {
    const int iw2 = 2 * p.IW, ih2 = 2 * p.IH;
    const int64_t plane_out = (int64_t)p.OH * p.OW, planes = (int64_t)p.B * p.N;
    const int64_t total = planes * ih2 * (iw2 / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / ((int64_t)ih2 * (iw2 / 4));
        const int q = (int)(i - row * (int64_t)ih2 * (iw2 / 4));
        const int Y = q / (iw2 / 4), X = (q - Y * (iw2 / 4)) * 4;
        const int64_t at = row * plane_out + (int64_t)Y * p.OW + X;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int s = 0; s < p.ks; ++s) {
            const float* src = p.partial + (int64_t)s * planes * plane_out + at;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += src[k];
        }
        const float os = p.oscale ? p.oscale[row] : 1.0f, ob = p.obias ? p.obias[row % p.N] : 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) p.out[at + k] = acc[k] * os + ob;
    }
}

// K slices of the fused kernel: the smallest of 2 / 4 that gives >= 192 workgroups with whole chunks and >= 64 channels per slice
int convt_fused_slices(int64_t fused_blocks, int C) {
    for (int ks = 2; ks <= 4; ks *= 2)
        if (C % (ks * TFused::KC) == 0 && C / ks >= 64 && fused_blocks * ks >= 192) return ks;
    return 1;
}

int64_t convt_fused_split_floats(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW) {
    const int64_t blocks = (IW / TFused::PW) * (IH / TFused::PH) * ((N + BN - 1) / BN) * B;
    if (IW % TFused::PW || IH % TFused::PH || blocks >= 192 || C <= 256) return 0;
    const int ks = convt_fused_slices(blocks, (int)C);
    return ks > 1 ? ks * B * N * (2 * IH + 1) * (2 * IW + 1) : 0;
}

int launch_convt_fused(ConvParams p, hipStream_t st) {
    p.tiles_x = p.IW / TFused::PW;
    p.tiles_y = p.IH / TFused::PH;
    p.tiles_n = (p.N + BN - 1) / BN;
    const int64_t blocks = (int64_t)p.tiles_x * p.tiles_y * p.tiles_n * p.B * p.ks;
    if (blocks > 0x7FFFFFFFLL) return SR_ERANGE;
    if (p.ks > 1) {
        const int lds = (TFused::STY + p.c_per_slice) * 4;
        hipLaunchKernelGGL(k_convt_fused<true>, dim3((unsigned)blocks), dim3(256), lds, st, p);
        const int64_t total = (int64_t)p.B * p.N * 2 * p.IH * (2 * p.IW / 4);
        hipLaunchKernelGGL(k_convt_fused_reduce, dim3(sr_stream_grid(total, 256)), dim3(256), 0, st, p);
        return sr_launch_status();
    }
    const int lds = (TFused::STY + p.C) * 4;
    hipLaunchKernelGGL(k_convt_fused<false>, dim3((unsigned)blocks), dim3(256), lds, st, p);
    return sr_launch_status();
}

void patch_shape(int GW, int& pw, int& ph, int& pb) {
    // patch shape from the grid width: 32x4, 16x8, 8x8x2, 4x4x8
    if (GW > 16) { pw = 32; ph = 4; pb = 1; }
    else if (GW > 8) { pw = 16; ph = 8; pb = 1; }
    else if (GW > 4) { pw = 8; ph = 8; pb = 2; }
    else { pw = 4; ph = 4; pb = 8; }
}

// Number of K slices for an output grid: 1 when the grid alone fills the chip, otherwise enough
// slices for ~2 workgroups per CU while keeping >= 32 channels (a multiple of 16) per slice.
void choose_split(int GH, int GW, int B, int N, int C, int& ks, int& c_per_slice) {
    int pw, ph, pb;
    patch_shape(GW, pw, ph, pb);
    const int64_t blocks = (int64_t)((GW + pw - 1) / pw) * ((GH + ph - 1) / ph) * ((B + pb - 1) / pb) *
                           ((N + BN - 1) / BN);
    ks = 1;
    c_per_slice = (C + 15) / 16 * 16;
    if (blocks >= SR_NUM_CU || C < 64) return;
    int want = (int)((2 * SR_NUM_CU + blocks - 1) / blocks);
    if (want > 16) want = 16;
    int per = (C + want - 1) / want;
    per = (per + 15) / 16 * 16;
    if (per < 32) per = 32;
    c_per_slice = per;
    ks = (C + per - 1) / per;
}

// Sums the K slices of one launch's region in a fixed order (deterministic) and writes the
// region's outputs with oscale / bias applied.
__global__ __launch_bounds__(256) void k_conv_reduce(const ConvParams p, int rh, int rw) {
    const int64_t region = (int64_t)rh * rw, total = (int64_t)p.B * p.N * region;
    const int64_t plane_out = (int64_t)p.OH * p.OW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        float acc = 0.0f;
#pragma unroll 8
        for (int s = 0; s < p.ks; ++s) acc += p.partial[s * total + i];    // loads in flight, fixed order
        const int64_t row = i / region;                                  // b * N + n
        const int q = (int)(i - row * region);
        const int gy = p.gy_base + q / rw, gx = p.gx_base + q % rw;
        if (p.oscale) acc *= p.oscale[row];
        if (p.obias) acc += p.obias[row % p.N];
        p.out[row * plane_out + (int64_t)(gy * p.osy + p.ooy) * p.OW + (gx * p.osx + p.oox)] = acc;
    }
}

// Tap-split transposed convolution (k_conv_mfma<..., TAP9>): output (Y, X) of phase (Y & 1, X & 1) at grid point
// (Y >> 1, X >> 1) is the sum of its phase's taps over all K slices, added in a fixed order (slice-major, taps ascending).
__global__ __launch_bounds__(256) void k_convt_tap_reduce(const ConvParams p) {
    const int rh = p.GH, rw = p.GW;
    const int64_t region = (int64_t)rh * rw, plane_out = (int64_t)p.OH * p.OW;
    const int64_t total = (int64_t)p.B * p.N * plane_out, slab = (int64_t)p.B * p.N * region;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / plane_out;
        const int q = (int)(i - row * plane_out);
        const int Y = q / p.OW, X = q - Y * p.OW;
        const int py = Y & 1, px = X & 1;
        const int64_t at = row * region + (int64_t)(Y >> 1) * rw + (X >> 1);
        float acc = 0.0f;
        for (int s = 0; s < p.ks; ++s)
            for (int ky = py; ky < 3; ky += 2)
                for (int kx = px; kx < 3; kx += 2) acc += p.partial[((int64_t)s * 9 + ky * 3 + kx) * slab + at];
        if (p.oscale) acc *= p.oscale[row];
        if (p.obias) acc += p.obias[row % p.N];
        p.out[i] = acc;
    }
}

// SR_CONVT_TAPS=0 keeps the per-phase launches for small maps (A/B measurements); =1 forces the tap-split form for any size
int convt_taps_mode() {
    const char* e = std::getenv("SR_CONVT_TAPS");
    return !e ? 2 : (e[0] == '0' ? 0 : 1);
}

// K slices of the tap-split launch: enough workgroups for ~2 per CU (the nine taps already multiply the tiles by 9),
// >= 32 channels (a multiple of 16) per slice
void convt_taps_plan(int IH, int IW, int B, int N, int C, int& ks, int& c_per_slice) {
    if (sr_convt_taps_gemm_eligible(B, C, N, IW, 4, 16, nullptr)) {
        // flattened-pixel form (csrc/conv1x1_gemm.hip): 128-point tiles of the (sample, grid point) index; ~2 workgroups
        // per CU (scripts/taps_wgs_sweep.sh: 16^2 at batch 4 0.148 / 0.104 / 0.126 / 0.127 ms for 1 / 2 / 3 / 4 — more slices
        // cover a chunk's load latency with other workgroups' MFMAs but multiply the partial sums the reduction reads)
        const int64_t tiles = sr_ceil_div((int64_t)B * (IH + 1) * (IW + 1), 128) * (N / 128) * 9;
        static const int per_cu = [] {
            const char* e = std::getenv("SR_TAPS_WGS");       // workgroups per CU the K slices aim at (measurements)
            const int v = e ? std::atoi(e) : 0;
            return v >= 1 && v <= 8 ? v : 2;
        }();
        int want = (int)sr_ceil_div((int64_t)per_cu * SR_NUM_CU, tiles);
        if (want > 8) want = 8;
        if (want < 1) want = 1;
        int per = (C + want - 1) / want;
        per = (per + 15) / 16 * 16;
        if (per < 64) per = 64;
        if (per > C) per = C;
        c_per_slice = per;
        ks = (C + per - 1) / per;
        return;
    }
    int pw, ph, pb;
    patch_shape(IW + 1, pw, ph, pb);
    const int64_t blocks = (int64_t)((IW + pw) / pw) * ((IH + ph) / ph) * ((B + pb - 1) / pb) * ((N + BN - 1) / BN) * 9;
    ks = 1;
    c_per_slice = (C + 15) / 16 * 16;
    if (blocks >= 2 * SR_NUM_CU || C < 64) return;
    int want = (int)((2 * SR_NUM_CU + blocks - 1) / blocks);
    if (want > 8) want = 8;
    int per = (C + want - 1) / want;
    per = (per + 15) / 16 * 16;
    if (per < 32) per = 32;
    c_per_slice = per;
    ks = (C + per - 1) / per;
}

int64_t convt_taps_floats(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW) {
    int ks, per;
    convt_taps_plan((int)IH, (int)IW, (int)B, (int)N, (int)C, ks, per);
    return (int64_t)ks * 9 * B * N * (IH + 1) * (IW + 1);
}

// Small problems only: measured inside captured graphs (scripts/bench_convt_small.py, profiles/r05_notes.md) the
// tap-split form wins up to ~10 GFLOP per call (4^2 .. 16^2 maps at batch 4, 32^2 .. 128^2 at batch 1: 0.042 / 0.069 /
// 0.128 / 0.135 / 0.168 / 0.172 ms against 0.071 / 0.095 / 0.218 / 0.226 / 0.258 / 0.231; 32^2 at batch 2 and 16^2 at
// batch 8, 9.7 GFLOP: 0.241 against 0.259 / 0.270) and loses above (its 1x1 workgroups top out at 45-60 TFLOP/s; 32^2
// at batch 4: 0.429 against 0.352 per-phase).
bool convt_taps_wanted(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW) {
    const int mode = convt_taps_mode();
    return mode == 1 || (mode == 2 && 18.0 * (double)B * (double)C * (double)N * (double)IH * (double)IW < 1.05e10);
}

// 16-byte halo DMAs for a tap-split launch: aligned rows and an aligned tile origin (one sample per patch is checked
// by the caller's choice of the patch shape); SR_CONVT_TAPS_V4=0 keeps the 4-byte form (A/B, tests)
bool taps_v4(const ConvParams& p) {
    const char* e = std::getenv("SR_CONVT_TAPS_V4");
    return !(e && e[0] == '0') && p.IW % 4 == 0 && p.gx_base % 4 == 0 && (reinterpret_cast<uintptr_t>(p.in) & 15) == 0;
}

int launch_convt_taps(ConvParams p, hipStream_t st) {
    p.GH = p.IH + 1; p.GW = p.IW + 1;                // common grid of the four phases
    p.gy_base = p.gx_base = 0;
    p.osy = p.osx = 2; p.ooy = p.oox = 0;
    p.dy0 = p.dx0 = 0;
    p.tap_first = 0; p.tap_step = 1; p.tap_count = 9;
    convt_taps_plan(p.IH, p.IW, p.B, p.N, p.C, p.ks, p.c_per_slice);
    int pw, ph, pb;
    patch_shape(p.GW, pw, ph, pb);
    p.tiles_x = (p.GW + pw - 1) / pw;
    p.tiles_y = (p.GH + ph - 1) / ph;
    p.tiles_b = (p.B + pb - 1) / pb;
    p.tiles_n = (p.N + BN - 1) / BN;
    const int64_t blocks = (int64_t)p.tiles_x * p.tiles_y * p.tiles_b * p.tiles_n * p.ks * 9;
    if (blocks > 0x7FFFFFFFLL) return SR_ERANGE;
    const dim3 grid((unsigned)blocks);
    const bool v4 = taps_v4(p);
    int rc;
    if (sr_convt_taps_gemm_eligible(p.B, p.C, p.N, p.IW, p.ldw, p.c_per_slice, p.wt))
        // pixels = the flattened (sample, grid point) index instead of 32 x 4 patches of a (2^k + 1)-wide grid
        rc = sr_convt_taps_gemm_launch(p.partial, p.in, p.wt, p.ldw, p.iscale, p.B, p.C, p.N, p.IH, p.IW, p.ks, p.c_per_slice, st);
    else if (pw == 32) rc = v4 ? launch_one<1, 1, 1, 32, 4, 1, true, true>(p, grid, st)
                          : launch_one<1, 1, 1, 32, 4, 1, false, true>(p, grid, st);
    else if (pw == 16) rc = v4 ? launch_one<1, 1, 1, 16, 8, 1, true, true>(p, grid, st)
                               : launch_one<1, 1, 1, 16, 8, 1, false, true>(p, grid, st);
    else if (pw == 8) rc = launch_one<1, 1, 1, 8, 8, 2, false, true>(p, grid, st);
    else rc = launch_one<1, 1, 1, 4, 4, 8, false, true>(p, grid, st);
    if (rc != SR_OK) return rc;
    const int64_t total = (int64_t)p.B * p.N * p.OH * p.OW;
    hipLaunchKernelGGL(k_convt_tap_reduce, dim3(sr_stream_grid(total, 256)), dim3(256), 0, st, p);
    return sr_launch_status();
}

// Border strips behind the fused kernel (output row 2*IH, column 2*IW) through the same tap machinery: only the taps
// ky = 2 reach the last row, only kx = 2 the last column — two TAP9 launches of three taps each (32-wide patches along
// the row, 4 x 4 x 8-sample patches down the column) and one reduction, instead of four thin per-phase launches with a
// split-K reduction behind each.  partial = [row: 3 x B x N x (IW + 1)] [column: 3 x B x N x IH].
__global__ __launch_bounds__(256) void k_convt_strip_reduce(const ConvParams p, const float* __restrict__ prow,
                                                            const float* __restrict__ pcol, int ks_row, int ks_col) {
    const int rw = p.IW + 1, ch = p.IH, OW = p.OW, OH = p.OH;
    const int per_row = OW + OH - 1;                                  // strip outputs per (sample, channel)
    const int64_t planes = (int64_t)p.B * p.N, total = planes * per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / per_row;
        const int q = (int)(i - row * per_row);
        float acc;
        int64_t o;
        acc = 0.0f;
        if (q < OW) {                                                 // output (2 IH, X = q): taps (2, 0) + (2, 2) | (2, 1)
            const int g = q >> 1;
            const float* t = prow + row * rw + g;
            for (int s = 0; s < ks_row; ++s, t += 3 * planes * rw)    // K slices in order
                acc += (q & 1) ? t[planes * rw] : t[0] + t[2 * planes * rw];
            o = (int64_t)(OH - 1) * OW + q;
        } else {                                                      // output (Y, 2 IW): taps (0, 2) + (2, 2) | (1, 2)
            const int Y = q - OW, g = Y >> 1;
            const float* t = pcol + row * ch + g;
            for (int s = 0; s < ks_col; ++s, t += 3 * planes * ch)
                acc += (Y & 1) ? t[planes * ch] : t[0] + t[2 * planes * ch];
            o = (int64_t)Y * OW + (OW - 1);
        }
        if (p.oscale) acc *= p.oscale[row];
        if (p.obias) acc += p.obias[row % p.N];
        p.out[row * (int64_t)OH * OW + o] = acc;
    }
}

// (up to STRIP_KS K slices per strip)
constexpr int STRIP_KS = 4;
int64_t convt_strip_floats(int64_t B, int64_t N, int64_t IH, int64_t IW) { return STRIP_KS * 3 * B * N * (IW + 1 + IH); }

template <int PW, int PH, int PB>
int launch_strip_part(ConvParams& p, hipStream_t st) {
    p.tiles_x = (p.GW - p.gx_base + PW - 1) / PW;
    p.tiles_y = (p.GH - p.gy_base + PH - 1) / PH;
    p.tiles_b = (p.B + PB - 1) / PB;
    p.tiles_n = (p.N + BN - 1) / BN;
    int64_t blocks = (int64_t)p.tiles_x * p.tiles_y * p.tiles_b * p.tiles_n * p.tap_count;
    // a strip is a few hundred workgroups walking the whole channel loop (32 - 64 chunks of ~1.2 us each): K slices
    // shorten the chain until ~2 workgroups per CU exist (>= 64 channels, a multiple of 16, per slice)
    p.ks = 1; p.c_per_slice = (p.C + 15) / 16 * 16;
    for (int ks = STRIP_KS; ks >= 2; ks /= 2)
        if (p.C % (16 * ks) == 0 && p.C / ks >= 64 && blocks * ks <= 4 * SR_NUM_CU) {
            p.ks = ks;
            p.c_per_slice = p.C / ks;
            break;
        }
    blocks *= p.ks;
    if (blocks > 0x7FFFFFFFLL) return SR_ERANGE;
    if (PB == 1 && taps_v4(p)) return launch_one<1, 1, 1, PW, PH, PB, PB == 1, true>(p, dim3((unsigned)blocks), st);
    return launch_one<1, 1, 1, PW, PH, PB, false, true>(p, dim3((unsigned)blocks), st);
}

int launch_convt_strips(ConvParams p, hipStream_t st) {
    float* prow = p.partial;
    float* pcol = p.partial + STRIP_KS * 3 * (int64_t)p.B * p.N * (p.IW + 1);
    p.osy = p.osx = 2; p.ooy = p.oox = 0;
    p.dy0 = p.dx0 = 0;
    p.ks = 1; p.c_per_slice = (p.C + 15) / 16 * 16;
    // row 2*IH: grid row IH, all IW + 1 grid columns, taps (2, 0) (2, 1) (2, 2)
    p.gy_base = p.IH; p.GH = p.IH + 1; p.gx_base = 0; p.GW = p.IW + 1;
    p.tap_first = 6; p.tap_step = 1; p.tap_count = 3;
    p.partial = prow;
    int rc = launch_strip_part<32, 4, 1>(p, st);
    if (rc != SR_OK) return rc;
    const int ks_row = p.ks;
    // column 2*IW: grid column IW, grid rows 0 .. IH - 1 (the corner is the row's), taps (0, 2) (1, 2) (2, 2)
    p.gy_base = 0; p.GH = p.IH; p.gx_base = p.IW; p.GW = p.IW + 1;
    p.tap_first = 2; p.tap_step = 3; p.tap_count = 3;
    p.partial = pcol;
    rc = launch_strip_part<4, 4, 8>(p, st);
    if (rc != SR_OK) return rc;
    const int ks_col = p.ks;
    const int64_t total = (int64_t)p.B * p.N * (p.OW + p.OH - 1);
    hipLaunchKernelGGL(k_convt_strip_reduce, dim3(sr_stream_grid(total, 256)), dim3(256), 0, st, p, prow, pcol, ks_row,
                       ks_col);
    return sr_launch_status();
}

// Winograd F(2x2,3x3) for the stride-1 3x3 convolution (csrc/conv_wino.hip); SR_WINOGRAD=0 keeps the
// direct implicit GEMM (A/B measurements, exact-fma-chain numerics).
bool wino_enabled() {
    const char* e = std::getenv("SR_WINOGRAD");      // read per call: tests flip it at run time
    return !(e && e[0] == '0');
}

// SR_CONV_ROT=0 keeps the 4-byte halo DMA of the stride-2 kernels (A/B measurements)
bool rot_enabled() {
    const char* e = std::getenv("SR_CONV_ROT");
    return !(e && e[0] == '0');
}

template <int IS, int TY, int TX>
int launch_by_patch(ConvParams& p, hipStream_t st) {
    int pw, ph, pb;
    const int ext_w = p.GW - p.gx_base, ext_h = p.GH - p.gy_base;
    if (ext_w <= 0 || ext_h <= 0) return SR_OK;
    // split-K is decided per launch (thin border strips of the transposed conv need it even when
    // the interior does not); each launch reduces its own compact slab
    choose_split(ext_h, ext_w, p.B, p.N, p.C, p.ks, p.c_per_slice);
    if (!p.partial || (int64_t)p.ks * p.B * p.N * ext_h * ext_w > p.partial_floats) {
        p.ks = 1;
        p.c_per_slice = (p.C + 15) / 16 * 16;
    }
    patch_shape(ext_w, pw, ph, pb);
    p.tiles_x = (ext_w + pw - 1) / pw;
    p.tiles_y = (ext_h + ph - 1) / ph;
    p.tiles_b = (p.B + pb - 1) / pb;
    p.tiles_n = (p.N + BN - 1) / BN;
    const int64_t blocks = (int64_t)p.tiles_x * p.tiles_y * p.tiles_b * p.tiles_n * p.ks;
    if (blocks <= 0) return SR_OK;
    if (blocks > 0x7FFFFFFFLL) return SR_ERANGE;
    const dim3 grid((unsigned)blocks);
    // 16-byte halo DMAs: stride 1, one sample per tile, aligned rows, window origin dx0 = -(TX > 1)
    const bool v4 = IS == 1 && pb == 1 && p.IW % 4 == 0 && (reinterpret_cast<uintptr_t>(p.in) & 15) == 0 &&
                    p.gx_base % 4 == 0 && p.dx0 == (TX > 1 ? -1 : 0);
    // rotating-lead 16-byte DMAs for stride 2 (see Geo): (2^k+1)-sized maps, window inside the image, whole tiles
    const bool rot = IS == 2 && pb == 1 && (p.IW & 3) == 1 && ((p.IH * p.IW) & 3) == 1 && p.C % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(p.in) & 15) == 0 && p.dx0 >= 0 && p.dy0 >= 0 && p.gx_base == 0 &&
                     p.gy_base == 0 && (p.GW - 1) * 2 + p.dx0 + TX <= p.IW && (p.GH - 1) * 2 + p.dy0 + TY <= p.IH &&
                     ext_w % pw == 0 && ext_h % ph == 0 && rot_enabled();
    int rc;
    if (IS == 1 && v4) {
        if (pw == 32) rc = launch_one<1, TY, TX, 32, 4, 1, true>(p, grid, st);
        else rc = launch_one<1, TY, TX, 16, 8, 1, true>(p, grid, st);
    } else if (IS == 2 && rot && TY == 3 && TX == 3) {
        if (pw == 32) rc = launch_one<2, 3, 3, 32, 4, 1, true>(p, grid, st);
        else rc = launch_one<2, 3, 3, 16, 8, 1, true>(p, grid, st);
    } else if (pw == 32) rc = launch_one<IS, TY, TX, 32, 4, 1, false>(p, grid, st);
    else if (pw == 16) rc = launch_one<IS, TY, TX, 16, 8, 1, false>(p, grid, st);
    else if (pw == 8) rc = launch_one<IS, TY, TX, 8, 8, 2, false>(p, grid, st);
    else rc = launch_one<IS, TY, TX, 4, 4, 8, false>(p, grid, st);
    if (rc != SR_OK || p.ks <= 1) return rc;
    const int64_t total = (int64_t)p.B * p.N * ext_h * ext_w;
    hipLaunchKernelGGL(k_conv_reduce, dim3(sr_stream_grid(total, 256)), dim3(256), 0, st, p, ext_h, ext_w);
    return sr_launch_status();
}

}  // namespace

extern "C" int64_t sr_conv2d_scratch_floats(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW,
                                            int64_t OH, int64_t OW, int ksize, int stride, int pad,
                                            int transposed) {
    (void)ksize; (void)stride; (void)pad;
    if (B <= 0 || C <= 0 || N <= 0 || OH <= 0 || OW <= 0) return 0;
    // regions a launch may split: the whole output grid (small maps) or, for the transposed conv,
    // its border strips.  Upper bound: ks * B * N * region for the largest split region.
    int ks, per;
    int64_t need = 0;
    auto consider = [&](int64_t gh, int64_t gw) {
        if (gh <= 0 || gw <= 0) return;
        choose_split((int)gh, (int)gw, (int)B, (int)N, (int)C, ks, per);
        if (ks > 1) need = need > ks * B * N * gh * gw ? need : ks * B * N * gh * gw;
    };
    if (!transposed) {
        consider(OH, OW);
        if (ksize == 3 && stride == 2 && pad == 0 && sr_wgrad_bf16x3_enabled('c') &&
            sr_conv_s2_bf16x3_eligible(B, C, N, IH, IW, OH, OW)) {
            const int64_t w = sr_conv_s2_bf16x3_scratch_floats(C, N);
            need = need > w ? need : w;
        }
        if (ksize == 3 && stride == 1 && pad == 1 && wino_enabled() &&
            sr_wino_eligible(B, C, N, IH, IW, nullptr, nullptr)) {
            const int64_t w = sr_wino_scratch_floats(C, N) + sr_wino_partial_floats(B, C, N, IH, IW);
            need = need > w ? need : w;
        }
    } else {
        consider(IH + 1, IW + 1);
        consider(IH, IW);
        consider(1, IW + 1);
        consider(IH, 1);
        if (ksize == 3 && stride == 2 && pad == 0 && convt_taps_wanted(B, C, N, IH, IW)) {
            const int64_t w = convt_taps_floats(B, C, N, IH, IW);
            need = need > w ? need : w;
        }
        if (ksize == 3 && stride == 2 && pad == 0) {
            const int64_t w = convt_strip_floats(B, N, IH, IW), f = convt_fused_split_floats(B, C, N, IH, IW);
            need = need > w ? need : w;
            need = need > f ? need : f;
        }
        if (ksize == 3 && stride == 2 && pad == 0 && sr_wgrad_bf16x3_enabled('t') && sr_convt_bf16x3_eligible(B, C, N, IH, IW)) {
            const int64_t w = sr_convt_bf16x3_scratch_floats(C, N);
            need = need > w ? need : w;
        }
    }
    return need;
}

// 1 when a stride-1 3x3 convolution call with these sizes AND these buffers runs the Winograd kernel (and therefore
// writes / reads the Winograd-domain weights at the head of its scratch), 0 when it takes the direct kernel.
extern "C" int sr_conv2d_uses_winograd(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW, const float* in,
                                       const float* out) {
    return (wino_enabled() && sr_wino_eligible(B, C, N, IH, IW, in, out)) ? 1 : 0;
}

extern "C" int sr_conv2d_mfma_ex(float* out, const float* in, const float* wt, const float* iscale,
                              const float* oscale, const float* obias, int64_t B, int64_t C,
                              int64_t N, int64_t wt_ld, int64_t IH, int64_t IW, int64_t OH, int64_t OW,
                              int ksize, int stride, int pad, int transposed, int flags, float* scratch,
                              sr_stream_t stream) {
    if (B < 0 || C <= 0 || N <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return SR_EINVAL;
    if (wt_ld < N || wt_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(wt) & 15)) return SR_EINVAL;
    if (B == 0) return SR_OK;
    if (!out || !in || !wt) return SR_EINVAL;
    // 32-bit element offsets inside the kernel
    if (B * C * IH * IW >= (1LL << 31) || B * N * OH * OW >= (1LL << 40) ||
        (int64_t)ksize * ksize * C * wt_ld >= (1LL << 31))
        return SR_ERANGE;
    hipStream_t st = sr_stream(stream);
    ConvParams p;
    p.in = in; p.wt = wt; p.iscale = iscale; p.oscale = oscale; p.obias = obias; p.out = out;
    p.B = (int)B; p.C = (int)C; p.N = (int)N; p.ldw = (int)wt_ld;
    p.IH = (int)IH; p.IW = (int)IW; p.OH = (int)OH; p.OW = (int)OW;
    p.partial = scratch;
    p.partial_floats = scratch ? sr_conv2d_scratch_floats(B, C, N, IH, IW, OH, OW, ksize, stride, pad, transposed) : 0;
    for (int i = 0; i < 9; ++i) p.wmap[i] = 0;
    p.ks = 1; p.c_per_slice = (p.C + 15) / 16 * 16;
    p.tap_first = 0; p.tap_step = 1; p.tap_count = 9;
    if (!transposed) {
        if (OH != (IH + 2 * pad - ksize) / stride + 1 || OW != (IW + 2 * pad - ksize) / stride + 1)
            return SR_EINVAL;
        p.GH = p.OH; p.GW = p.OW;
        p.gy_base = p.gx_base = 0;
        p.osy = p.osx = 1; p.ooy = p.oox = 0;
        p.dy0 = p.dx0 = -pad;
        for (int i = 0; i < ksize * ksize; ++i) p.wmap[i] = i;
        int rc;
        if (ksize == 3 && stride == 1 && pad == 1 && scratch && wino_enabled() &&
            sr_wino_eligible(B, C, N, IH, IW, in, out))
            return sr_wino_conv3x3(out, in, wt, wt_ld, iscale, oscale, obias, B, C, N, IH, IW, scratch, st, nullptr,
                                   (flags & SR_CONV_U_READY) != 0);
        if (ksize == 3 && stride == 2 && pad == 0 && scratch && sr_wgrad_bf16x3_enabled('c') &&
            sr_conv_s2_bf16x3_eligible(B, C, N, IH, IW, OH, OW))
            // opt-in spike (SR_CONV_SPLIT_BF16=1): split-bf16 matrix path for the down-sampling convolution and the
            // data gradient of the up-sampling one
            return sr_conv_s2_bf16x3_launch(out, in, wt, wt_ld, iscale, oscale, obias, B, C, N, IH, IW, OH, OW, scratch, st);
        if (ksize == 3 && stride == 1) rc = launch_by_patch<1, 3, 3>(p, st);
        else if (ksize == 3 && stride == 2) rc = launch_by_patch<2, 3, 3>(p, st);
        else if (ksize == 1 && stride == 1) {
            // no window: a plain GEMM with both operands K-major (csrc/conv1x1_gemm.hip) where its tiles fill the chip
            if (pad == 0 && sr_conv1x1_gemm_eligible(B, C, N, wt_ld, IH * IW, in, wt, out))
                return sr_conv1x1_gemm_launch(out, in, wt, wt_ld, iscale, oscale, obias, B, C, N, IH * IW, st);
            rc = launch_by_patch<1, 1, 1>(p, st);
        }
        else if (ksize == 1 && stride == 2) rc = launch_by_patch<2, 1, 1>(p, st);
        else return SR_EINVAL;
        return rc;
    }
    // transposed 3x3 stride 2, no padding: out[2y + ky, 2x + kx] += in[y, x] * W[ky][kx].
    // Output phase (py, px) of grid point (j, i) = output (2j + py, 2i + px); window position ty
    // reads input row j + ty - (TY - 1) and pairs with ky = py + 2 * (TY - 1 - ty).
    if (ksize != 3 || stride != 2 || pad != 0 || OH != 2 * IH + 1 || OW != 2 * IW + 1) return SR_EINVAL;
    // The phase grids have 2^k + 1 points per side: the 2^k x 2^k interior tiles the 32-wide
    // patches exactly, the last grid row / column (output row 2*IH, column 2*IW: even phases only)
    // runs as thin strip launches instead of padding every tile row by up to 50 %.
    // interior of the map: all four phases in one workgroup (k_convt_fused); SR_CONVT_FUSED=0 keeps the
    // per-phase launches
    // small problems: nine shifted 1x1 convolutions in one launch + one reduction (k_conv_mfma<..., TAP9>)
    if (scratch && convt_taps_wanted(B, C, N, IH, IW) && convt_taps_floats(B, C, N, IH, IW) <= p.partial_floats &&
        !(sr_wgrad_bf16x3_enabled('t') && sr_convt_bf16x3_eligible(B, C, N, IH, IW)))
        return launch_convt_taps(p, st);
    bool fused_ok = false;
    {
        const char* e = std::getenv("SR_CONVT_FUSED");
        // the fused kernel has no split-K: when its workgroups cover less than ~3/4 of the CUs AND the channel loop is
        // long (C > 256: 64 chunks, ~0.38 ms whatever the batch), each one walks the whole loop alone and the per-phase
        // launches, which split K, are faster — in-graph, scripts/bench_convt_small.py: 32^2 512->512 at batch 1 / 2 / 4
        // 0.23 / 0.26 / 0.35 ms against 0.38 / 0.39 / 0.40 fused (batch 8: 0.56 against 0.43, fused stays);
        // 64^2 512->256 at batch 1 / 2 0.26 / 0.33 against 0.39; 128^2 256->128 (32 chunks) is fused from batch 1 on
        const int64_t fused_blocks = (int64_t)(p.IW / TFused::PW) * (p.IH / TFused::PH) * ((p.N + BN - 1) / BN) * p.B;
        bool fused_pays = fused_blocks >= 192 || p.C <= 256;
        if (!fused_pays && scratch && !(e && e[0] == '0')) {
            // few tiles, long channel loop: K slices inside the fused kernel (SR_CONVT_FUSED_KS=0: the per-phase launches)
            const char* k = std::getenv("SR_CONVT_FUSED_KS");
            const int ks = (k && k[0] == '0') ? 1 : convt_fused_slices(fused_blocks, p.C);
            if (ks > 1 && convt_fused_eligible(p) && (int64_t)ks * B * N * OH * OW <= p.partial_floats) {
                p.ks = ks;
                p.c_per_slice = p.C / ks;
                fused_pays = true;
            }
        }
        if (scratch && sr_wgrad_bf16x3_enabled('t') && sr_convt_bf16x3_eligible(B, C, N, IH, IW)) {
            // opt-in spike (SR_CONV_SPLIT_BF16=1): the interior of the map on the bf16 matrix cores (three-way operand
            // split); the border strips below stay on the exact-fp32 kernels
            const int rc = sr_convt_bf16x3_launch(out, in, wt, wt_ld, iscale, oscale, obias, B, C, N, IH, IW, scratch, st);
            if (rc != SR_OK) return rc;
            fused_ok = true;
        } else if (!(e && e[0] == '0') && p.IW >= 16 && convt_fused_eligible(p) && (fused_pays || (e && e[0] == '1'))) {
            for (int i = 0; i < 9; ++i) p.wmap[i] = i;
            const int rc = launch_convt_fused(p, st);
            p.ks = 1; p.c_per_slice = (p.C + 15) / 16 * 16;          // (the strips below plan their own)
            if (rc != SR_OK) return rc;
            fused_ok = true;
        }
    }
    if (fused_ok && scratch && convt_strip_floats(B, N, IH, IW) <= p.partial_floats) {
        // the interior is done: output row 2*IH and column 2*IW as two three-tap launches + one reduction
        // (SR_CONVT_STRIPS=0: the four thin per-phase launches and their split-K reductions)
        const char* e = std::getenv("SR_CONVT_STRIPS");
        if (!(e && e[0] == '0')) return launch_convt_strips(p, st);
    }
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const int TYp = py == 0 ? 2 : 1, TXp = px == 0 ? 2 : 1;
            p.osy = p.osx = 2; p.ooy = py; p.oox = px;
            p.dy0 = -(TYp - 1); p.dx0 = -(TXp - 1);
            for (int ty = 0; ty < TYp; ++ty)
                for (int tx = 0; tx < TXp; ++tx) {
                    const int ky = py + 2 * (TYp - 1 - ty), kx = px + 2 * (TXp - 1 - tx);
                    p.wmap[ty * TXp + tx] = ky * 3 + kx;
                }
            const int gh_full = py == 0 ? p.IH + 1 : p.IH, gw_full = px == 0 ? p.IW + 1 : p.IW;
            // sub-grids: {rows, cols} ranges as [y0, y1) x [x0, x1)
            int regions[3][4] = {{0, p.IH, 0, p.IW},                       // interior
                                 {p.IH, gh_full, 0, gw_full},              // last row (py == 0)
                                 {0, p.IH, p.IW, gw_full}};                // last column (px == 0)
            // (tiny maps are launch bound: one launch per phase there)
            const bool split_border = p.IW >= 16;
            if (!split_border) {
                regions[0][1] = gh_full;
                regions[0][3] = gw_full;
            }
            const bool fused = split_border && fused_ok;
            for (int rg = fused ? 1 : 0; rg < (split_border ? 3 : 1); ++rg) {
                p.gy_base = regions[rg][0]; p.GH = regions[rg][1];
                p.gx_base = regions[rg][2]; p.GW = regions[rg][3];
                if (p.GH <= p.gy_base || p.GW <= p.gx_base) continue;
                int rc;
                if (py == 0 && px == 0) rc = launch_by_patch<1, 2, 2>(p, st);
                else if (py == 0) rc = launch_by_patch<1, 2, 1>(p, st);
                else if (px == 0) rc = launch_by_patch<1, 1, 2>(p, st);
                else rc = launch_by_patch<1, 1, 1>(p, st);
                if (rc != SR_OK) return rc;
            }
        }
    return SR_OK;
}

extern "C" int sr_conv2d_mfma(float* out, const float* in, const float* wt, const float* iscale,
                              const float* oscale, const float* obias, int64_t B, int64_t C,
                              int64_t N, int64_t wt_ld, int64_t IH, int64_t IW, int64_t OH, int64_t OW,
                              int ksize, int stride, int pad, int transposed, float* scratch,
                              sr_stream_t stream) {
    return sr_conv2d_mfma_ex(out, in, wt, iscale, oscale, obias, B, C, N, wt_ld, IH, IW, OH, OW, ksize, stride, pad,
                             transposed, 0, scratch, stream);
}

// Modulated 3x3 stride-1 convolution with the StyledConv tail fused into the store (Winograd kernel only).
// SR_EINVAL when the shape is not taken by the Winograd path: the caller then runs sr_conv2d_mfma followed
// by sr_noise_bias_act.
extern "C" int sr_conv2d_nba_ex(float* out, const float* in, const float* wt, const float* iscale, const float* oscale,
                                const float* noise, const float* noise_w, const float* abias, float alpha, float gain,
                                int64_t B, int64_t C, int64_t N, int64_t wt_ld, int64_t H, int64_t W,
                                int64_t noise_bstride, int flags, float* scratch, sr_stream_t stream) {
    if (B < 0 || C <= 0 || N <= 0 || H <= 0 || W <= 0) return SR_EINVAL;
    if (wt_ld < N || wt_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(wt) & 15)) return SR_EINVAL;
    if (B == 0) return SR_OK;
    if (!out || !in || !wt || !scratch || (noise && !noise_w)) return SR_EINVAL;
    if (!wino_enabled() || !sr_wino_eligible(B, C, N, H, W, in, out)) return SR_EINVAL;
    const WinoNba nba{noise, noise_w, abias, noise_bstride, alpha, gain};
    return sr_wino_conv3x3(out, in, wt, wt_ld, iscale, oscale, nullptr, B, C, N, H, W, scratch, sr_stream(stream), &nba,
                           (flags & SR_CONV_U_READY) != 0);
}

extern "C" int sr_conv2d_nba(float* out, const float* in, const float* wt, const float* iscale, const float* oscale,
                             const float* noise, const float* noise_w, const float* abias, float alpha, float gain,
                             int64_t B, int64_t C, int64_t N, int64_t wt_ld, int64_t H, int64_t W,
                             int64_t noise_bstride, float* scratch, sr_stream_t stream) {
    return sr_conv2d_nba_ex(out, in, wt, iscale, oscale, noise, noise_w, abias, alpha, gain, B, C, N, wt_ld, H, W,
                            noise_bstride, 0, scratch, stream);
}
