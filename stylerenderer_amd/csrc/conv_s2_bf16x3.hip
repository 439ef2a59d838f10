// SPIKE (opt-in, SR_CONV_SPLIT_BF16=1): the stride-2 3x3 convolution (pad 0: the down-sampling convolutions and the
// data gradient of the up-sampling transposed convolutions) on the BF16 matrix cores at fp32-level accuracy — see
// conv_wgrad_bf16x3.hip for the three-way operand split and the six-product scheme.
//
//   out[b,n,oy,ox] = oscale[b,n] * sum_{tap,c} W[tap][c][n] * (iscale[b,c] * in[b,c,2 oy + ky,2 ox + kx]) (+ obias[n])
//
// GEMM roles: A = weights (rows n), B = input window (columns = 32 output pixels of one row), K = (tap, channel).
// Unlike the weight gradient, only ONE operand has to be split inside the loop: the weights are split once per call by
// k_split_w_s2 into bf16 pieces laid out exactly as the workgroups stage them, and every split input fragment feeds all
// four 32-channel blocks of its wave (24 MFMAs per ~37 VALU operations).
// K steps pair two taps: lanes 0-31 carry channels 0..7 of tap 2m, lanes 32-63 of tap 2m + 1 (m = 0..4; the tenth
// "tap" reads a zero weight block).  The bf16 MFMA wants 8 consecutive k per lane = 8 channels at one window
// position: eight ds_read_b32 at channel stride from the fp32 patch (same LDS bytes as an fp32 operand fetch), window
// columns stored de-interleaved (even | odd) so that the 32 pixels of a fetch hit 32 banks for every kx.
// Tile: workgroup = 128 output channels x (4 rows x 32 columns) pixels of one sample, wave = one row x 128 channels
// (4 accumulator tiles); K chunk = 8 input channels: weights 9 taps x 3 pieces x 128 x 8 bf16 = 55 KB by 16-byte
// LDS-DMA, input patch 8 x 9 x 65 floats through registers (de-interleave, iscale).  ONE 75 KB buffer per workgroup and
// TWO workgroups per CU: a workgroup's staging (DMA + patch, latency exposed) runs under the other workgroup's MFMA
// block.  (v1 double-buffered 156 KB in one workgroup per CU: with a single wave per SIMD its ~650 non-MFMA
// instructions and 14 DMA issues per chunk ran beside, not under, its 120 MFMAs — 0.41 of the matrix pipe, and the
// ablation without any MFMA still took 65 % of the time.)
#include "common.h"
#include "conv_s2_bf16x3.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

constexpr int NT = 128, KC = 8, THREADS = 256;
#ifndef SR_FLIP_LOG2
#define SR_FLIP_LOG2 1            // chunks per sign block of the floor-bias cancellation = 1 << SR_FLIP_LOG2 (A/B: r06 notes)
#endif
constexpr int PX = 68, ODD = 34;                 // patch row: 33 even columns | 32 odd columns
constexpr int XCH = 9 * PX;                      // floats per patch channel
constexpr int X_FLOATS = KC * XCH;               // 4 896
constexpr int W_TAP = 3 * 4 * 32 * 4;            // dwords per tap: 3 pieces x 4 channel blocks x 32 lanes x 16 B
constexpr int W_DWORDS = 9 * W_TAP;              // 13 824 dwords = 55 296 B per chunk
constexpr int BUF = W_DWORDS + X_FLOATS;         // 18 720 dwords
constexpr int LDS_BYTES = BUF * 4;               // 74 880: ONE buffer, TWO workgroups per CU
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
constexpr int W_INSTR = W_DWORDS / 256;          // 54 DMA instructions of 1 KB per chunk
constexpr int X_ITEMS = KC * 9 / 4;              // (channel, row) pairs per wave: 18

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float bf16_hi(unsigned pk) { return __builtin_bit_cast(float, pk & 0xFFFF0000u); }
__device__ __forceinline__ void split2(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = pack_bf16(x0, x1);
    const float r0 = x0 - bf16_lo(p1), r1 = x1 - bf16_hi(p1);
    p2 = pack_bf16(r0, r1);
    const float q0 = r0 - bf16_lo(p2), q1 = r1 - bf16_hi(p2);
    p3 = pack_bf16(q0, q1);
}
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& h1, u32x4& h2, u32x4& h3) {
    unsigned a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
    split2(x[0], x[1], a0, b0, c0);
    split2(x[2], x[3], a1, b1, c1);
    split2(x[4], x[5], a2, b2, c2);
    split2(x[6], x[7], a3, b3, c3);
    h1 = u32x4{a0, a1, a2, a3};
    h2 = u32x4{b0, b1, b2, b3};
    h3 = u32x4{c0, c1, c2, c3};
}
__device__ __forceinline__ f32x16 mma(const u32x4 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- weights: fp32 [9][C][ldw] -> bf16 pieces in staging order ------------------------------------------------------
//   Wb[(g * tiles_n + tn)][tap][piece][nb][l31][8 c]   (g = channel chunk of 8, tn = 128-channel tile, nb = 32-block)
__global__ __launch_bounds__(256) void k_split_w_s2(unsigned* __restrict__ wb, const float* __restrict__ wt, int C, int N,
                                                    int ldw, int tiles_n) {
    const int64_t total = (int64_t)(C / KC) * tiles_n * 9 * 4 * 32;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int l31 = (int)(i & 31), nb = (int)((i >> 5) & 3);
        int64_t r = i >> 7;
        const int tap = (int)(r % 9);
        r /= 9;
        const int tn = (int)(r % tiles_n), g = (int)(r / tiles_n);
        const int n = tn * NT + nb * 32 + l31;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = n < N ? wt[((int64_t)tap * C + g * KC + j) * ldw + n] : 0.0f;
        u32x4 h1, h2, h3;
        split8(x, h1, h2, h3);
        unsigned* dst = wb + ((int64_t)(g * tiles_n + tn) * 9 + tap) * W_TAP + (nb * 32 + l31) * 4;
        *reinterpret_cast<u32x4*>(dst) = h1;
        *reinterpret_cast<u32x4*>(dst + 4 * 32 * 4) = h2;
        *reinterpret_cast<u32x4*>(dst + 2 * 4 * 32 * 4) = h3;
    }
}

struct PS2 {
    const float* in;
    const unsigned* wb;
    const float* iscale;
    const float* oscale;
    const float* obias;
    float* out;
    int B, C, N, IH, IW, OH, OW;
    int tiles_x, tiles_y, tiles_n;
};

// FLOOR BIAS of the bf16 matrix core and how it is cancelled (scripts/mfma_guard_probe.cpp, scripts/split_bias_probe.py).
// v_mfma_f32_32x32x16_bf16 aligns the 16 products of a row to the accumulator's exponent with 8 guard bits and CHOPS what
// lies below in two's complement — towards -infinity, whatever the signs (C = 1, p = -2^-26 ulp still moves the result
// down; C = 1, p = +2^-9 ulp is lost) — before one round-to-nearest.  The h x h products sit above the guard bits, but
// every cross-term MFMA (h m, m h, h l, l h, m m: 2^-8 .. 2^-16 of the accumulator) loses an expected 2^-9 ulp(acc): over
// the ~1 500 such MFMAs of a 512-channel convolution that is a SYSTEMATIC -1e-7 of sum|a||b| (the fp32 MFMA kernel:
// -2e-10).  Invisible in a max-error bound (2.4e-7 against 1.2e-7), it is a common-mode error over the output channels of
// a pixel, and gradient sums that cancel to 1 % of their terms (the bias gradients of GeneratorWithMap's map heads)
// amplify it to 6e-5.  There is no register room for a second accumulator set (245 of 256), so the sign of the
// ACCUMULATION alternates instead: odd chunks stage the negated input fragment and the accumulators are negated at every
// chunk boundary (64 v_xor under the patch's load latency), so consecutive chunks' floor errors enter the sum with opposite
// signs and telescope; the last chunk's sign is taken out in the epilogue.  Measured: mean signed error -1.1e-7 -> see
// profiles/r06_notes.md.
__global__ __launch_bounds__(THREADS, 2) void k_conv_s2_bf16x3(const PS2 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    int bid = blockIdx.x;
    const int tn = bid % p.tiles_n;
    bid /= p.tiles_n;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y, b = bid / p.tiles_y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int oy0 = ty * 4, ox0 = tx * 32, n0 = tn * NT;
    const int plane_in = p.IH * p.IW;
    const int nchunk = p.C / KC;

    // ---- staging.  Weights: chunk g of tile tn is 54 KB contiguous in `wb`, copied by 54 DMA instructions of 1 KB
    // (13 per wave, the last two by waves 0 and 1).  Patch: wave w stages channels 2 w, 2 w + 1 (9 rows each, 64 columns
    // per load, de-interleaved); lanes 0..17 the 65th column of (channel 2 w + lane / 9, row lane % 9).
    const unsigned* wsrc = p.wb + (int64_t)tn * W_DWORDS + lane * 4;
    const int64_t w_chunk = (int64_t)p.tiles_n * W_DWORDS;
    const float* xin = p.in + (int64_t)b * p.C * plane_in + (2 * oy0) * p.IW + 2 * ox0;
    const float* isb = p.iscale ? p.iscale + (int64_t)b * p.C : nullptr;
    const int l65 = lane < 18 ? lane : 0;
    const int ch65 = 2 * wave + l65 / 9, r65 = l65 % 9;
    const int off65 = ch65 * plane_in + r65 * p.IW + 64;
    const int lds_col = (lane & 1) ? ODD + (lane >> 1) : (lane >> 1);
    typedef const float __attribute__((address_space(4)))* cptr_t;
    float* sXw = reinterpret_cast<float*>(smem + W_DWORDS);

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    for (int g = 0; g < nchunk; ++g) {
        __syncthreads();              // every wave is done reading the buffer
        {
            const unsigned* src = wsrc + g * w_chunk;
#pragma unroll
            for (int i = 0; i < W_INSTR / 4; ++i) {
                const int j = wave + 4 * i;
                __builtin_amdgcn_global_load_lds((gptr_t)(src + j * 256), (lptr_t)(smem + j * 256), 16, 0, 0);
            }
            if (wave < W_INSTR % 4) {
                const int j = wave + 4 * (W_INSTR / 4);
                __builtin_amdgcn_global_load_lds((gptr_t)(src + j * 256), (lptr_t)(smem + j * 256), 16, 0, 0);
            }
            float sc0 = 1.0f, sc1 = 1.0f;
            if (isb) {
                const cptr_t c = (cptr_t)(isb + g * KC + 2 * wave);
                sc0 = c[0];
                sc1 = c[1];
            }
#ifndef SR_ABL_NOFLIP
            // Floor-bias cancellation (see the note above k_conv_s2_bf16x3): odd chunks accumulate the NEGATED sum — the
            // input fragment takes the sign here, the accumulators are negated at every chunk boundary below.
            if ((g >> SR_FLIP_LOG2) & 1) {
                sc0 = -sc0;
                sc1 = -sc1;
            }
#endif
            const float* xb = xin + (int64_t)g * KC * plane_in;
            const float* xw = xb + (2 * wave) * plane_in + lane;
            float stX[X_ITEMS];
#pragma unroll
            for (int k = 0; k < X_ITEMS; ++k) stX[k] = xw[(k / 9) * plane_in + (k % 9) * p.IW];
            const float stX65 = xb[off65];
#if !defined(SR_ABL_NOFLIP) && !defined(SR_FLIP_AT_MFMA)
            // acc = -acc (chunk 0: zeros): under the global-load latency of this chunk's patch
            if (SR_FLIP_LOG2 == 0 || (g & ((1 << SR_FLIP_LOG2) - 1)) == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = -acc[i][r];
            }
#endif
            float* dw = sXw + (2 * wave) * XCH + lds_col;
#pragma unroll
            for (int k = 0; k < X_ITEMS; ++k) dw[(k / 9) * XCH + (k % 9) * PX] = stX[k] * (k < 9 ? sc0 : sc1);
            if (lane < 18) sXw[ch65 * XCH + r65 * PX + 32] = stX65 * (lane < 9 ? sc0 : sc1);     // column 64 = even #32
        }
        __syncthreads();              // chunk complete (the barrier drains this wave's DMA: vmcnt(0))
#if !defined(SR_ABL_NOFLIP) && defined(SR_FLIP_AT_MFMA)
        if (SR_FLIP_LOG2 == 0 || (g & ((1 << SR_FLIP_LOG2) - 1)) == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = -acc[i][r];
        }
#endif
        const unsigned* sW = smem;
        const float* sX = sXw + (2 * wave) * PX;
        // Operand fetch is register double-buffered by hand: the LDS reads of k step m + 1 go out BEFORE the MFMA block
        // of step m.
        u32x4 a1[2][4], a2[2][4], a3[2][4];
        float xr[2][8];
        auto fetch = [&](int m, int set) {
            // this lane's tap: 2 m + half; the tenth "tap" (m = 4, half 1) reads tap 8's operands and multiplies by a
            // ZERO input fragment
            constexpr int TA[5] = {0, 2, 4, 6, 8}, TB[5] = {1, 3, 5, 7, 8};
            const int ta = TA[m], tb = TB[m];
            const int offa = (ta / 3) * PX + ((ta % 3) == 1 ? ODD : (ta % 3) >> 1);
            const int offb = (tb / 3) * PX + ((tb % 3) == 1 ? ODD : (tb % 3) >> 1);
            const float* px = sX + l31 + (half ? offb : offa);
#pragma unroll
            for (int c = 0; c < 8; ++c) xr[set][c] = px[c * XCH];
            const unsigned* pw = sW + (half ? tb : ta) * W_TAP + l31 * 4;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                a1[set][nb] = *reinterpret_cast<const u32x4*>(pw + nb * 128);
                a2[set][nb] = *reinterpret_cast<const u32x4*>(pw + 512 + nb * 128);
                a3[set][nb] = *reinterpret_cast<const u32x4*>(pw + 1024 + nb * 128);
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            const int cur = m & 1;
            if (m == 4) {
#pragma unroll
                for (int c = 0; c < 8; ++c) xr[cur][c] = half ? 0.0f : xr[cur][c];
            }
            u32x4 b1, b2, b3;
            split8(xr[cur], b1, b2, b3);
            if (m + 1 < 5) fetch(m + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);          // the reads go out BEFORE the MFMA block
            // term by term over the four channel blocks: consecutive MFMAs never share an accumulator
#define SR_TERM(A, Bv)                                                              \
    acc[0] = mma(A[cur][0], Bv, acc[0]); acc[1] = mma(A[cur][1], Bv, acc[1]);       \
    acc[2] = mma(A[cur][2], Bv, acc[2]); acc[3] = mma(A[cur][3], Bv, acc[3]);
            SR_TERM(a3, b1) SR_TERM(a1, b3) SR_TERM(a2, b2) SR_TERM(a2, b1) SR_TERM(a1, b2) SR_TERM(a1, b1)
#undef SR_TERM
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // epilogue: C/D layout column = lane & 31 (pixel), row = (r & 3) + 8 (r >> 2) + 4 half (channel of the block)
    const int oy = oy0 + wave, ox = ox0 + l31;
#ifndef SR_ABL_NOFLIP
    const float fsign = (((nchunk - 1) >> SR_FLIP_LOG2) & 1) ? -1.0f : 1.0f;          // the last chunk's sign
#else
    const float fsign = 1.0f;
#endif
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // channel of this accumulator register: wave-uniform per half -> two scalar loads and a select
            const int nu = n0 + nb * 32 + (r & 3) + 8 * (r >> 2);
            const int n = nu + 4 * half;
            float v = acc[nb][r] * fsign;
            if (p.oscale) {
                const cptr_t os = (cptr_t)(p.oscale + (int64_t)b * p.N + nu);
                v *= half ? os[4] : os[0];
            }
            if (p.obias) {
                const cptr_t ob = (cptr_t)(p.obias + nu);
                v += half ? ob[4] : ob[0];
            }
            p.out[(((int64_t)b * p.N + n) * p.OH + oy) * p.OW + ox] = v;
        }
}


// ---- stride-2 TRANSPOSED 3x3 convolution (the up-sampling layers' forward), interior of the map ----------------------
//   out[b,n,2j+ky,2i+kx] += oscale[b,n] * W[ky][kx][c][n] * (iscale[b,c] * in[b,c,j,i])
// As four output phases of the grid point (a, b) = (oy >> 1, ox >> 1):
//   (0,0): in(a,b) W00 + in(a-1,b) W20 + in(a,b-1) W02 + in(a-1,b-1) W22      (0,1): in(a,b) W01 + in(a-1,b) W21
//   (1,0): in(a,b) W10 + in(a,b-1) W12                                         (1,1): in(a,b) W11
// Only FOUR input shifts feed all nine taps, so a k step pairs (shift, tap) couples of the SAME phase on the two lane
// halves: {S00 W00 | S10 W20}, {S01 W02 | S11 W22} -> (0,0); {S00 W01 | S10 W21} -> (0,1); {S00 W10 | S01 W12} -> (1,0);
// {S00 W11 | zero} -> (1,1): five groups of six products, three operand splits per lane.  Interior only (a < IH,
// b < IW): the last output row / column stay with the fp32 strip launches of conv_mfma.hip.
// Tile: workgroup = 64 output channels x (4 x 32) grid points, wave = one grid row: 4 phases x 2 channel blocks = 8
// accumulator tiles; K chunk = 8 input channels (weights 27 KB by LDS-DMA + a 5 x 33 patch per channel), one buffer,
// several workgroups per CU.
namespace tc {
constexpr int NT = 64, KC = 16, THREADS = 256;   // KC: channels per chunk = two 8-channel weight blocks
constexpr int PXT = 36, XCH = 5 * PXT;           // patch: 5 rows (a0 - 1 .. a0 + 3) x 33 columns (b0 - 1 .. b0 + 31)
constexpr int X_FLOATS = KC * XCH;               // 2 880
constexpr int W_TAP = 3 * 2 * 32 * 4;            // dwords per tap: 3 pieces x 2 channel blocks x 32 lanes x 16 B
constexpr int W_BLOCK = 9 * W_TAP;               // 6 912 dwords = 27 648 B per 8 input channels
constexpr int W_DWORDS = (KC / 8) * W_BLOCK;     // per chunk
constexpr int LDS_BYTES = (W_DWORDS + X_FLOATS) * 4;     // 66 816: two workgroups per CU
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
constexpr int W_INSTR = W_BLOCK / 256;           // 27 DMA instructions per 8-channel block
constexpr int X_ITEMS = KC * 5 / 4;              // (channel, row) pairs per wave: 20
}  // namespace tc

__global__ __launch_bounds__(256) void k_split_w_t(unsigned* __restrict__ wb, const float* __restrict__ wt, int C, int N,
                                                   int ldw, int tiles_n) {
    const int64_t total = (int64_t)(C / 8) * tiles_n * 9 * 2 * 32;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int l31 = (int)(i & 31), nb = (int)((i >> 5) & 1);
        int64_t r = i >> 6;
        const int tap = (int)(r % 9);
        r /= 9;
        const int tn = (int)(r % tiles_n), g = (int)(r / tiles_n);
        const int n = tn * tc::NT + nb * 32 + l31;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = n < N ? wt[((int64_t)tap * C + g * 8 + j) * ldw + n] : 0.0f;
        u32x4 h1, h2, h3;
        split8(x, h1, h2, h3);
        unsigned* dst = wb + ((int64_t)(g * tiles_n + tn) * 9 + tap) * tc::W_TAP + (nb * 32 + l31) * 4;
        *reinterpret_cast<u32x4*>(dst) = h1;
        *reinterpret_cast<u32x4*>(dst + 2 * 32 * 4) = h2;
        *reinterpret_cast<u32x4*>(dst + 2 * 2 * 32 * 4) = h3;
    }
}

__global__ __launch_bounds__(256, 2) void k_convt_bf16x3(const PS2 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    int bid = blockIdx.x;
    const int tn = bid % p.tiles_n;
    bid /= p.tiles_n;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y, b = bid / p.tiles_y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int a0 = ty * 4, b0 = tx * 32, n0 = tn * tc::NT;
    const int plane_in = p.IH * p.IW;
    const int nchunk = p.C / tc::KC;
    typedef const float __attribute__((address_space(4)))* cptr_t;

    const unsigned* wsrc = p.wb + (int64_t)tn * tc::W_BLOCK + lane * 4;
    const int64_t w_block = (int64_t)p.tiles_n * tc::W_BLOCK;      // between consecutive 8-channel blocks
    const float* isb = p.iscale ? p.iscale + (int64_t)b * p.C : nullptr;
    float* sXw = reinterpret_cast<float*>(smem + tc::W_DWORDS);
    // patch staging: wave w stages channels 4 w .. 4 w + 3, five rows each; lane = patch column (33 used).  Row a0 - 1 + r,
    // column b0 - 1 + lane: zero outside the image (top row / left column of the map), clamped address.
    const int col = b0 - 1 + lane;
    const bool col_ok = lane < 33 && col >= 0;
    const int colc = col_ok ? col : 0;
    int xoff[5];
    bool xok[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int row = a0 - 1 + r;
        xok[r] = col_ok && row >= 0;
        xoff[r] = (row >= 0 ? row : 0) * p.IW + colc;
    }
    const float* xin = p.in + (int64_t)b * p.C * plane_in;

    f32x16 acc[4][2];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ph][i][r] = 0.0f;

    for (int g = 0; g < nchunk; ++g) {
        __syncthreads();              // every wave is done reading the buffer
        {
#pragma unroll
            for (int q = 0; q < tc::KC / 8; ++q) {
                const unsigned* src = wsrc + (int64_t)(g * (tc::KC / 8) + q) * w_block;
                unsigned* dst = smem + q * tc::W_BLOCK;
#pragma unroll
                for (int i = 0; i < tc::W_INSTR / 4; ++i) {
                    const int j = wave + 4 * i;
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + j * 256), (lptr_t)(dst + j * 256), 16, 0, 0);
                }
                if (wave < tc::W_INSTR % 4) {
                    const int j = wave + 4 * (tc::W_INSTR / 4);
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + j * 256), (lptr_t)(dst + j * 256), 16, 0, 0);
                }
            }
            float sc[4] = {1.0f, 1.0f, 1.0f, 1.0f};
            if (isb) {
                const cptr_t c = (cptr_t)(isb + g * tc::KC + 4 * wave);
                sc[0] = c[0]; sc[1] = c[1]; sc[2] = c[2]; sc[3] = c[3];
            }
#ifndef SR_ABL_NOFLIP
            if ((g >> SR_FLIP_LOG2) & 1) {                 // floor-bias cancellation, as in k_conv_s2_bf16x3
                sc[0] = -sc[0]; sc[1] = -sc[1]; sc[2] = -sc[2]; sc[3] = -sc[3];
            }
#endif
            const float* xw = xin + (int64_t)(g * tc::KC + 4 * wave) * plane_in;
            float stX[tc::X_ITEMS];
#pragma unroll
            for (int k = 0; k < tc::X_ITEMS; ++k) stX[k] = xw[(k / 5) * plane_in + xoff[k % 5]];
#if !defined(SR_ABL_NOFLIP) && !defined(SR_FLIP_AT_MFMA)
            if (SR_FLIP_LOG2 == 0 || (g & ((1 << SR_FLIP_LOG2) - 1)) == 0) {
#pragma unroll
                for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[ph][i][r] = -acc[ph][i][r];
            }
#endif
            if (lane < 33) {
                float* dw = sXw + (4 * wave) * tc::XCH + lane;
#pragma unroll
                for (int k = 0; k < tc::X_ITEMS; ++k)
                    dw[(k / 5) * tc::XCH + (k % 5) * tc::PXT] = xok[k % 5] ? stX[k] * sc[k / 5] : 0.0f;
            }
        }
        __syncthreads();              // chunk complete (the barrier drains this wave's DMA)
#if !defined(SR_ABL_NOFLIP) && defined(SR_FLIP_AT_MFMA)
        if (SR_FLIP_LOG2 == 0 || (g & ((1 << SR_FLIP_LOG2) - 1)) == 0) {
#pragma unroll
            for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ph][i][r] = -acc[ph][i][r];
        }
#endif
#pragma unroll
        for (int q = 0; q < tc::KC / 8; ++q) {
        const unsigned* sW = smem + q * tc::W_BLOCK;
        // grid point (a0 + wave, b0 + l31) sits at patch (row wave + 1, column l31 + 1)
        const float* sX = sXw + q * 8 * tc::XCH + (wave + 1) * tc::PXT + l31 + 1;
        // operands of this lane: half 0 -> S00, S01 ; half 1 -> S10, S11, S01  (S_dy_dx = in(a - dy, b - dx))
        const float* p0 = half ? sX - tc::PXT : sX;            // S00 | S10
        const float* p1 = half ? sX - tc::PXT - 1 : sX - 1;    // S01 | S11
        const float* p2 = sX - 1;                          // (half 1 of group 4) S01
        float x0[8], x1[8], x2[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            x0[c] = p0[c * tc::XCH];
            x1[c] = p1[c * tc::XCH];
            x2[c] = half ? p2[c * tc::XCH] : p0[c * tc::XCH];
        }
        u32x4 f0[3], f1[3], f2[3], f3[3];
        split8(x0, f0[0], f0[1], f0[2]);                    // group 1, 3: S00 | S10
        split8(x1, f1[0], f1[1], f1[2]);                    // group 2:    S01 | S11
        split8(x2, f2[0], f2[1], f2[2]);                    // group 4:    S00 | S01
#pragma unroll
        for (int k = 0; k < 3; ++k) {                       // group 5:    S00 | zero
            f3[k].x = half ? 0u : f0[k].x; f3[k].y = half ? 0u : f0[k].y;
            f3[k].z = half ? 0u : f0[k].z; f3[k].w = half ? 0u : f0[k].w;
        }
        // weight fragments of group q: tap TA[q] on half 0, TB[q] on half 1
        auto group = [&](int ph, const u32x4 (&bf)[3], int ta, int tb) {
            const unsigned* pw = sW + (half ? tb : ta) * tc::W_TAP + l31 * 4;
            u32x4 w1[2], w2[2], w3[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                w1[nb] = *reinterpret_cast<const u32x4*>(pw + nb * 128);
                w2[nb] = *reinterpret_cast<const u32x4*>(pw + 256 + nb * 128);
                w3[nb] = *reinterpret_cast<const u32x4*>(pw + 512 + nb * 128);
            }
#define SR_TERM(A, Bv) acc[ph][0] = mma(A[0], Bv, acc[ph][0]); acc[ph][1] = mma(A[1], Bv, acc[ph][1]);
            SR_TERM(w3, bf[0]) SR_TERM(w1, bf[2]) SR_TERM(w2, bf[1]) SR_TERM(w2, bf[0]) SR_TERM(w1, bf[1]) SR_TERM(w1, bf[0])
#undef SR_TERM
        };
        // taps: index ky * 3 + kx
        group(0, f0, 0, 6);        // (0,0): S00 W00 | S10 W20
        group(1, f0, 1, 7);        // (0,1): S00 W01 | S10 W21
        group(0, f1, 2, 8);        // (0,0): S01 W02 | S11 W22
        group(2, f2, 3, 5);        // (1,0): S00 W10 | S01 W12
        group(3, f3, 4, 4);        // (1,1): S00 W11 | zero
        }
    }

    // epilogue: grid point (a, b) -> outputs (2a + py, 2b + px); C/D layout column = lane & 31 (grid column)
    const int a = a0 + wave, bcol = b0 + l31;
#ifndef SR_ABL_NOFLIP
    if (((nchunk - 1) >> SR_FLIP_LOG2) & 1) {                 // the last chunk's sign
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ph][i][r] = -acc[ph][i][r];
    }
#endif
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nu = n0 + nb * 32 + (r & 3) + 8 * (r >> 2);
            const int n = nu + 4 * half;
            float s = 1.0f, bi = 0.0f;
            if (p.oscale) {
                const cptr_t os = (cptr_t)(p.oscale + (int64_t)b * p.N + nu);
                s = half ? os[4] : os[0];
            }
            if (p.obias) {
                const cptr_t ob = (cptr_t)(p.obias + nu);
                bi = half ? ob[4] : ob[0];
            }
            float* o = p.out + (((int64_t)b * p.N + n) * p.OH + 2 * a) * p.OW + 2 * bcol;
            o[0] = acc[0][nb][r] * s + bi;
            o[1] = acc[1][nb][r] * s + bi;
            o[p.OW] = acc[2][nb][r] * s + bi;
            o[p.OW + 1] = acc[3][nb][r] * s + bi;
        }
}

}  // namespace

bool sr_conv_s2_bf16x3_eligible(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW, int64_t OH, int64_t OW) {
    return B > 0 && C % KC == 0 && N % NT == 0 && OW % 32 == 0 && OH % 4 == 0 && IH == 2 * OH + 1 && IW == 2 * OW + 1 &&
           B * C * IH * IW < (1LL << 31) && B * N * OH * OW < (1LL << 31);
}

int64_t sr_conv_s2_bf16x3_scratch_floats(int64_t C, int64_t N) {
    return (C / KC) * (N / NT) * (int64_t)W_DWORDS + 4;
}

int sr_conv_s2_bf16x3_launch(float* out, const float* in, const float* wt, int64_t ldw, const float* iscale,
                             const float* oscale, const float* obias, int64_t B, int64_t C, int64_t N, int64_t IH,
                             int64_t IW, int64_t OH, int64_t OW, float* scratch, hipStream_t st) {
    unsigned* wb = reinterpret_cast<unsigned*>(scratch);
    const int tiles_n = (int)(N / NT);
    const int64_t items = (C / KC) * tiles_n * 9 * 4 * 32;
    hipLaunchKernelGGL(k_split_w_s2, dim3(sr_stream_grid(items, 256)), dim3(256), 0, st, wb, wt, (int)C, (int)N, (int)ldw,
                       tiles_n);
    PS2 p;
    p.in = in; p.wb = wb; p.iscale = iscale; p.oscale = oscale; p.obias = obias; p.out = out;
    p.B = (int)B; p.C = (int)C; p.N = (int)N; p.IH = (int)IH; p.IW = (int)IW; p.OH = (int)OH; p.OW = (int)OW;
    p.tiles_x = (int)(OW / 32); p.tiles_y = (int)(OH / 4); p.tiles_n = tiles_n;
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_s2_bf16x3), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS_BYTES);
        configured = true;
    }
    hipLaunchKernelGGL(k_conv_s2_bf16x3, dim3((unsigned)(B * p.tiles_y * p.tiles_x * tiles_n)), dim3(THREADS), LDS_BYTES, st,
                       p);
    return sr_launch_status();
}

// ---- transposed --------------------------------------------------------------------------------------------------
bool sr_convt_bf16x3_eligible(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW) {
    return B > 0 && C % tc::KC == 0 && N % tc::NT == 0 && IW % 32 == 0 && IH % 4 == 0 &&
           B * C * IH * IW < (1LL << 31) && B * N * (2 * IH + 1) * (2 * IW + 1) < (1LL << 31);
}

int64_t sr_convt_bf16x3_scratch_floats(int64_t C, int64_t N) {
    return (C / 8) * (N / tc::NT) * (int64_t)tc::W_BLOCK + 4;
}

// interior of the map only (grid points a < IH, b < IW, all four phases); the caller runs the border strips
int sr_convt_bf16x3_launch(float* out, const float* in, const float* wt, int64_t ldw, const float* iscale,
                           const float* oscale, const float* obias, int64_t B, int64_t C, int64_t N, int64_t IH,
                           int64_t IW, float* scratch, hipStream_t st) {
    unsigned* wb = reinterpret_cast<unsigned*>(scratch);
    const int tiles_n = (int)(N / tc::NT);
    const int64_t items = (C / 8) * tiles_n * 9 * 2 * 32;
    hipLaunchKernelGGL(k_split_w_t, dim3(sr_stream_grid(items, 256)), dim3(256), 0, st, wb, wt, (int)C, (int)N, (int)ldw,
                       tiles_n);
    PS2 p;
    p.in = in; p.wb = wb; p.iscale = iscale; p.oscale = oscale; p.obias = obias; p.out = out;
    p.B = (int)B; p.C = (int)C; p.N = (int)N; p.IH = (int)IH; p.IW = (int)IW; p.OH = (int)(2 * IH + 1);
    p.OW = (int)(2 * IW + 1);
    p.tiles_x = (int)(IW / 32); p.tiles_y = (int)(IH / 4); p.tiles_n = tiles_n;
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_convt_bf16x3), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  tc::LDS_BYTES);
        configured = true;
    }
    hipLaunchKernelGGL(k_convt_bf16x3, dim3((unsigned)(B * p.tiles_y * p.tiles_x * tiles_n)), dim3(tc::THREADS),
                       tc::LDS_BYTES, st, p);
    return sr_launch_status();
}
