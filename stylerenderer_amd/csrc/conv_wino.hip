// Winograd F(2x2, 3x3) convolution on the gfx950 matrix cores (fp32, v_mfma_f32_32x32x2_f32).
//
// The stride-1 3x3 modulated convolution (forward and data-gradient: 43 % of the generator step as a
// direct implicit GEMM, csrc/conv_mfma.hip) does 36 multiply-adds per 2x2 output tile, input channel
// and output channel.  The minimal-filtering form needs 16:
//
//     Y = A^T [ sum_c (G g_c G^T) (.) (B^T d_c B) ] A          d: 4x4 input tile, g: 3x3 filter
//
// i.e. 16 independent GEMMs  M[pos][n][tile] = sum_c U[pos][c][n] * V[pos][c][tile]  — 2.25x less
// matrix-core work for the same result up to fp32 round-off (the same algorithm MIOpen / cuDNN pick
// for fp32 3x3 convolutions, i.e. what the reference's F.conv2d runs on a GPU).
//
// One 256-thread workgroup = 64 tiles (16 x 4 -> 32 x 8 output pixels) x 64 output channels:
//   * wave (i, j) owns tile block i (32 tiles) and channel block j (32 channels) for ALL 16 positions:
//     16 accumulator tiles of 32x32 = 256 registers, so the output transform A^T M A is register-local
//     (the 16 positions of a (tile, channel) pair sit in the same lane).
//   * K loop over chunks of KC input channels.  Per chunk: the halo patch d (10 x 40 floats per
//     channel, 16-byte aligned rows) and the pre-transformed weights U (a contiguous 32 KB block, see
//     k_wino_weights) arrive by LDS-DMA; the input transform B^T d B (32 add/sub per tile-channel, the
//     style s[b,c] multiplied in) runs on the VALU between the MFMAs of the PREVIOUS chunk and writes V
//     in the operand layout; operands are fetched with one conflict-free ds_read_b64 per two k-steps.
//   * pipeline (one barrier per chunk):  iteration k:  DMA d[k+2], U[k+1]  |  transform d[k+1] -> V[k+1]
//     |  MFMA over V[k], U[k].
//   * epilogue: output transform in registers, demodulation scale / bias, float2 stores (16 lanes =
//     128 contiguous bytes of an output row).
// LDS: 2 x (U 32 KB + V 32 KB + d 13 KB) + style = 156 KB of the CU's 160 KB; one workgroup per CU
// (the 256 accumulators allow one wave per SIMD anyway).
#include <type_traits>
#include "common.h"
#include "conv_wino.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

__device__ __attribute__((aligned(16))) const float g_wino_zero[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int TW = 16, TH = 4;            // tiles per workgroup: 32 x 8 output pixels
constexpr int NB = 64;                    // output channels per workgroup
constexpr int EH = 2 * TH + 2;            // halo rows
constexpr int LEAD = 3;                   // halo columns start at x0 - 4 (16-byte aligned); x0 - 1 is column 3
constexpr int EWP = 40;                   // LEAD + 2*TW + 2 = 37 -> row pitch 40 floats
constexpr int PLANE = EH * EWP;           // 400 floats per channel

#ifdef WINO_TIMING
// debug build only (scripts/build_variant.sh): per-workgroup time stamps of wave 0
__device__ long long g_wino_stamps[8 * 16384];
#define WINO_STAMP(i) do { if (tid == 0) { g_wino_stamps[(blockIdx.x & 16383) * 8 + (i)] = wall_clock64(); } } while (0)
#else
#define WINO_STAMP(i)
#endif

template <int KC>
struct WG {
    static constexpr int CQ = KC / 4;                       // channel quads per chunk
    static constexpr int PPS = CQ * 512;                    // position-pair stride in a U / V buffer
    static constexpr int UV = 8 * PPS;                      // floats per U (or V) chunk: [pos / 2][cq][e][h][64][pos % 2]
    static constexpr int U_INSTR = UV / 256;                // 16-byte DMA wave-instructions per U chunk
    static constexpr int D_FLOATS = KC * PLANE;
    static constexpr int D_INSTR = (D_FLOATS / 4 + 63) / 64;
    static constexpr int D_BUF = D_INSTR * 256;
    static constexpr int D_PER_WAVE = (D_INSTR + 3) / 4;
    static constexpr int PAD = 4 * UV + 2 * D_BUF;          // landing zone of surplus DMA instructions (3 x 1 KB)
    static constexpr int STY = PAD + 3 * 256;               // style row offset
};

template <int KC>
__global__ __launch_bounds__(256) void k_conv_wino(const WinoParams p) {
#if __HIP_DEVICE_COMPILE__   // the buffer-resource builtins exist in the device pass only; the host pass needs the stub
    using G = WG<KC>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const ubuf = smem;
    float* const vbuf = smem + 2 * G::UV;
    float* const dbuf = smem + 4 * G::UV + 1;               // + 1: see the input transform
    float* const sty = smem + G::STY;

    // ---- tile decode (XCD-chunked: consecutive ids = same input patch / neighbouring patches on one L2)
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg / SR_NUM_XCD, r = nwg % SR_NUM_XCD, xcd = bid % SR_NUM_XCD;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / SR_NUM_XCD;
    }
    const int slice = bid % p.ks;           // K slices of one tile are neighbours: same halo patch on one L2
    bid /= p.ks;
    const int n_t = bid % p.tiles_n;
    bid /= p.tiles_n;
    const int tx_i = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty_i = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int oy0 = ty_i * (2 * TH), ox0 = tx_i * (2 * TW), n0 = n_t * NB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wn = wave & 1, wt = wave >> 1;
    // this workgroup's channel range: chunks [k_lo, k_lo + nchunks) — realised by moving the bases of the two buffer
    // resources and of the style row, so the loop below runs k = 0 .. nchunks - 1 whatever the slice
    const int nchunks = p.kchunks;
    const int k_lo = slice * p.kchunks;
    const int all_chunks = p.C / KC;
    WINO_STAMP(0);
#ifdef WINO_TIMING
    const long long wino_c0 = clock64();
    if (tid == 0) {
        g_wino_stamps[(blockIdx.x & 16383) * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        g_wino_stamps[(blockIdx.x & 16383) * 8 + 7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif

    // ---- DMA descriptors of the halo patch: byte offset of this lane's 16-byte line inside the sample (chunk 0);
    // lines outside the image get an offset beyond the buffer's range: a buffer load returns zeros for them.
    // Buffer addressing keeps the per-chunk part of every address in SGPRs (soffset): no VALU work per DMA.
    int d_off[G::D_PER_WAVE];
#pragma unroll
    for (int i = 0; i < G::D_PER_WAVE; ++i) {
        const int j = wave + 4 * i;
        const int f = (j * 64 + lane) * 4;
        int off = 0x7FFFFFF0;
        if (j < G::D_INSTR && f < G::D_FLOATS) {
            const int c = f / PLANE, q = f % PLANE;
            const int r = q / EWP, cola = q % EWP;
            const int gy = oy0 - 1 + r, gx = ox0 - 4 + cola;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx + 3 < p.W) off = ((c * p.H + gy) * p.W + gx) * 4;
        }
        d_off[i] = off;
    }
    const int chunk_in_bytes = KC * p.H * p.W * 4;
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in + ((int64_t)b * p.C + (int64_t)k_lo * KC) * p.H * p.W), 0, nchunks * chunk_in_bytes,
        0x00020000);
    const __amdgpu_buffer_rsrc_t r_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.u + ((int64_t)n_t * all_chunks + k_lo) * G::UV), 0, nchunks * G::UV * 4, 0x00020000);
    const int u_voff = lane * 16;

    // one DMA instruction each (all waves issue the same number; surplus ones land in the pad zone)
    auto dma_d1 = [&](int k, int buf, int i) {
        const int j = wave + 4 * i;
        float* dst = j < G::D_INSTR ? dbuf + buf * G::D_BUF + j * 256 : smem + G::PAD + (wave - 1) * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in, (lptr_t)dst, 16, d_off[i], k * chunk_in_bytes, 0, 0);
    };
    auto dma_u1 = [&](int k, int buf, int i) {
        const int j = wave + 4 * i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_u, (lptr_t)(ubuf + buf * G::UV + j * 256), 16, u_voff,
                                                 (k * G::UV + j * 256) * 4, 0, 0);
    };
    auto dma_d = [&](int k, int buf) {
#pragma unroll
        for (int i = 0; i < G::D_PER_WAVE; ++i) dma_d1(k, buf, i);
    };
    auto dma_u = [&](int k, int buf) {
#pragma unroll
        for (int i = 0; i < G::U_INSTR / 4; ++i) dma_u1(k, buf, i);
    };

    // ---- input transform item of this thread: tile = lane, channels (ca, ca + 2) of the chunk (quad t_cq, half
    // t_h, k-steps e = 0 / 1).  dbuf is shifted by one float so that column x0 - 1 of a tile sits on an even index:
    // a row of the 4x4 patch is two aligned float pairs, and the whole transform runs on (column j, column j + 1)
    // pairs with packed ops — no register shuffles between the LDS reads, the arithmetic and the LDS writes.
    static_assert(KC == 8, "two channel quads per chunk");
    const int t_cq = wave & 1, t_h = wave >> 1;
    const int t_ca = 4 * t_cq + t_h;
    const int t_rd = t_ca * PLANE + (2 * (lane >> 4)) * EWP + LEAD + 2 * (lane & 15);
    const int t_wr = t_cq * 512 + t_h * 128 + lane * 2;

    typedef float f2 __attribute__((ext_vector_type(2)));
    auto ld2 = [](const float* q) { return *reinterpret_cast<const f2*>(q); };
    auto st2 = [](float* q, f2 v) { *reinterpret_cast<f2*>(q) = v; };
    // explicit v_pk_*_f32 (the compiler scalarises ext_vector arithmetic here).  `ch` picks the style of channel a
    // (low half of s) or b (high half), broadcast to both lanes of the pair by op_sel.
    auto pk_add = [](f2 x, f2 y) { f2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
    auto pk_sub = [](f2 x, f2 y) {
        f2 r;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
        return r;
    };
    auto pk_mul_s = [](f2 x, f2 sv, int ch) {                     // x * s
        f2 r;
        if (ch == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(sv));
        else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x), "v"(sv));
        return r;
    };
    auto pk_fms_s = [](f2 x, f2 sv, f2 y, int ch) {               // x * s - y
        f2 r;
        if (ch == 0)
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]"
                : "=v"(r) : "v"(x), "v"(sv), "v"(y));
        else
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]"
                : "=v"(r) : "v"(x), "v"(sv), "v"(y));
        return r;
    };
    auto pk_fnma_s = [](f2 x, f2 sv, f2 y, int ch) {              // y - x * s
        f2 r;
        if (ch == 0)
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
                : "=v"(r) : "v"(x), "v"(sv), "v"(y));
        else
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
                : "=v"(r) : "v"(x), "v"(sv), "v"(y));
        return r;
    };
    // column stage on t = (t0, t1 | t2, t3): (t0 - t2, t1 + t2) and (t2 - t1, t1 - t3)
    auto col01 = [](f2 tp, f2 tq) {
        f2 r;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(tp), "v"(tq));
        return r;
    };
    auto col23 = [](f2 tp, f2 tq) {
        f2 r;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(tq), "v"(tp));
        return r;
    };
    // B^T d B of the two channels in four steps (the loop pins one step per MFMA slot).  Row stage with the style
    // folded in: s*d0 - s*d2, s*d1 + s*d2, s*d2 - s*d1, s*d1 - s*d3 (20 packed ops per channel pair of columns).
    struct XF {
        f2 P[2][4], Q[2][4];          // halo rows: columns (0,1), (2,3) of channel a / b
        f2 sp[2][2], sq[2][2];        // s * rows 1, 2
        f2 tp[2][4], tq[2][4];        // after the row stage
        f2 o01[2][4], o23[2][4];      // result rows q: positions (4q, 4q+1), (4q+2, 4q+3)
    };
    auto xf_read = [&](XF& x, const float* d0, int r) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            x.P[ch][r] = ld2(d0 + ch * 2 * PLANE + r * EWP);
            x.Q[ch][r] = ld2(d0 + ch * 2 * PLANE + r * EWP + 2);
        }
    };
    auto xf_step = [&](XF& x, int step, f2 sv) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            if (step == 0) {
                x.sp[ch][0] = pk_mul_s(x.P[ch][1], sv, ch);
                x.sp[ch][1] = pk_mul_s(x.P[ch][2], sv, ch);
                x.sq[ch][0] = pk_mul_s(x.Q[ch][1], sv, ch);
                x.sq[ch][1] = pk_mul_s(x.Q[ch][2], sv, ch);
                x.tp[ch][0] = pk_fms_s(x.P[ch][0], sv, x.sp[ch][1], ch);
                x.tq[ch][0] = pk_fms_s(x.Q[ch][0], sv, x.sq[ch][1], ch);
            } else if (step == 1) {
                x.tp[ch][1] = pk_add(x.sp[ch][0], x.sp[ch][1]);
                x.tq[ch][1] = pk_add(x.sq[ch][0], x.sq[ch][1]);
                x.tp[ch][2] = pk_sub(x.sp[ch][1], x.sp[ch][0]);
                x.tq[ch][2] = pk_sub(x.sq[ch][1], x.sq[ch][0]);
                x.tp[ch][3] = pk_fnma_s(x.P[ch][3], sv, x.sp[ch][0], ch);
                x.tq[ch][3] = pk_fnma_s(x.Q[ch][3], sv, x.sq[ch][0], ch);
            } else {
#pragma unroll
                for (int q = 2 * (step - 2); q < 2 * (step - 2) + 2; ++q) {
                    x.o01[ch][q] = col01(x.tp[ch][q], x.tq[ch][q]);
                    x.o23[ch][q] = col23(x.tp[ch][q], x.tq[ch][q]);
                }
            }
        }
    };
    // V layout (and U, see k_wino_weights): [position pair 8][quad 2][k-step 2][half 2][64][position parity 2]
    auto xf_write = [&](const XF& x, float* vout, int q) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            st2(vout + (2 * q) * G::PPS + ch * 256, x.o01[ch][q]);
            st2(vout + (2 * q + 1) * G::PPS + ch * 256, x.o23[ch][q]);
        }
    };
    // whole transform of chunk k (prologue only; inside the loop it is sliced between the MFMAs)
    auto transform = [&](int k, int dsel, int vsel) {
        const float* d0 = dbuf + dsel * G::D_BUF + t_rd;
        float* v = vbuf + vsel * G::UV + t_wr;
        f2 sv;
        sv.x = sty[k * KC + t_ca];
        sv.y = sty[k * KC + t_ca + 2];
        XF x;
#pragma unroll
        for (int r = 0; r < 4; ++r) xf_read(x, d0, r);
#pragma unroll
        for (int st = 0; st < 4; ++st) xf_step(x, st, sv);
#pragma unroll
        for (int q = 0; q < 4; ++q) xf_write(x, v, q);
    };

    f32x16 acc[16];
#pragma unroll
    for (int pos = 0; pos < 16; ++pos)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pos][r] = 0.0f;

    const int a_off = (half * 64 + wn * 32 + l31) * 2;
    const int b_off = (half * 64 + wt * 32 + l31) * 2;

    // ---- prologue: d[0], U[0], d[1] in flight; V[0] from d[0]
    dma_d(0, 0);
    dma_u(0, 0);
    if (nchunks > 1) dma_d(1, 1);
    // style row of this sample (ones when absent), consumed by the input transform: fetched behind the DMAs, so
    // its round trip overlaps theirs
    for (int c = tid; c < nchunks * KC; c += 256)
        sty[c] = p.iscale ? p.iscale[(int64_t)b * p.C + k_lo * KC + c] : 1.0f;
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's DMAs have landed
    __syncthreads();
    WINO_STAMP(1);
    transform(0, 0, 0);
    WINO_STAMP(2);

    static_assert(KC == 8, "the slot schedule below is written for two channel quads per chunk");
    // Operand registers of the two channel quads.  The MFMA stream runs HALF A CHUNK behind the operand fetch:
    // body k issues the second quad of chunk k-1 (registers loaded in body k-1) in slots 0-7 while it fetches the
    // first quad of chunk k, then the first quad of chunk k in slots 8-15 while it fetches the second one.  No
    // MFMA waits on an LDS read issued after the barrier (measured before: a 16-read round trip, ~500 cycles of
    // idle matrix pipe per chunk).
    f2 au[2][2][8], bv[2][2][8];                                   // [quad][k-step][position pair]
    auto mfma_step = [&](int cq, int m) {
        const int e = m >> 4, pos = m & 15, pp = pos >> 1;
        if (pos & 1) acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(au[cq][e][pp].y, bv[cq][e][pp].y, acc[pos], 0, 0, 0);
        else acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(au[cq][e][pp].x, bv[cq][e][pp].x, acc[pos], 0, 0, 0);
    };
    auto body = [&](int k, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        // V[k] complete, d[k+1] / U[k] landed (every wave drained its own DMAs), body k-1's buffers free
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        // chunks fetched during this body (clamped at the end: a redundant fetch into a free buffer keeps the
        // body branch-free): d[k+2] -> dbuf[k & 1], U[k+1] -> ubuf[(k+1) & 1]
        const int kd = min(k + 2, nchunks - 1), ku = min(k + 1, nchunks - 1);
        const float* ub = ubuf + (k & 1) * G::UV + a_off;
        const float* vb_ = vbuf + (k & 1) * G::UV + b_off;
        // transform of chunk k+1 (garbage in, unused out in the last body: no branch in this block)
        const int kn = (k + 1 < nchunks) ? k + 1 : k;
        const float* d0 = dbuf + ((k + 1) & 1) * G::D_BUF + t_rd;
        float* vout = vbuf + ((k + 1) & 1) * G::UV + t_wr;
        f2 sab;
        sab.x = sty[kn * KC + t_ca];
        sab.y = sty[kn * KC + t_ca + 2];
        XF x;
        auto load_ops = [&](int cq, int pp) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                au[cq][e][pp] = ld2(ub + pp * G::PPS + cq * 512 + e * 256);
                bv[cq][e][pp] = ld2(vb_ + pp * G::PPS + cq * 512 + e * 256);
            }
        };
        // 16 slots of 4 MFMAs; the LDS / VALU / DMA work is pinned between them (sched_barrier: nothing crosses
        // a slot edge) by the schedule strings below, one hex digit = the slot of a unit of work:
        //   WINO_RD  halo rows 0..3 of the transform item          WINO_ST  transform steps 0..3
        //   WINO_WR  operand stores of result rows 0..3            WINO_L0 / WINO_L1  operand fetches (position pairs
        //   2i, 2i+1) of the first quad of chunk k / of its second quad (consumed in slots 8+i / in the next body's i)
        //   WINO_DD  halo DMA instruction pairs 0, 1               WINO_DU  weight DMA instruction pairs 0..3
        // Every schedule that respects read -> step -> write and "refill after the last use" gives the same bits.
#ifndef WINO_RD
// round 3: L0 0123, DD 01, DU 4567 (operand fetches and DMA issue beside the transform's halo reads); round 4
// (scripts/wino_ablate.sh, 20-launch means, same box): 1.398 / 1.175 / 1.103 ms -> 1.374 / 1.143 / 1.074 at
// 128 ch 256^2 / 256 ch 128^2 / 512 ch 64^2, batch 16.  Earlier stores (WR 89AB) cost 3 %, later weight DMA (CDEF) 4 %.
#define WINO_RD "0123"
#define WINO_ST "4567"
#define WINO_WR "CDEF"
#define WINO_L0 "4567"
#define WINO_L1 "89AB"
#define WINO_DD "45"
#define WINO_DU "89AB"
#endif
        constexpr char RD[] = WINO_RD, ST[] = WINO_ST, WR[] = WINO_WR, L0[] = WINO_L0, L1[] = WINO_L1, DD[] = WINO_DD,
                       DU[] = WINO_DU;
        static_assert(sizeof(RD) == 5 && sizeof(ST) == 5 && sizeof(WR) == 5 && sizeof(L0) == 5 && sizeof(L1) == 5 &&
                      sizeof(DD) == 3 && sizeof(DU) == 5, "schedule strings");
        static_assert(G::D_PER_WAVE == 4 && G::U_INSTR / 4 == 8, "4 halo + 8 weight DMA instructions per wave");
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int cq = (s < 8) ? 1 : 0, j = s & 7;
            const char d = s < 10 ? '0' + s : 'A' + (s - 10);
            if (!(FIRST && s < 8)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) mfma_step(cq, 4 * j + u);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (L0[i] == d) { load_ops(0, 2 * i); load_ops(0, 2 * i + 1); }            // first quad of chunk k
                if (L1[i] == d) { load_ops(1, 2 * i); load_ops(1, 2 * i + 1); }            // second quad (next body)
            }
#ifndef WINO_NO_DMA
            // the halo patch (HBM for the first output-channel tile that touches it) has to be in flight long before
            // the next barrier, the (L2-resident) weights less so
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < 2 && DD[i] == d) { dma_d1(kd, k & 1, 2 * i); dma_d1(kd, k & 1, 2 * i + 1); }
                if (DU[i] == d) { dma_u1(ku, (k + 1) & 1, 2 * i); dma_u1(ku, (k + 1) & 1, 2 * i + 1); }
            }
#endif
#ifndef WINO_NO_XFORM
#pragma unroll
            for (int i = 0; i < 4; ++i) if (RD[i] == d) xf_read(x, d0, i);
#pragma unroll
            for (int i = 0; i < 4; ++i) if (ST[i] == d) xf_step(x, i, sab);
#pragma unroll
            for (int i = 0; i < 4; ++i) if (WR[i] == d) xf_write(x, vout, i);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    body(0, std::true_type{});
    for (int k = 1; k < nchunks; ++k) body(k, std::false_type{});
    // second quad of the last chunk
#pragma unroll
    for (int m = 0; m < 32; ++m) mfma_step(1, m);

    // the clamped fetches of the last iteration are still landing in this workgroup's LDS: drain them
    // before the wave can retire
    __builtin_amdgcn_s_waitcnt(0x0F70);
    WINO_STAMP(3);

    // ---- epilogue: Y = A^T M A per (tile, channel); C/D layout: column (tile) = lane & 31,
    // row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int tile = wt * 32 + l31;
    const int oy = oy0 + 2 * (tile >> 4), ox = ox0 + 2 * (tile & 15);
    const int64_t plane = (int64_t)p.H * p.W;
    // noise of this lane's 2x2 output pixels (the same for every channel): w * noise, once
    float nzv[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    if (p.nba && p.nz) {
#pragma clang fp contract(off)
        const float nw = p.nz_w[0];
        const float* nzp = p.nz + b * p.nz_bstride + (int64_t)oy * p.W + ox;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            nzv[a][0] = nw * nzp[a * p.W];
            nzv[a][1] = nw * nzp[a * p.W + 1];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m0 = acc[j][r], m1 = acc[4 + j][r], m2 = acc[8 + j][r], m3 = acc[12 + j][r];
            s[0][j] = (m0 + m1) + m2;
            s[1][j] = (m1 - m2) - m3;
        }
        if (p.ks > 1) {
            // raw slice sums; scales, bias and the fused tail are k_wino_reduce's
            float* o = p.partial + (((int64_t)slice * p.B + b) * p.N + n) * plane + (int64_t)oy * p.W + ox;
            *reinterpret_cast<float2*>(o) = make_float2((s[0][0] + s[0][1]) + s[0][2], (s[0][1] - s[0][2]) - s[0][3]);
            *reinterpret_cast<float2*>(o + p.W) = make_float2((s[1][0] + s[1][1]) + s[1][2], (s[1][1] - s[1][2]) - s[1][3]);
            continue;
        }
        const float os = p.oscale ? p.oscale[(int64_t)b * p.N + n] : 1.0f;
        const float ob = p.obias ? p.obias[n] : 0.0f;
        float* o = p.out + ((int64_t)b * p.N + n) * plane + (int64_t)oy * p.W + ox;
        const float ab = (p.nba && p.abias) ? p.abias[n] : 0.0f;
        float y[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float y0 = (s[a][0] + s[a][1]) + s[a][2];
            const float y1 = (s[a][1] - s[a][2]) - s[a][3];
            float v0 = y0 * os + ob, v1 = y1 * os + ob;
            if (p.nba) {
                // same operation order as k_nba_fwd: (y + w * noise) + bias, unfused multiply-add
#pragma clang fp contract(off)
                v0 = v0 + nzv[a][0];
                v1 = v1 + nzv[a][1];
                v0 = v0 + ab;
                v1 = v1 + ab;
                v0 = ((v0 > 0.0f) ? v0 : v0 * p.alpha) * p.gain;
                v1 = ((v1 > 0.0f) ? v1 : v1 * p.alpha) * p.gain;
            }
            y[a][0] = v0;
            y[a][1] = v1;
        }
        // (one 16-byte store per lane pair-row via a DPP swap measured 0.2 us slower than these two)
        *reinterpret_cast<float2*>(o) = make_float2(y[0][0], y[0][1]);
        *reinterpret_cast<float2*>(o + p.W) = make_float2(y[1][0], y[1][1]);
    }
    WINO_STAMP(4);
#ifdef WINO_TIMING
    if (tid == 0) g_wino_stamps[(blockIdx.x & 16383) * 8 + 5] = clock64() - wino_c0;
#endif
#endif
}

// Sums the K slices in a fixed order (deterministic) and applies what the unsplit epilogue applies, in the same order:
// y * oscale + bias, then the optional fused tail lrelu((y + w * noise) + abias) * gain.  Four pixels per lane.
__global__ __launch_bounds__(256) void k_wino_reduce(const WinoParams p) {
    const int64_t plane = (int64_t)p.H * p.W, total4 = (int64_t)p.B * p.N * plane / 4;
    const int64_t slab4 = total4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const float4* src = reinterpret_cast<const float4*>(p.partial) + i;
        float4 a = src[0];
        for (int s = 1; s < p.ks; ++s) {
            const float4 q = src[s * slab4];
            a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
        }
        const int64_t row = (i * 4) / plane;                  // b * N + n  (plane % 4 == 0)
        const int n = (int)(row % p.N), b = (int)(row / p.N);
        const int64_t pix = i * 4 - row * plane;
        const float os = p.oscale ? p.oscale[row] : 1.0f;
        const float ob = p.obias ? p.obias[n] : 0.0f;
        float v[4] = {a.x * os + ob, a.y * os + ob, a.z * os + ob, a.w * os + ob};
        if (p.nba) {
#pragma clang fp contract(off)
            const float ab = p.abias ? p.abias[n] : 0.0f;
            float nz[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (p.nz) {
                const float nw = p.nz_w[0];
                const float4 q = *reinterpret_cast<const float4*>(p.nz + b * p.nz_bstride + pix);
                nz[0] = nw * q.x; nz[1] = nw * q.y; nz[2] = nw * q.z; nz[3] = nw * q.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = v[j] + nz[j];
                t = t + ab;
                v[j] = ((t > 0.0f) ? t : t * p.alpha) * p.gain;
            }
        }
        reinterpret_cast<float4*>(p.out)[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// U[pos][c][n] = (G g G^T)[pos], written in the chunk order the kernel DMAs:
//   [n / 64][c / KC][pos / 2][cq][e][h][n % 64][pos % 2],  chunk-local channel = 4 cq + 2 e + h
template <int KC>
__global__ __launch_bounds__(64) void k_wino_weights(float* __restrict__ u, const float* __restrict__ wt, int C,
                                                     int N, int ldw) {
    using G = WG<KC>;
    const int n = blockIdx.x * 64 + threadIdx.x, c = blockIdx.y;
    float g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = wt[((int64_t)t * C + c) * ldw + n];
    float h[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        h[0][j] = g[0][j];
        h[1][j] = 0.5f * ((g[0][j] + g[2][j]) + g[1][j]);
        h[2][j] = 0.5f * ((g[0][j] + g[2][j]) - g[1][j]);
        h[3][j] = g[2][j];
    }
    const int cl = c % KC, cq = cl / 4, e = (cl % 4) / 2, hh = cl % 2;
    float* dst = u + ((int64_t)blockIdx.x * (C / KC) + c / KC) * G::UV + cq * 512 + e * 256 + hh * 128 + threadIdx.x * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<float2*>(dst + (2 * i) * G::PPS) = make_float2(h[i][0], 0.5f * ((h[i][0] + h[i][2]) + h[i][1]));
        *reinterpret_cast<float2*>(dst + (2 * i + 1) * G::PPS) = make_float2(0.5f * ((h[i][0] + h[i][2]) - h[i][1]), h[i][2]);
    }
}

template <int KC>
int launch(const WinoParams& p, float* u, const float* wt, int ldw, hipStream_t st, bool u_ready) {
    using G = WG<KC>;
    const int lds = (G::STY + p.kchunks * KC) * 4;
    auto kern = k_conv_wino<KC>;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return SR_EINVAL;
        }
        configured = true;
    }
    // (u_ready: the caller kept the scratch of an earlier call with the same weights — a frozen network)
    if (!u_ready) hipLaunchKernelGGL(k_wino_weights<KC>, dim3(p.N / 64, p.C), dim3(64), 0, st, u, wt, p.C, p.N, ldw);
    const int64_t blocks = (int64_t)p.tiles_n * p.tiles_x * p.tiles_y * p.B * p.ks;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, p);
    if (p.ks > 1) {
        const int64_t total4 = (int64_t)p.B * p.N * p.H * p.W / 4;
        int64_t grid = (total4 + 255) / 256;
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(k_wino_reduce, dim3((unsigned)grid), dim3(256), 0, st, p);
    }
    return sr_launch_status();
}

}  // namespace

#ifdef WINO_TIMING
extern "C" int sr_debug_wino_stamps(long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wino_stamps), (size_t)n * sizeof(long long));
}
#endif

bool sr_wino_eligible(int64_t B, int64_t C, int64_t N, int64_t H, int64_t W, const void* in, const void* out) {
    if (B <= 0 || C % 8 != 0 || N % 64 != 0 || H % 8 != 0 || W % 32 != 0) return false;
    // LDS: 157 KB of buffers + the style row (C floats) must fit the 160 KB of a CU
    if (C > 512 || B * ((W / 32) * (H / 8)) * (N / 64) > 0x7FFFFFFFLL) return false;
    // buffer addressing: byte offsets inside one sample / one output-channel tile of U stay below 2^31 - 16
    if (C * H * W >= (1LL << 29) - 4 || 16 * C * 64 >= (1LL << 29)) return false;
    return ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
}

int64_t sr_wino_scratch_floats(int64_t C, int64_t N) { return 16 * C * N; }

// K slices: only when the tiles alone leave most of the chip idle; equal slices of >= 8 chunks (64 channels).
// SR_WINO_SPLIT=0 disables (A/B measurements).
int sr_wino_split(int64_t B, int64_t C, int64_t N, int64_t H, int64_t W) {
    const char* e = std::getenv("SR_WINO_SPLIT");
    if (e && e[0] == '0') return 1;
    const int64_t blocks = B * (W / (2 * TW)) * (H / (2 * TH)) * (N / NB), nchunks = C / 8;
    int ks = 1;
    while (blocks * ks < 192 && ks < 8 && nchunks % (2 * ks) == 0 && nchunks / (2 * ks) >= 8) ks *= 2;
    return ks;
}
int64_t sr_wino_partial_floats(int64_t B, int64_t C, int64_t N, int64_t H, int64_t W) {
    const int ks = sr_wino_split(B, C, N, H, W);
    return ks > 1 ? (int64_t)ks * B * N * H * W : 0;
}

int sr_wino_conv3x3(float* out, const float* in, const float* wt, int64_t ldw, const float* iscale,
                    const float* oscale, const float* obias, int64_t B, int64_t C, int64_t N, int64_t H, int64_t W,
                    float* u_scratch, hipStream_t st, const WinoNba* nba, bool u_ready) {
    WinoParams p;
    p.in = in; p.u = u_scratch; p.iscale = iscale; p.oscale = oscale; p.obias = obias; p.out = out;
    p.nba = nba ? 1 : 0;
    p.nz = nba ? nba->nz : nullptr; p.nz_w = nba ? nba->nz_w : nullptr; p.abias = nba ? nba->abias : nullptr;
    p.nz_bstride = nba ? nba->nz_bstride : 0; p.alpha = nba ? nba->alpha : 0.0f; p.gain = nba ? nba->gain : 1.0f;
    p.B = (int)B; p.C = (int)C; p.N = (int)N; p.H = (int)H; p.W = (int)W;
    p.tiles_x = (int)(W / (2 * TW)); p.tiles_y = (int)(H / (2 * TH)); p.tiles_n = (int)(N / NB);
    p.ks = sr_wino_split(B, C, N, H, W);
    p.kchunks = (int)(C / 8) / p.ks;
    p.partial = u_scratch + sr_wino_scratch_floats(C, N);          // behind the U block (sr_conv2d_scratch_floats)
    return launch<8>(p, u_scratch, wt, (int)ldw, st, u_ready);
}
