// Winograd F(3x3, 2x2) weight gradient of the stride-1 3x3 convolution on the gfx950 matrix cores.
//
//   dg[c][n] = G^T [ sum_{b, tile} (B^T d B)[c] (.) (A dY A^T)[n] ] G
//
// d = 4x4 input tile of channel c (style-scaled), dY = 2x2 output-gradient tile of channel n
// (demodulation-scaled): 16 independent GEMMs  dU[pos][c][n] = sum_k V[pos][k][c] * Z[pos][k][n]  over
// k = (sample, tile) — 16 multiply-adds per tile instead of the 36 of the direct correlation
// (csrc/conv_wgrad_mfma.hip), the same 2.25x as the forward Winograd kernel (csrc/conv_wino.hip).
//
// One 256-thread workgroup = 64 input channels x 64 output channels x all 16 positions; wave (i, j)
// owns channel block i and output block j (16 accumulator tiles of 32x32 = 256 registers).  The K
// range (all tiles of all samples) is split over workgroups; partial dU slabs are summed in a fixed
// order and pulled back to 3x3 by k_wgrad_wino_finish (deterministic, no atomics).
//
// K loop in strips of 8 tiles (16 x 2 pixels), two sub-chunks of 4 tiles each:
//   * the strip's x halo (64 channels x 4 rows x 24 columns, 16-byte aligned window) and gy pixels (64 channels
//     x 2 rows x 16 columns) arrive by buffer-addressed LDS-DMA, double buffered, a strip and a half ahead; every
//     channel occupies 25 (resp. 9) float4 slots = 24 (8) payload + 1 hole, so that 16 lanes that own 16 different
//     channels hit different banks.  The x image is shifted by one float: column x0 - 1 of the first tile sits on
//     an even index, so a tile row is two aligned float pairs wherever the tile is.
//   * each (channel, tile) pair is transformed ONCE per workgroup — thread (tile = wave, channel = lane) computes
//     B^T d B of an input tile (20 packed ops on (column, column+1) pairs, style folded into the row stage) and
//     A dY A^T of an output-gradient tile (12 packed ops) — and written to LDS in MFMA operand layout
//     [position pair][k pair][k parity][64 channels][2]; the operands of a sub-chunk are 2 x 16 KB.
//   * the waves fetch their operands with one ds_read_b64 per two MFMAs; the MFMA stream runs one sub-chunk behind
//     the fetch (each slot of four MFMAs refills the registers it has just read), so no MFMA ever waits for an
//     LDS round trip behind the (one per sub-chunk) barrier.  Body j: MFMAs of sub-chunk j-1 | fetch sub-chunk j | transform sub-chunk j+1
//     | (odd j) DMA of strip (j+1)/2 + 1.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "conv_wino.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef void __attribute__((address_space(3)))* lptr_t;

constexpr int XS = 25;                      // float4 slots per channel of the x strip (4 rows x 6 + hole)
constexpr int GS = 9;                       // float4 slots per channel of the gy strip (2 rows x 4 + hole)
constexpr int X_INSTR = 64 * XS / 64;       // 25 wave-instructions
constexpr int G_INSTR = 64 * GS / 64;       // 9
constexpr int X_FLOATS = X_INSTR * 256 + 4; // 6404: the image starts one float in (see above)
constexpr int G_FLOATS = G_INSTR * 256;     // 2304
constexpr int RAW = X_FLOATS + G_FLOATS;    // 8708 floats = 34 KB per strip
constexpr int OB = 4096;                    // floats of one operand (V or Z) of a sub-chunk: [4][2][2][64][4]
constexpr int OBUF = 2 * OB;                // V then Z
constexpr int MAX_B = 32;                   // scale tables [B][64] x 2 in LDS
constexpr int OOB = 0x7FFFFFF0;             // buffer offset beyond every tensor: the load returns zeros
// NW waves per workgroup (4: one per SIMD, 8: two per SIMD, see k_wgrad_wino): DMA instructions per wave and strip,
// landing zone of the surplus ones (never read)
template <int NW>
struct WgW {
    static constexpr int PAD = NW * 256;
    static constexpr int X_PER_WAVE = (X_INSTR + NW - 1) / NW;   // 7 | 4
    static constexpr int G_PER_WAVE = (G_INSTR + NW - 1) / NW;   // 3 | 2
    static constexpr int N_DMA = X_PER_WAVE + G_PER_WAVE;        // 10 | 6
    static constexpr int NPOS = 64 / NW;                         // positions per wave: 16 | 8
    static constexpr int LDS_FLOATS = 2 * RAW + 2 * OBUF + PAD + 2 * MAX_B * 64;
};

constexpr int digit_sum(const char (&a)[9]) {
    int c = 0;
    for (int i = 0; i < 8; ++i) c += a[i] - '0';
    return c;
}

#ifdef WGW_TIMING
// debug build only (scripts/build_variant.sh): per-workgroup time stamps of wave 0
__device__ long long g_wgw_stamps[8 * 16384];
#define WGW_STAMP(i) do { if (threadIdx.x == 0) { g_wgw_stamps[(blockIdx.x & 16383) * 8 + (i)] = wall_clock64(); } } while (0)
#else
#define WGW_STAMP(i)
#endif

struct WgWinoParams {
    const float* x;
    const float* gy;
    const float* xscale;
    const float* gscale;
    float* partial;          // [slices][16][C][N]
    int B, C, N, H, W;
    int tiles_c, tiles_n, slices, chunks_per_slice, chunks_total;
    int cty, ctx;            // strip grid per sample: H/2 x W/16
};

// NW = 8 (two waves per SIMD): wave w < 4 owns positions 0..7 of its 32 x 32 (channel, output) block and transforms the
// INPUT tile (w, lane) of every sub-chunk, wave w + 4 positions 8..15 and the OUTPUT-GRADIENT tile — the same slot
// schedule with half the MFMAs, operand fetches, transform work and DMA instructions per wave, so that on every SIMD the
// vector / LDS / DMA instructions of one wave are issued while an MFMA of the other is executing
// (scripts/mfma_valu_probe.cpp: a second wave's vector instructions cost the MFMA wave nothing, the same instructions in
// the MFMA wave's own stream ~10 cycles each).  Same transforms, same MFMA chain per accumulator: bit-identical slabs.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_wgrad_wino(const WgWinoParams p) {
#if __HIP_DEVICE_COMPILE__   // buffer-resource builtins: device pass only (the host pass needs just the stub)
    using K = WgW<NW>;
    constexpr int X_PER_WAVE = K::X_PER_WAVE, G_PER_WAVE = K::G_PER_WAVE, N_DMA = K::N_DMA, NPOS = K::NPOS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const raw = smem;                          // [2][RAW]
    float* const obuf = smem + 2 * RAW;               // [2][OBUF]
    float* const pad = obuf + 2 * OBUF;
    float* const tabx = pad + K::PAD;                 // [B][64] style of this channel block
    float* const tabg = tabx + MAX_B * 64;            // [B][64] demodulation of this output block

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg / SR_NUM_XCD, r = nwg % SR_NUM_XCD, xcd = bid % SR_NUM_XCD;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / SR_NUM_XCD;
    }
    const int n_t = bid % p.tiles_n;
    bid /= p.tiles_n;
    const int c_t = bid % p.tiles_c;
    const int slice = bid / p.tiles_c;
    const int c0 = c_t * 64, n0 = n_t * 64;
    const int k_beg = slice * p.chunks_per_slice;
    const int k_end = min(p.chunks_total, k_beg + p.chunks_per_slice);
    const int nstrips = k_end - k_beg;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int ph = wave >> 2, wq = wave & 3;          // position half (NW = 8); tile of this wave's transform item
    const int wc = wq >> 1, wn = wq & 1;
    WGW_STAMP(0);
#ifdef WGW_TIMING
    const long long wgw_c0 = clock64();
    if (tid == 0) {
        g_wgw_stamps[(blockIdx.x & 16383) * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        g_wgw_stamps[(blockIdx.x & 16383) * 8 + 7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif

    // ---- DMA descriptors.  Slot s = 64 j + lane of the strip image; every wave issues the same number of
    // instructions (the surplus ones land in a pad region nobody reads).  Byte offsets are relative to the strip
    // origin (row y0 - 1, column x0 - 4 of channel c0); holes, pad slots and halo elements outside the image get
    // an out-of-range offset and receive zeros: flag bits 1 top row, 2 bottom row, 4 left column group, 8 right
    // column group (kept in the low bits of the offset, a multiple of 16) are matched against the strip's position.
    int xd[X_PER_WAVE];                      // byte offset (a multiple of 16) | flags
#pragma unroll
    for (int i = 0; i < X_PER_WAVE; ++i) {
        const int s = 64 * (wave + NW * i) + lane;
        const int c = s / XS, rem = s % XS;
        const int r = rem / 6, q = rem % 6;
        const bool payload = wave + NW * i < X_INSTR && rem < XS - 1;
        xd[i] = payload ? (((c * p.H + r) * p.W + 4 * q) * 4) | (r == 0 ? 1 : 0) | (r == 3 ? 2 : 0) | (q == 0 ? 4 : 0) | (q == 5 ? 8 : 0)
                        : OOB;
    }
    int gd_off[G_PER_WAVE];
#pragma unroll
    for (int i = 0; i < G_PER_WAVE; ++i) {
        const int s = 64 * (wave + NW * i) + lane;
        const int n = s / GS, rem = s % GS;
        gd_off[i] = (wave + NW * i < G_INSTR && rem < GS - 1) ? ((n * p.H + rem / 4) * p.W + 4 * (rem % 4)) * 4 : OOB;
    }
    const int plane = p.H * p.W;
    // x descriptor starts W + 4 floats in front of the tensor: strip offsets stay non-negative
    const __amdgpu_buffer_rsrc_t r_x = uniform_rsrc(p.x - (p.W + 4), (p.B * p.C * plane + p.W + 4) * 4);
    const __amdgpu_buffer_rsrc_t r_g = uniform_rsrc(p.gy, p.B * p.N * plane * 4);

    // strip coordinates of the next fetch (advanced incrementally, clamped at the last strip of the tensor so that a
    // surplus fetch at the end of a slice stays in bounds)
    int d_txc = k_beg % p.ctx, d_ty = (k_beg / p.ctx) % p.cty, d_b = k_beg / (p.ctx * p.cty);
    int d_xs, d_gs, d_edge;
    float* d_dst;
    auto dma_begin = [&](int buf) {          // uniform part of one strip fetch
        const int y0 = 2 * d_ty, x0 = 16 * d_txc;
        d_xs = ((d_b * p.C + c0) * plane + y0 * p.W + x0) * 4;
        d_gs = ((d_b * p.N + n0) * plane + y0 * p.W + x0) * 4;
        d_edge = (d_ty == 0 ? 1 : 0) | (d_ty == p.cty - 1 ? 2 : 0) | (d_txc == 0 ? 4 : 0) | (d_txc == p.ctx - 1 ? 8 : 0);
        d_dst = raw + buf * RAW;
        const bool last = d_txc == p.ctx - 1 && d_ty == p.cty - 1 && d_b == p.B - 1;
        if (!last && ++d_txc == p.ctx) {
            d_txc = 0;
            if (++d_ty == p.cty) { d_ty = 0; ++d_b; }
        }
    };
    auto dma1 = [&](int i) {                 // instruction i (0 .. 9) of the strip begun last
        if (i < X_PER_WAVE) {
            const int off = (xd[i] & d_edge) ? OOB : (xd[i] & ~15);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                r_x, (lptr_t)(wave + NW * i < X_INSTR ? d_dst + 1 + (wave + NW * i) * 256 : pad + wave * 256), 16, off, d_xs, 0, 0);
        } else {
            const int g = i - X_PER_WAVE;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                r_g, (lptr_t)(wave + NW * g < G_INSTR ? d_dst + X_FLOATS + (wave + NW * g) * 256 : pad + wave * 256), 16,
                gd_off[g], d_gs, 0, 0);
        }
    };

    f32x16 acc[NPOS];
#pragma unroll
    for (int pos = 0; pos < NPOS; ++pos)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pos][r] = 0.0f;

    // ---- packed fp32 helpers (explicit v_pk_*: the compiler scalarises ext_vector arithmetic here)
    typedef float f2 __attribute__((ext_vector_type(2)));
    auto ld2 = [](const float* q) { return *reinterpret_cast<const f2*>(q); };
    auto st2 = [](float* q, f2 v) { *reinterpret_cast<f2*>(q) = v; };
    typedef float f4 __attribute__((ext_vector_type(4)));
    auto ld4 = [](const float* q) { return *reinterpret_cast<const f4*>(q); };
    auto st4 = [](float* q, f2 lo, f2 hi) {        // one ds_write_b128: positions (4q .. 4q + 3) of a (k, channel)
        f4 v;
        v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
        *reinterpret_cast<f4*>(q) = v;
    };
    (void)st2;
    auto pk_mul = [](f2 a, f2 b) { f2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    auto pk_add = [](f2 a, f2 b) { f2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    auto pk_sub = [](f2 a, f2 b) {
        f2 r;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        return r;
    };
    auto pk_fms = [](f2 a, f2 b, f2 c) {         // a * b - c
        f2 r;
        asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        return r;
    };
    auto pk_fnma = [](f2 a, f2 b, f2 c) {        // c - a * b
        f2 r;
        asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        return r;
    };
    // column stage of B^T d B on t = (t0, t1 | t2, t3): (t0 - t2, t1 + t2) and (t2 - t1, t1 - t3)
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
    // rows of A dY A^T from w = (a, b): (a, a + b | a - b, -b); last row from (r, s): (-r, -r - s | -r + s, s).
    // c01 = (0, 1), c10 = (1, 0): one fused multiply-add each (a non-finite gradient turns its 0-weighted
    // neighbour into NaN instead of leaving it finite — such a step is lost either way).
    auto z_lo = [](f2 w, f2 c01) {               // (a, a + b)
        f2 r;
        asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(w), "v"(c01));
        return r;
    };
    auto z_hi = [](f2 w, f2 c10) {               // (a - b, -b)
        f2 r;
        asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]"
            : "=v"(r) : "v"(w), "v"(c10));
        return r;
    };
    auto z3_lo = [](f2 w, f2 c01) {              // (-r, -r - s)
        f2 r;
        asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,1,0] neg_lo:[1,0,1] neg_hi:[1,0,1]"
            : "=v"(r) : "v"(w), "v"(c01));
        return r;
    };
    auto z3_hi = [](f2 w, f2 c10) {              // (-r + s, s)
        f2 r;
        asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
            : "=v"(r) : "v"(w), "v"(c10));
        return r;
    };
    f2 c01, c10;
    c01.x = 0.0f; c01.y = 1.0f;
    c10.x = 1.0f; c10.y = 0.0f;

    // ---- transform item of this thread in a sub-chunk: tile = wq (k pair wq >> 1, k parity wq & 1), channel = lane.
    // Eight steps, pinned one per MFMA slot inside the loop.  NW = 4: every thread transforms the input tile AND the
    // output-gradient tile; NW = 8: waves 0..3 the input tile, waves 4..7 the output-gradient tile (uniform branches).
    const bool do_v = NW == 4 || ph == 0, do_z = NW == 4 || ph == 1;
    struct XF {
        f2 P[4], Q[4], g0, g1;            // raw: tile rows (columns 0,1 | 2,3), gradient rows
        f2 sp1, sp2, sq1, sq2;            // s * rows 1, 2
        f2 tp[4], tq[4];                  // after the row stage
        f2 o01[4], o23[4];                // B^T d B rows: positions (4q, 4q+1), (4q+2, 4q+3)
        f2 w0, w1, w2, w3;
        f2 z01[4], z23[4];
        f2 sv, gv;                        // style / demodulation of this (sample, channel), both halves
    };
    const int t_wr = (wq >> 1) * 512 + (wq & 1) * 256 + lane * 4;
    // The transform of an item is cut into micro-steps that are pinned to the MFMA slots of a body by the schedule
    // strings below (one digit = the slot, 0..7, of a micro-step); every schedule produces the same bits.
    //   input tile  (WGW_VS): load rows 0,1 + style | load rows 2,3 | row stage a | row stage b | column stage of rows
    //                         0,1 | of rows 2,3 | store positions 0..7 | store positions 8..15
    //   output tile (WGW_ZS): demodulation | load | row stage | positions 0..7 | their store | positions 8..15 | store
    // Measured (round 4, scripts/wgw_ablate.sh, 256 -> 256 at 128^2, batch 16): round 3's schedule (VS 01234556, ZS
    // 0146777, DMA instructions in slots 0..2) 1.381 ms; output-tile transform early 1.31; + both row stages of the input
    // tile in slot 2 1.30; + DMA instructions in slots 3..5 (behind the transform loads, not beside them) 1.257.  The
    // same bits every time: only the issue order of one wave's non-MFMA instructions changes.
#ifndef WGW_VS
#define WGW_VS "01223345"
#endif
#ifndef WGW_ZS
#define WGW_ZS "0011223"
#endif
    constexpr char VS[] = WGW_VS, ZS[] = WGW_ZS;
    static_assert(sizeof(VS) == 9 && sizeof(ZS) == 8, "schedule strings: 8 and 7 slots");
    auto xf_step = [&](XF& x, int step, const float* rw, int hsel, int b_smp, float* ob) {
        // rw: strip image; hsel: which half of the strip (tiles 4 hsel + wave); ob: operand buffer written
        const float* xr = rw + 1 + lane * (XS * 4) + 3 + 2 * (4 * hsel + wq);
        const float* gr = rw + X_FLOATS + lane * (GS * 4) + 2 * (4 * hsel + wq);
        const int d = '0' + step;
        if (do_v) {
            if (VS[0] == d) {
                x.P[0] = ld2(xr); x.Q[0] = ld2(xr + 2);
                x.P[1] = ld2(xr + 24); x.Q[1] = ld2(xr + 26);
                const float sx = tabx[b_smp * 64 + lane];
                x.sv.x = x.sv.y = sx;
            }
            if (VS[1] == d) {
                x.P[2] = ld2(xr + 48); x.Q[2] = ld2(xr + 50);
                x.P[3] = ld2(xr + 72); x.Q[3] = ld2(xr + 74);
            }
            if (VS[2] == d) {
                x.sp1 = pk_mul(x.P[1], x.sv); x.sp2 = pk_mul(x.P[2], x.sv);
                x.sq1 = pk_mul(x.Q[1], x.sv); x.sq2 = pk_mul(x.Q[2], x.sv);
                x.tp[0] = pk_fms(x.P[0], x.sv, x.sp2); x.tq[0] = pk_fms(x.Q[0], x.sv, x.sq2);
            }
            if (VS[3] == d) {
                x.tp[1] = pk_add(x.sp1, x.sp2); x.tq[1] = pk_add(x.sq1, x.sq2);
                x.tp[2] = pk_sub(x.sp2, x.sp1); x.tq[2] = pk_sub(x.sq2, x.sq1);
                x.tp[3] = pk_fnma(x.P[3], x.sv, x.sp1); x.tq[3] = pk_fnma(x.Q[3], x.sv, x.sq1);
            }
            if (VS[4] == d) {
                x.o01[0] = col01(x.tp[0], x.tq[0]); x.o23[0] = col23(x.tp[0], x.tq[0]);
                x.o01[1] = col01(x.tp[1], x.tq[1]); x.o23[1] = col23(x.tp[1], x.tq[1]);
            }
            if (VS[5] == d) {
                x.o01[2] = col01(x.tp[2], x.tq[2]); x.o23[2] = col23(x.tp[2], x.tq[2]);
                x.o01[3] = col01(x.tp[3], x.tq[3]); x.o23[3] = col23(x.tp[3], x.tq[3]);
            }
            if (VS[6] == d) {
#pragma unroll
                for (int q = 0; q < 2; ++q) st4(ob + t_wr + q * 1024, x.o01[q], x.o23[q]);
            }
            if (VS[7] == d) {
#pragma unroll
                for (int q = 2; q < 4; ++q) st4(ob + t_wr + q * 1024, x.o01[q], x.o23[q]);
            }
        }
        if (do_z) {
            if (ZS[0] == d) {
                const float sg = tabg[b_smp * 64 + lane];
                x.gv.x = x.gv.y = sg;
            }
            if (ZS[1] == d) {
                x.g0 = ld2(gr);
                x.g1 = ld2(gr + 16);
            }
            if (ZS[2] == d) {
                x.w0 = pk_mul(x.g0, x.gv); x.w3 = pk_mul(x.g1, x.gv);
                x.w1 = pk_add(x.w0, x.w3); x.w2 = pk_sub(x.w0, x.w3);
            }
            if (ZS[3] == d) {
                x.z01[0] = z_lo(x.w0, c01); x.z23[0] = z_hi(x.w0, c10);
                x.z01[1] = z_lo(x.w1, c01); x.z23[1] = z_hi(x.w1, c10);
            }
            if (ZS[4] == d) {
#pragma unroll
                for (int q = 0; q < 2; ++q) st4(ob + OB + t_wr + q * 1024, x.z01[q], x.z23[q]);
            }
            if (ZS[5] == d) {
                x.z01[2] = z_lo(x.w2, c01); x.z23[2] = z_hi(x.w2, c10);
                x.z01[3] = z3_lo(x.w3, c01); x.z23[3] = z3_hi(x.w3, c10);
            }
            if (ZS[6] == d) {
#pragma unroll
                for (int q = 2; q < 4; ++q) st4(ob + OB + t_wr + q * 1024, x.z01[q], x.z23[q]);
            }
        }
    };

    // ---- operand registers [k pair][position quad]: ONE set, four positions per 16-byte LDS read.  A slot issues the
    // MFMAs that read av / bz [kp][quad] and, once a quad is consumed, refills exactly those registers with the next
    // sub-chunk's values (consumed in the same slot of the next body, a whole body later).
    f4 av[2][NPOS / 4], bz[2][NPOS / 4];
    const int a_rd = half * 256 + (wc * 32 + l31) * 4 + ph * 2 * 1024;     // (NW = 8: position quads 2 ph, 2 ph + 1)
    const int b_rd = OB + half * 256 + (wn * 32 + l31) * 4 + ph * 2 * 1024;
    auto mfma_step = [&](int m) {
        const int kp = m / NPOS, pos = m % NPOS, pq = pos >> 2;
        switch (pos & 3) {
        case 0: acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp][pq].x, bz[kp][pq].x, acc[pos], 0, 0, 0); break;
        case 1: acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp][pq].y, bz[kp][pq].y, acc[pos], 0, 0, 0); break;
        case 2: acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp][pq].z, bz[kp][pq].z, acc[pos], 0, 0, 0); break;
        default: acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp][pq].w, bz[kp][pq].w, acc[pos], 0, 0, 0); break;
        }
    };

    // sample of the strip whose tiles are being transformed (advanced once per strip)
    const int per_sample = p.ctx * p.cty;
    int c_left = per_sample - (k_beg % per_sample), c_b = k_beg / per_sample;
    auto next_strip_sample = [&]() {
        if (--c_left == 0) { c_left = per_sample; ++c_b; }
    };

    // ---- prologue: strips 0 and 1 in flight, scale tables, operands of sub-chunk 0
    dma_begin(0);
#pragma unroll
    for (int i = 0; i < N_DMA; ++i) dma1(i);
    dma_begin(1);
#pragma unroll
    for (int i = 0; i < N_DMA; ++i) dma1(i);
    for (int i = tid; i < p.B * 64; i += 64 * NW) {
        const int b = i >> 6, ch = i & 63;
        tabx[i] = p.xscale ? p.xscale[(int64_t)b * p.C + c0 + ch] : 1.0f;
        tabg[i] = p.gscale ? p.gscale[(int64_t)b * p.N + n0 + ch] : 1.0f;
    }
    // the table loads were issued after the DMAs: vmcnt(10) = "strip 0 landed" cannot be told apart from them,
    // so drain everything once here
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {
        XF x;
#pragma unroll
        for (int st = 0; st < 8; ++st) xf_step(x, st, raw, 0, min(c_b, p.B - 1), obuf);
    }

    // Raw s_barrier, not __syncthreads: the release fence of the latter drains vmcnt to 0 and with it the strip
    // that is meant to stay in flight.  "My DMAs landed" (vmcnt) + "my LDS accesses retired" (lgkmcnt) before
    // the barrier is the whole protocol; the asm memory clobbers keep the compiler from moving LDS accesses
    // across it.
    auto body = [&](auto j_tag, int strip_par, auto first_tag) {
        // body j (parity j_par, compile time) of strip s = j >> 1 (parity strip_par, run time)
        constexpr int j_par = decltype(j_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        // even body: the strip fetched last (N_DMA instructions of this wave) may still be in flight
        if (j_par == 0 && NW == 4) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (j_par == 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // even body: transforms the second half of the current strip; odd body: the first half of the next one
        // (sample advanced) and starts the fetch of the strip after that into the buffer of the current one
        if (j_par == 1) { next_strip_sample(); dma_begin(strip_par); }
        const float* rw = raw + (j_par == 0 ? strip_par : strip_par ^ 1) * RAW;
        const int b_smp = min(c_b, p.B - 1);
        const float* ob_rd = obuf + j_par * OBUF;
        float* ob_wr = obuf + (j_par ^ 1) * OBUF;
        XF x;
        auto load_ops = [&](int kp, int pq) {
            av[kp][pq] = ld4(ob_rd + a_rd + pq * 1024 + kp * 512);
            bz[kp][pq] = ld4(ob_rd + b_rd + pq * 1024 + kp * 512);
        };
        // MFMAs (in sixteenths of the body's 2 NPOS) and DMA instructions per slot: WGW_MS / WGW_DS, one digit per slot
#ifndef WGW_MS
#define WGW_MS "22222222"
#endif
#ifndef WGW_DS8
#define WGW_DS8 "00022200"
#endif
#ifndef WGW_DS4
#define WGW_DS4 "00022222"
#endif
        constexpr char MS[] = WGW_MS, DS8[] = WGW_DS8, DS4[] = WGW_DS4;
        static_assert(sizeof(MS) == 9 && sizeof(DS8) == 9 && sizeof(DS4) == 9, "one digit per slot");
        const char* const DS = NW == 8 ? DS8 : DS4;
        static_assert(digit_sum(MS) == 16 && digit_sum(DS8) >= WgW<8>::N_DMA && digit_sum(DS4) >= WgW<4>::N_DMA,
                      "schedule strings: 16 sixteenths of the MFMAs, every DMA instruction of a strip");
        int m_done = 0, d_done = 0;                                     // (compile-time after unrolling)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int m_next = m_done + (MS[s] - '0') * (NPOS / 8);
            if (!FIRST) {
#pragma unroll
                for (int u = 0; u < 9 * (NPOS / 8); ++u)                  // (fixed bound: unrolls whatever the digits)
                    if (m_done + u < m_next) mfma_step(m_done + u);
            }
            // refill the quads whose MFMAs have all been issued (quad g = MFMAs 4g .. 4g + 3 of the body)
#pragma unroll
            for (int g = 0; g < NPOS / 2; ++g)
                if (4 * (g + 1) > m_done && 4 * (g + 1) <= m_next) load_ops(g / (NPOS / 4), g % (NPOS / 4));
            m_done = m_next;
#ifndef WGW_NO_DMA
            if (j_par == 1) {
#pragma unroll
                for (int u = 0; u < 9; ++u)
                    if (u < DS[s] - '0' && d_done + u < N_DMA) dma1(d_done + u);
            }
            d_done += DS[s] - '0';
#endif
#ifndef WGW_NO_XFORM
            xf_step(x, s, rw, j_par == 0 ? 1 : 0, b_smp, ob_wr);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    WGW_STAMP(1);
    body(std::integral_constant<int, 0>{}, 0, std::true_type{});
    body(std::integral_constant<int, 1>{}, 0, std::false_type{});
    for (int s = 1; s < nstrips; ++s) {      // straight-line loop body: the strip parity is a run-time LDS offset
        body(std::integral_constant<int, 0>{}, s & 1, std::false_type{});
        body(std::integral_constant<int, 1>{}, s & 1, std::false_type{});
    }
    // MFMAs of the last sub-chunk
#pragma unroll
    for (int m = 0; m < 2 * NPOS; ++m) mfma_step(m);
    // surplus fetches are still landing in this workgroup's LDS: drain them before the wave can retire
    __builtin_amdgcn_s_waitcnt(0x0F70);
    WGW_STAMP(2);

    // ---- partial dU slab of this slice: rows = channels (r & 3) + 8 (r >> 2) + 4 half, cols = l31
    float* out = p.partial + (int64_t)slice * 16 * p.C * p.N;
#pragma unroll
    for (int pos = 0; pos < NPOS; ++pos)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            out[((int64_t)(ph * 8 + pos) * p.C + c) * p.N + n0 + wn * 32 + l31] = acc[pos][r];
        }
    WGW_STAMP(3);
#ifdef WGW_TIMING
    if (tid == 0) g_wgw_stamps[(blockIdx.x & 16383) * 8 + 5] = clock64() - wgw_c0;
#endif
#endif
}

// dwt[tap][c][n] = (G^T dU G)[tap], dU = sum over slices (fixed order).  A workgroup covers 64 (c, n)
// pairs: thread (pair, g) sums positions 4g .. 4g+3 over the slices (4 independent chains, coalesced
// rows), LDS hands the 16 sums of a pair to one thread for the 4x4 -> 3x3 pull-back.
__global__ __launch_bounds__(256) void k_wgrad_wino_finish(float* __restrict__ dwt, const float* __restrict__ partial,
                                                           int slices, int C, int N) {
    __shared__ float sm[16][65];
    const int64_t cn = (int64_t)C * N;
    const int pr = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + pr;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < cn) {
        const float* src = partial + (int64_t)(4 * g) * cn + i;
#pragma unroll 4
        for (int s = 0; s < slices; ++s) {               // 16 independent loads in flight, fixed order
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += src[((int64_t)s * 16 + u) * cn];
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) sm[4 * g + u][pr] = a[u];
    __syncthreads();
    if (g != 0 || i >= cn) return;
    float m[4][4];
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) m[pos >> 2][pos & 3] = sm[pos][pr];
    float h[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[0][j] = m[0][j] + 0.5f * (m[1][j] + m[2][j]);
        h[1][j] = 0.5f * (m[1][j] - m[2][j]);
        h[2][j] = m[3][j] + 0.5f * (m[1][j] + m[2][j]);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        dwt[(t * 3 + 0) * cn + i] = h[t][0] + 0.5f * (h[t][1] + h[t][2]);
        dwt[(t * 3 + 1) * cn + i] = 0.5f * (h[t][1] - h[t][2]);
        dwt[(t * 3 + 2) * cn + i] = h[t][3] + 0.5f * (h[t][1] + h[t][2]);
    }
}

void plan(int64_t B, int64_t C, int64_t N, int64_t H, int64_t W, int& slices, int& cps, int& total) {
    total = (int)(B * (H / 2) * (W / 16));
    const int tiles = (int)((C / 64) * (N / 64));
    int want = (2 * SR_NUM_CU + tiles - 1) / tiles;          // ~2 workgroups per CU over the launch
    if (want < 1) want = 1;
    cps = (total + want - 1) / want;
    if (cps < 16) cps = total < 16 ? total : 16;             // keep the K loop long enough to pipeline
    slices = (total + cps - 1) / cps;
}

}  // namespace

#ifdef WGW_TIMING
extern "C" int sr_debug_wgw_stamps(long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wgw_stamps), (size_t)n * sizeof(long long));
}
#endif

bool sr_wgrad_wino_eligible(int64_t B, int64_t C, int64_t N, int64_t H, int64_t W, const void* x, const void* gy) {
    if (B <= 0 || B > MAX_B || C % 64 != 0 || N % 64 != 0 || H % 2 != 0 || W % 16 != 0) return false;
    // buffer addressing: byte offsets inside x / gy stay below 2^31 - 16
    if (B * C * H * W + W + 4 >= (1LL << 29) - 4 || B * N * H * W >= (1LL << 29) - 4) return false;
    return ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0;
}

int64_t sr_wgrad_wino_scratch_floats(int64_t B, int64_t C, int64_t N, int64_t H, int64_t W) {
    int slices, cps, total;
    plan(B, C, N, H, W, slices, cps, total);
    return (int64_t)slices * 16 * C * N;
}

int sr_wgrad_wino_3x3(float* dwt, const float* x, const float* gy, const float* xscale, const float* gscale,
                      int64_t B, int64_t C, int64_t N, int64_t H, int64_t W, float* scratch, hipStream_t st) {
    WgWinoParams p;
    p.x = x; p.gy = gy; p.xscale = xscale; p.gscale = gscale; p.partial = scratch;
    p.B = (int)B; p.C = (int)C; p.N = (int)N; p.H = (int)H; p.W = (int)W;
    p.tiles_c = (int)(C / 64); p.tiles_n = (int)(N / 64);
    p.cty = (int)(H / 2); p.ctx = (int)(W / 16);
    plan(B, C, N, H, W, p.slices, p.chunks_per_slice, p.chunks_total);
    // SR_WGW_WAVES=4: one wave per SIMD (round 2's form); default 8: two per SIMD (same bits)
    const char* e = getenv("SR_WGW_WAVES");
    const bool two = !(e && e[0] == '4');
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_wino<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_wino<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return SR_EINVAL;
        }
        configured = true;
    }
    const int64_t blocks = (int64_t)p.tiles_c * p.tiles_n * p.slices;
    if (two) hipLaunchKernelGGL(k_wgrad_wino<8>, dim3((unsigned)blocks), dim3(512), WgW<8>::LDS_FLOATS * 4, st, p);
    else hipLaunchKernelGGL(k_wgrad_wino<4>, dim3((unsigned)blocks), dim3(256), WgW<4>::LDS_FLOATS * 4, st, p);
    const int64_t cn = C * N;
    hipLaunchKernelGGL(k_wgrad_wino_finish, dim3((unsigned)sr_ceil_div(cn, 64)), dim3(256), 0, st, dwt, scratch,
                       p.slices, (int)C, (int)N);
    return sr_launch_status();
}
