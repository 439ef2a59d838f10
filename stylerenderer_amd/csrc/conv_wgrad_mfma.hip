// Weight gradient of the (modulated) convolution on the gfx950 matrix cores, exact fp32 MFMA.
//
//   D[tap][u][v] = sum_{b, p in grid}  (uscale[b,u] * U[b, u, p*IS + d0 + tap]) * (vscale[b,v] * V[b, v, p])
//
// U is the operand read through the tap window, V the operand on the base grid:
//   * regular conv  y = conv(x, W):      U = x (u = input channel),  V = dL/dy (v = output channel)
//   * transposed conv (upsampling, out[2y+ky] += x[y] W[ky]):  U = dL/dy read with stride 2, V = x
// The modulation of reference layers.py:295-299 enters as the per-(sample, channel) scales, so the
// per-sample weight gradients [B, Cout, Cin, k, k] the reference's grouped convolution would
// produce are never materialised: the batch is part of the K (pixel) dimension.
//
// GEMM per tap:  D[u][v] = sum_k A[u][k] * Bm[k][v],  k = pixel.  A workgroup owns a (UT x VT) tile
// of channels for ALL taps of the window (each staged U halo patch feeds every tap), a wave owns
// 32 x 32 x taps = 9 accumulator tiles (144 registers).  K is split over workgroups
// (grid = channel tiles x K slices ~ 2 x 256 CUs); slices write fp32 partial slabs that a second
// kernel sums in a fixed order (deterministic; no float atomics) while applying the output layout.
//
// Data movement: one 64-pixel patch per stage, LDS double buffered.  A staging instruction covers
// 64 consecutive floats of ONE channel plane (coalesced 256 B), so its per-lane source offset
// depends only on the lane (computed once per patch) and the channel only adds a wave-uniform
// stride.  The loads of patch i+1 are software-pipelined INTO the MFMA stream of patch i: a few
// global loads per pixel pair, written to the other LDS buffer four pixel pairs later (counted
// vmcnt, ~a dozen staging registers in flight), so the matrix pipe never waits for memory.
// (4-byte LDS-DMA was measured at ~90 cycles per instruction per CU — 62 TFLOP/s here — and
// 16-byte DMA needs 16-byte aligned planes, which halo rows are not: profiles/r01_notes.md.)
// Planes have an ODD pitch: the 32 lanes of an operand fetch (32 channels, same pixel) hit 32
// different banks.
// The modulation scales are fetched per patch as two small LDS rows and multiplied onto the
// operands after the ds_read (never onto freshly loaded registers: no wait on memory in the loop).
#include "common.h"
#include "conv_wgrad_bf16x3.h"
#include "conv_wino.h"

#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

__device__ __attribute__((aligned(16))) const float g_wzero_line[4] = {0.f, 0.f, 0.f, 0.f};
__device__ __attribute__((aligned(16))) const float g_wones_line[4] = {1.f, 1.f, 1.f, 1.f};

struct WgradParams {
    const float* U;
    const float* V;
    const float* uscale;
    const float* vscale;
    float* partial;       // [KS][NT][UP][VP]
    int B, CU, CV;        // channels of U and V
    int UH, UW;           // extent of U
    int GH, GW;           // base grid (= extent of V)
    int dy0, dx0;         // U coordinate = grid * IS + d0 + tap
    int tiles_x, tiles_y, tiles_b;    // patches
    int tiles_u, tiles_v;
    int ks, patches_per_slice;
    int UP, VP;           // padded channel extents (tiles_u * UT, tiles_v * VT)
};

// NG = number of 4-wave groups per workgroup.  With NG = 2 (512 threads) the two groups share
// the staged patch and split its pixel pairs (rows), each with its own accumulators and its own
// partial slab: two waves per SIMD cover each other's LDS / barrier stalls while the 133 KB
// double buffer still fits one workgroup per CU.
template <int IS, int TY, int TX, int PW, int PH, int PB, int UT, int VT, int NG>
struct WG {
    static constexpr int NWAVE = 4 * NG, THREADS = 256 * NG;
    static constexpr int PHG = PH / NG;                     // patch rows per group
    static_assert(NG == 1 || (PB == 1 && PH % NG == 0), "row split needs a single-sample patch");
    static constexpr int NT = TY * TX;
    static constexpr int EH = (PH - 1) * IS + TY;
    static constexpr int EW = (PW - 1) * IS + TX;
    static constexpr int NU = PB * EH * EW;                 // floats of one U channel plane
    static constexpr int USLOT = (NU + 63) / 64;            // staging instructions per U plane
    static constexpr int UPL = USLOT * 64 + 1;              // odd plane pitch
    static constexpr int NPIX = PB * PH * PW;               // 64
    static constexpr int VPL = NPIX + 1;
    static constexpr int WU = UT / 32, WV = VT / 32;
    static constexpr int SC = PB * (UT + VT);               // scale rows: [pb][UT] then [pb][VT]
    static constexpr int SC_ITEMS = (SC + THREADS - 1) / THREADS;   // one float per thread per item
    static constexpr int OFF_V = UT * UPL;
    static constexpr int OFF_S = OFF_V + VT * VPL;
    static constexpr int BUF = ((OFF_S + SC_ITEMS * THREADS + 3) / 4) * 4;
    static constexpr int LDS_BYTES = 2 * BUF * 4;
    // staging items of ONE wave per stage: its U channels x slots, its V channels, the scale rows
    static constexpr int U_ITEMS = (UT / NWAVE) * USLOT, V_ITEMS = VT / NWAVE;
    static constexpr int ITEMS = U_ITEMS + V_ITEMS + SC_ITEMS;
    static constexpr int KSTEPS = NPIX / 2 / NG;            // pixel pairs per stage and group
#ifndef SR_WGRAD_LAG
#define SR_WGRAD_LAG 4
#endif
    static constexpr int LAG = SR_WGRAD_LAG;                // pixel pairs between a load and its LDS write
    static constexpr int PER_STEP = (ITEMS + (KSTEPS - LAG) - 1) / (KSTEPS - LAG);
    static_assert(PER_STEP * (KSTEPS - LAG) >= ITEMS, "staging must finish inside the stage");
};

template <int IS, int TY, int TX, int PW, int PH, int PB, int UT, int VT, int NG>
__global__ __launch_bounds__(256 * NG) void k_wgrad_mfma(const WgradParams p) {
    using G = WG<IS, TY, TX, PW, PH, PB, UT, VT, NG>;
    static_assert(G::WU * G::WV == 4, "4 waves per workgroup");
    static_assert(G::NPIX == 64, "one V plane = one staging instruction");
    static_assert(PW % 2 == 0, "pixel pairs run along x");
    extern __shared__ __attribute__((aligned(16))) float smem[];      // the ONLY LDS object

    int bid = blockIdx.x;
    const int tile_uv = bid % (p.tiles_u * p.tiles_v);
    const int slice = bid / (p.tiles_u * p.tiles_v);
    const int u0 = (tile_uv / p.tiles_v) * UT, v0 = (tile_uv % p.tiles_v) * VT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int grp = wave >> 2, w4 = wave & 3;            // group = which rows of the patch
    const int wu = w4 / G::WV, wv = w4 % G::WV;
    const int a_base = (wu * 32 + l31) * G::UPL + half * IS + grp * G::PHG * IS * G::EW;
    const int b_base = G::OFF_V + (wv * 32 + l31) * G::VPL + half + grp * G::PHG * PW;

    const int npatch = p.tiles_x * p.tiles_y * p.tiles_b;
    const int first = slice * p.patches_per_slice;
    int last = first + p.patches_per_slice;
    if (last > npatch) last = npatch;
    const int plane_u = p.UH * p.UW, plane_v = p.GH * p.GW;

    // ---- per-lane, patch-invariant decode of the staging slots
    int u_r[G::USLOT], u_c[G::USLOT], u_pb[G::USLOT];      // window row / col / sample of slot float
#pragma unroll
    for (int s = 0; s < G::USLOT; ++s) {
        const int f = s * 64 + lane;
        u_c[s] = f % G::EW;
        u_r[s] = (f / G::EW) % G::EH;
        u_pb[s] = (f < G::NU) ? f / (G::EW * G::EH) : -1;
    }
    const int v_px = lane % PW, v_py = (lane / PW) % PH, v_pb = lane / (PW * PH);
    // LDS write bases of this lane (channel `wave`, slot 0); items add immediates
    const int uw_base = wave * G::UPL + lane;
    const int vw_base = G::OFF_V + wave * G::VPL + lane;

    // ---- per-patch state, refreshed by prepare(): clamped per-lane offsets + validity
    int u_off[G::USLOT];
    bool u_ok[G::USLOT];
    int v_off;
    bool v_ok;
    int pat_b0 = 0;
    auto prepare = [&](int patch) {
        const int tx_i = patch % p.tiles_x;
        const int ty_i = (patch / p.tiles_x) % p.tiles_y;
        const int tb_i = patch / (p.tiles_x * p.tiles_y);
        const int gy0 = ty_i * PH, gx0 = tx_i * PW, b0 = tb_i * PB;
        const int iy0 = gy0 * IS + p.dy0, ix0 = gx0 * IS + p.dx0;
        pat_b0 = b0;
#pragma unroll
        for (int s = 0; s < G::USLOT; ++s) {
            const int gy = iy0 + u_r[s], gx = ix0 + u_c[s], b = b0 + u_pb[s];
            u_ok[s] = u_pb[s] >= 0 && b < p.B && gy >= 0 && gy < p.UH && gx >= 0 && gx < p.UW;
            u_off[s] = u_ok[s] ? (b * p.CU) * plane_u + gy * p.UW + gx : 0;
        }
        const int gy = gy0 + v_py, gx = gx0 + v_px, b = b0 + v_pb;
        v_ok = b < p.B && gy < p.GH && gx < p.GW;
        v_off = v_ok ? (b * p.CV) * plane_v + gy * p.GW + gx : 0;
    };

    // Staging item k of this wave (k is a compile-time constant wherever these are called):
    //   k <  U_ITEMS            : U channel wave + 4*(k / USLOT), slot k % USLOT
    //   k <  U_ITEMS + V_ITEMS  : V channel wave + 4*(k - U_ITEMS)
    //   else                    : 256 floats of the scale rows (one per thread)
    // Every load is unconditional: the address is a wave-uniform channel base (SGPR) plus a
    // clamped per-lane offset; masks are applied when the value is written to LDS.
    auto item_load = [&](int k) -> float {
        if (k < G::U_ITEMS) {
            const int cu = k / G::USLOT, s = k % G::USLOT;
            const int u = min(u0 + wave + G::NWAVE * cu, p.CU - 1);
            return (p.U + (int64_t)u * plane_u)[u_off[s]];
        }
        if (k < G::U_ITEMS + G::V_ITEMS) {
            const int v = min(v0 + wave + G::NWAVE * (k - G::U_ITEMS), p.CV - 1);
            return (p.V + (int64_t)v * plane_v)[v_off];
        }
        const int e = (k - G::U_ITEMS - G::V_ITEMS) * G::THREADS + tid;
        float val = 1.0f;
        if (e < PB * UT) {
            const int pb = e / UT, u = e % UT;
            if (p.uscale && pat_b0 + pb < p.B && u0 + u < p.CU) val = p.uscale[(int64_t)(pat_b0 + pb) * p.CU + u0 + u];
        } else if (e < G::SC) {
            const int pb = (e - PB * UT) / VT, v = (e - PB * UT) % VT;
            if (p.vscale && pat_b0 + pb < p.B && v0 + v < p.CV) val = p.vscale[(int64_t)(pat_b0 + pb) * p.CV + v0 + v];
        }
        return val;
    };
    auto item_store = [&](int k, float val, float* dst) {
        if (k < G::U_ITEMS) {
            const int cu = k / G::USLOT, s = k % G::USLOT;
            const bool ok = u_ok[s] && (u0 + wave + G::NWAVE * cu < p.CU);
            dst[uw_base + G::NWAVE * cu * G::UPL + s * 64] = ok ? val : 0.0f;
        } else if (k < G::U_ITEMS + G::V_ITEMS) {
            const int cv = k - G::U_ITEMS;
            const bool ok = v_ok && (v0 + wave + G::NWAVE * cv < p.CV);
            dst[vw_base + G::NWAVE * cv * G::VPL] = ok ? val : 0.0f;
        } else {
            dst[G::OFF_S + (k - G::U_ITEMS - G::V_ITEMS) * G::THREADS + tid] = val;
        }
    };

    f32x16 acc[G::NT];
#pragma unroll
    for (int t = 0; t < G::NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    if (first < last) {
        // first patch of the slice: straight load -> write, eight items at a time
        prepare(first);
#pragma unroll
        for (int k0 = 0; k0 < G::ITEMS; k0 += 8) {
            float t8[8];
#pragma unroll
            for (int d = 0; d < 8; ++d)
                if (k0 + d < G::ITEMS) t8[d] = item_load(k0 + d);
#pragma unroll
            for (int d = 0; d < 8; ++d)
                if (k0 + d < G::ITEMS) item_store(k0 + d, t8[d], smem);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    int buf = 0;
    float stg[G::ITEMS];         // staged values; live ranges span LAG pixel pairs (~a dozen registers)
    for (int patch = first; patch < last; ++patch) {
        __syncthreads();     // buffer `buf` completely written; the other buffer no longer read
        // the staging stream is branch-free (a per-step branch makes the compiler's vmcnt
        // bookkeeping conservative and collapses the pipeline): after the last patch it simply
        // re-stages that patch into the idle buffer
        prepare(patch + 1 < last ? patch + 1 : patch);
        const float* sb = smem + buf * G::BUF;
        float* so = smem + (buf ^ 1) * G::BUF;
        float a_sc[PB], b_sc[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            a_sc[pb] = sb[G::OFF_S + pb * UT + wu * 32 + l31];
            b_sc[pb] = sb[G::OFF_S + PB * UT + pb * VT + wv * 32 + l31];
        }
        // MFMA operands are register double-buffered: the ds_reads of pixel pair s+1 are issued
        // BEFORE the MFMA block of pair s (one wave per SIMD: nobody else would hide the LDS
        // latency), and multiplied by the modulation scales at the top of the next iteration.
        float a_raw[G::NT], b_raw;
        auto fetch_operands = [&](int step) {
            const int qx = step % (PW / 2), py = (step / (PW / 2)) % G::PHG, pb = step / ((PW / 2) * G::PHG);
            b_raw = sb[b_base + (pb * PH + py) * PW + 2 * qx];
            const int ua = a_base + (pb * G::EH + py * IS) * G::EW + 2 * qx * IS;
#pragma unroll
            for (int ty = 0; ty < TY; ++ty)
#pragma unroll
                for (int tx = 0; tx < TX; ++tx) a_raw[ty * TX + tx] = sb[ua + ty * G::EW + tx];
        };
        fetch_operands(0);
#pragma unroll
        for (int step = 0; step < G::KSTEPS; ++step) {
            // next patch: PER_STEP loads per pixel pair, written to the other LDS buffer LAG pixel
            // pairs later — all in the MFMA shadow
#ifndef SR_ABL_W_NODMA
#pragma unroll
            for (int d = 0; d < G::PER_STEP; ++d) {
                const int k = step * G::PER_STEP + d;
                if (k < G::ITEMS) stg[k] = item_load(k);
                const int w = (step - G::LAG) * G::PER_STEP + d;
                if (w >= 0 && w < G::ITEMS) item_store(w, stg[w], so);
            }
#endif
            const int pb = step / ((PW / 2) * G::PHG);
            const float bv = b_raw * b_sc[pb];
            float av[G::NT];
#pragma unroll
            for (int t = 0; t < G::NT; ++t) av[t] = a_raw[t] * a_sc[pb];
            if (step + 1 < G::KSTEPS) fetch_operands(step + 1);
            __builtin_amdgcn_sched_barrier(0);      // the reads go out BEFORE the MFMA block
#pragma unroll
            for (int t = 0; t < G::NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv, acc[t], 0, 0, 0);
            // keep the software pipeline as written (the scheduler would otherwise hoist every
            // staging load to the top of the stage)
            __builtin_amdgcn_sched_barrier(0);
        }
        buf ^= 1;
    }

    // partial[slice][tap][u][v]; C/D layout: column (v) = lane & 31, row (u) = (r&3) + 8*(r>>2) + 4*half
    float* dst = p.partial + ((int64_t)slice * NG + grp) * G::NT * p.UP * p.VP;
#pragma unroll
    for (int t = 0; t < G::NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = u0 + wu * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int v = v0 + wv * 32 + l31;
            dst[((int64_t)t * p.UP + u) * p.VP + v] = acc[t][r];
        }
}

// ---- stride-2 3x3 weight gradient with LDS-DMA staging -----------------------------------------------------------
// Same tiling, same k order and same arithmetic as k_wgrad_mfma<2, 3, 3, 16, 4, 1, 32, 128, 2> (bit-identical
// results), but the patch arrives by 16-byte buffer-addressed LDS-DMA instead of dword loads through registers:
// 12 DMA instructions per wave and stage replace 36 global loads + 36 LDS writes + their masks and the per-lane
// offset arithmetic of every patch.  Possible because a 16-byte DMA takes any 4-byte aligned global address and LDS
// destination (profiles/r02_notes.md), and because for pad-0 stride-2 layers (U extent 2G + 1) the window of a full
// patch never leaves the image: no zero fill, the per-lane offsets are patch-invariant and the patch origin is one
// wave-uniform soffset.
//   U: per channel and row group g (window rows 4g .. 4g + 4; row 4 is staged for both groups so that every operand
//      address is a compile-time immediate) one instruction of 45 lanes: 5 rows x 9 float4 (33 used columns + 3);
//      channel pitch 181 floats (odd: the 32 lanes of an operand fetch hit 32 banks)
//   V: one instruction stages the 64 pixels of FOUR channels l, l + 32, l + 64, l + 96 (one per wave column): the
//      channels one ds_read touches lie in 32 different instructions, pitch 257.
namespace s2d {
constexpr int RP = 36;                    // floats per staged window row
constexpr int UP = 5 * RP + 1;            // 181
constexpr int OFF_UG = 32 * UP;           // second row group
constexpr int OFF_V = 2 * OFF_UG;         // 11584
constexpr int VP = 4 * 64 + 1;            // 257
constexpr int OFF_S = OFF_V + 32 * VP;    // 19808: scale rows [32 u][128 v]
constexpr int BUF = OFF_S + 160;          // 19968 floats
constexpr int LDS_BYTES = 2 * BUF * 4;    // 159744
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
}  // namespace s2d

__global__ __launch_bounds__(512) void k_wgrad_s2_dma(const WgradParams p) {
#if __HIP_DEVICE_COMPILE__
    using namespace s2d;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    const int tile_uv = bid % (p.tiles_u * p.tiles_v);
    const int slice = bid / (p.tiles_u * p.tiles_v);
    const int u0 = (tile_uv / p.tiles_v) * 32, v0 = (tile_uv % p.tiles_v) * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int grp = wave >> 2, wv = wave & 3;
    const int a_base = grp * OFF_UG + l31 * UP + half * 2;
    const int b_base = OFF_V + l31 * VP + wv * 64 + half + grp * 2 * 16;

    const int npatch = p.tiles_x * p.tiles_y * p.tiles_b;
    const int first = slice * p.patches_per_slice;
    int last = first + p.patches_per_slice;
    if (last > npatch) last = npatch;
    const int plane_u = p.UH * p.UW, plane_v = p.GH * p.GW;

    const __amdgpu_buffer_rsrc_t r_u = uniform_rsrc(p.U, p.B * p.CU * plane_u * 4);
    const __amdgpu_buffer_rsrc_t r_v = uniform_rsrc(p.V, p.B * p.CV * plane_v * 4);
    const int u_lane_off = ((lane / 9) * p.UW + 4 * (lane % 9)) * 4;
    const int v_lane_off = ((lane >> 4) * 32 * plane_v + ((lane >> 2) & 3) * p.GW + 4 * (lane & 3)) * 4;
    const bool u_lane = lane < 45;

    int d_us = 0, d_vs = 0, pat_b = 0;       // wave-uniform byte offsets of the patch origin at channels u0 / v0
    auto prepare = [&](int patch) {
        const int tx_i = patch % p.tiles_x;
        const int ty_i = (patch / p.tiles_x) % p.tiles_y;
        pat_b = patch / (p.tiles_x * p.tiles_y);
        d_us = ((pat_b * p.CU + u0) * plane_u + ty_i * 8 * p.UW + tx_i * 32) * 4;
        d_vs = ((pat_b * p.CV + v0) * plane_v + ty_i * 4 * p.GW + tx_i * 16) * 4;
    };
    // DMA item i of this wave (compile-time i): 0..7 = U channel wave + 8 (i >> 1), row group i & 1; 8..11 = V
    // quadruple wave + 8 (i - 8)
    auto dma = [&](int i, float* dst) {
        if (i < 8) {
            const int ch = wave + 8 * (i >> 1), g = i & 1;
            if (u_lane)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_u, (lptr_t)(dst + g * OFF_UG + ch * UP), 16, u_lane_off,
                                                         d_us + (ch * plane_u + g * 4 * p.UW) * 4, 0, 0);
        } else {
            const int l = wave + 8 * (i - 8);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_v, (lptr_t)(dst + OFF_V + l * VP), 16, v_lane_off,
                                                     d_vs + l * plane_v * 4, 0, 0);
        }
    };
    auto scale_load = [&]() -> float {
        float val = 1.0f;
        if (tid < 32) {
            if (p.uscale) val = p.uscale[(int64_t)pat_b * p.CU + u0 + tid];
        } else if (tid < 160) {
            if (p.vscale) val = p.vscale[(int64_t)pat_b * p.CV + v0 + tid - 32];
        }
        return val;
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    if (first < last) {
        prepare(first);
#pragma unroll
        for (int i = 0; i < 12; ++i) dma(i, smem);
        const float sv = scale_load();
        if (tid < 160) smem[OFF_S + tid] = sv;
    }
    int buf = 0;
    for (int patch = first; patch < last; ++patch) {
        // "my DMAs landed" + "my LDS accesses retired", then the barrier: buffer `buf` complete, the other one idle
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        prepare(patch + 1 < last ? patch + 1 : patch);        // after the last patch: re-stage it into the idle buffer
        const float* sb = smem + buf * BUF;
        float* so = smem + (buf ^ 1) * BUF;
        const float a_sc = sb[OFF_S + l31], b_sc = sb[OFF_S + 32 + wv * 32 + l31];
        float a_raw[9], b_raw;
        auto fetch_operands = [&](int step) {
            const int qx = step & 7, pyl = step >> 3;
            b_raw = sb[b_base + pyl * 16 + 2 * qx];
            const int ua = a_base + 2 * pyl * RP + 4 * qx;
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) a_raw[ty * 3 + tx] = sb[ua + ty * RP + tx];
        };
        fetch_operands(0);
        float sc_next = 1.0f;
#pragma unroll
        for (int step = 0; step < 16; ++step) {
            if (step < 12) dma(step, so);
            if (step == 12) sc_next = scale_load();
            if (step == 15 && tid < 160) so[OFF_S + tid] = sc_next;
            const float bv = b_raw * b_sc;
            float av[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) av[t] = a_raw[t] * a_sc;
            if (step + 1 < 16) fetch_operands(step + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        buf ^= 1;
    }
    // surplus fetches are still landing in this workgroup's LDS: drain them before the wave can retire
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    float* dst = p.partial + ((int64_t)slice * 2 + grp) * 9 * p.UP * p.VP;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = u0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int v = v0 + wv * 32 + l31;
            dst[((int64_t)t * p.UP + u) * p.VP + v] = acc[t][r];
        }
#endif
}

// Sums the K slices in a fixed order and writes the requested layout: element (t, u, v) goes to
// out[tmap[t] * slab + u * su + v * sv].
struct ReduceParams {
    const float* partial;
    float* out;
    int ks, nt, UP, VP, CU, CV;
    int64_t slab, su, sv;
    int tmap[9];
};

__global__ __launch_bounds__(256) void k_wgrad_reduce(const ReduceParams p) {
    const int64_t total = (int64_t)p.nt * p.CU * p.CV;
    const int64_t stride_s = (int64_t)p.nt * p.UP * p.VP;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % p.CV);
        const int u = (int)((i / p.CV) % p.CU);
        const int t = (int)(i / ((int64_t)p.CV * p.CU));
        const float* src = p.partial + ((int64_t)t * p.UP + u) * p.VP + v;
        float acc = 0.0f;
#pragma unroll 8
        for (int s = 0; s < p.ks; ++s) acc += src[s * stride_s];      // loads in flight, fixed summation order
        int slab = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            if (t == k) slab = p.tmap[k];
        p.out[slab * p.slab + u * p.su + v * p.sv] = acc;
    }
}

// The same reduction for FEW outputs over MANY slices (the tiny-channel layers of the map heads: 12 ... 144 outputs, up to
// 512 slices): one wave per output, lane l adds slices l, l + 64, ... in order, then the wave's fixed-order sum — instead of
// one lane walking 512 slices (10 us in a single workgroup).
__global__ __launch_bounds__(256) void k_wgrad_reduce_wave(const ReduceParams p) {
    const int64_t total = (int64_t)p.nt * p.CU * p.CV;
    const int64_t stride_s = (int64_t)p.nt * p.UP * p.VP;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= total) return;
    const int v = (int)(i % p.CV);
    const int u = (int)((i / p.CV) % p.CU);
    const int t = (int)(i / ((int64_t)p.CV * p.CU));
    const float* src = p.partial + ((int64_t)t * p.UP + u) * p.VP + v;
    float acc = 0.0f;
    for (int s = lane; s < p.ks; s += 64) acc += src[s * stride_s];
    acc = sr_wave_sum(acc);
    if (lane != 0) return;
    int slab = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k)
        if (t == k) slab = p.tmap[k];
    p.out[slab * p.slab + u * p.su + v * p.sv] = acc;
}

inline void launch_wgrad_reduce(const ReduceParams& r, int64_t total, hipStream_t st) {
    if (total <= 1024 && r.ks >= 32)
        hipLaunchKernelGGL(k_wgrad_reduce_wave, dim3((unsigned)sr_ceil_div(total, 4)), dim3(256), 0, st, r);
    else
        hipLaunchKernelGGL(k_wgrad_reduce, dim3(sr_stream_grid(total, 256)), dim3(256), 0, st, r);
}

struct Plan {
    int pw, ph, pb, ut, vt;
    int tiles_x, tiles_y, tiles_b, tiles_u, tiles_v, ks, pps;
};

Plan make_plan(int is, int B, int CU, int CV, int GH, int GW) {
    Plan pl;
    if (is == 1) {
        if (GW > 16) { pl.pw = 32; pl.ph = 2; pl.pb = 1; }
        else if (GW > 8) { pl.pw = 16; pl.ph = 4; pl.pb = 1; }
        else if (GW > 4) { pl.pw = 8; pl.ph = 8; pl.pb = 1; }
        else { pl.pw = 4; pl.ph = 4; pl.pb = 4; }
        pl.ut = 64; pl.vt = 64;
    } else {
        // the stride-2 window patch is 4x larger: narrower U tile, and never more than 5 DMA
        // slots per plane
        if (GW > 8) { pl.pw = 16; pl.ph = 4; pl.pb = 1; }
        else { pl.pw = 8; pl.ph = 8; pl.pb = 1; }
        pl.ut = 32; pl.vt = 128;
    }
    pl.tiles_x = (GW + pl.pw - 1) / pl.pw;
    pl.tiles_y = (GH + pl.ph - 1) / pl.ph;
    pl.tiles_b = (B + pl.pb - 1) / pl.pb;
    pl.tiles_u = (CU + pl.ut - 1) / pl.ut;
    pl.tiles_v = (CV + pl.vt - 1) / pl.vt;
    const int npatch = pl.tiles_x * pl.tiles_y * pl.tiles_b;
    const int tiles_uv = pl.tiles_u * pl.tiles_v;
    // one 512-thread workgroup per CU (two 256-thread ones for the multi-sample patch type)
    const int target = (pl.pb == 1 ? 1 : 2) * SR_NUM_CU;
    int ks = (target + tiles_uv - 1) / tiles_uv;
    if (ks > npatch) ks = npatch;
    if (ks < 1) ks = 1;
    pl.pps = (npatch + ks - 1) / ks;
    pl.ks = (npatch + pl.pps - 1) / pl.pps;
    return pl;
}

constexpr int NG_OF(int pb) { return pb == 1 ? 2 : 1; }     // groups per workgroup by patch type

template <int IS, int TY, int TX, int PW, int PH, int PB, int UT, int VT>
int launch_wgrad_one(const WgradParams& p, dim3 grid, hipStream_t st) {
    constexpr int NG = NG_OF(PB);
    using G = WG<IS, TY, TX, PW, PH, PB, UT, VT, NG>;
    static_assert(G::LDS_BYTES <= 160 * 1024, "LDS budget");
    auto kern = k_wgrad_mfma<IS, TY, TX, PW, PH, PB, UT, VT, NG>;
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        configured = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(G::THREADS), G::LDS_BYTES, st, p);
    return sr_launch_status();
}

template <int TY, int TX>
int launch_wgrad_s1(const WgradParams& p, const Plan& pl, hipStream_t st) {
    const dim3 grid((unsigned)(pl.tiles_u * pl.tiles_v * pl.ks));
    if (pl.pw == 32) return launch_wgrad_one<1, TY, TX, 32, 2, 1, 64, 64>(p, grid, st);
    if (pl.pw == 16) return launch_wgrad_one<1, TY, TX, 16, 4, 1, 64, 64>(p, grid, st);
    if (pl.pw == 8) return launch_wgrad_one<1, TY, TX, 8, 8, 1, 64, 64>(p, grid, st);
    return launch_wgrad_one<1, TY, TX, 4, 4, 4, 64, 64>(p, grid, st);
}
// k_wgrad_s2_dma serves full 16 x 4 patches of whole channel tiles whose windows stay inside U (pad-0 stride-2 layers
// of 2G + 1 pixels: every up- / down-sampling convolution from 16^2 up); SR_WGRAD_DMA=0 keeps the dword staging.
bool s2_dma_ok(const WgradParams& p, const Plan& pl) {
    const char* e = std::getenv("SR_WGRAD_DMA");
    if (e && e[0] == '0') return false;
    return pl.pw == 16 && pl.ph == 4 && pl.pb == 1 && pl.ut == 32 && pl.vt == 128 && p.CU % 32 == 0 && p.CV % 128 == 0 &&
           p.GW % 16 == 0 && p.GH % 4 == 0 && p.dy0 == 0 && p.dx0 == 0 && p.UH >= 2 * p.GH + 1 && p.UW >= 2 * p.GW + 1 &&
           (int64_t)p.B * p.CU * p.UH * p.UW < (1LL << 29) && (int64_t)p.B * p.CV * p.GH * p.GW < (1LL << 29);
}

template <int TY, int TX>
int launch_wgrad_s2(const WgradParams& p, const Plan& pl, hipStream_t st) {
    const dim3 grid((unsigned)(pl.tiles_u * pl.tiles_v * pl.ks));
    if (TY == 3 && TX == 3 && s2_dma_ok(p, pl)) {
        static bool configured = false;
        if (!configured) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_s2_dma),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, s2d::LDS_BYTES);
            configured = true;
        }
        hipLaunchKernelGGL(k_wgrad_s2_dma, grid, dim3(512), s2d::LDS_BYTES, st, p);
        return sr_launch_status();
    }
    if (pl.pw == 16) return launch_wgrad_one<2, TY, TX, 16, 4, 1, 32, 128>(p, grid, st);
    return launch_wgrad_one<2, TY, TX, 8, 8, 1, 32, 128>(p, grid, st);
}

// ---- weight gradient of the map heads' 3x3 convolutions (<= 4 x <= 4 channels) ---------------------------------------
// GeneratorWithMap turns every rasterised normal map into per-pixel affine planes with ResBlocks of 3 -> 3 -> 4 channels
// (reference model.py:224-247).  On the 64 x 64 channel tiles of k_wgrad_mfma such a layer is 0.2 % useful work and a
// K-split scratch of ~150 MB: 176 us at 256^2 x 4 samples.  It is a streaming reduction: a lane owns pixels, keeps the
// 9 x 4 x 4 products in registers, the block folds them (shuffles, then LDS across its four waves) into one slab and
// k_wgrad_small3_finish adds the slabs in block order — deterministic, 7 MB of reads.
constexpr int WS_CM = 4, WS_NM = 4, WS_ROW = 9 * WS_NM, WS_ACC = WS_CM * WS_ROW;

// blockIdx.y = input channel: 36 accumulators per lane (9 taps x <= 4 output channels), eight waves per SIMD
__global__ __launch_bounds__(256) void k_wgrad_small3(float* __restrict__ partial, const float* __restrict__ x,
                                                      const float* __restrict__ gy, const float* __restrict__ xs,
                                                      const float* __restrict__ gs, int B, int C, int N, int H, int W) {
    float acc[9][WS_NM];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < WS_NM; ++n) acc[t][n] = 0.0f;
    const int c = blockIdx.y;
    const int64_t plane = (int64_t)H * W;
    const int64_t total = (int64_t)B * plane;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / plane);
        const int r = (int)(i - (int64_t)b * plane);
        const int y = r / W, xw = r - y * W;
        float g[WS_NM];
#pragma unroll
        for (int n = 0; n < WS_NM; ++n)
            g[n] = n < N ? gy[((int64_t)b * N + n) * plane + r] * (gs ? gs[b * N + n] : 1.0f) : 0.0f;
        const float s = xs ? xs[b * C + c] : 1.0f;
        const float* xp = x + ((int64_t)b * C + c) * plane;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = xw + kx - 1;
                const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? xp[(int64_t)yy * W + xx] * s : 0.0f;
#pragma unroll
                for (int n = 0; n < WS_NM; ++n) acc[ky * 3 + kx][n] += v * g[n];
            }
        }
    }
    __shared__ float fold[4][WS_ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < WS_NM; ++n) {
            float v = acc[t][n];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) fold[wave][t * WS_NM + n] = v;
        }
    __syncthreads();
    if (threadIdx.x < WS_ROW) {
        const int j = threadIdx.x;
        partial[((int64_t)blockIdx.x * WS_CM + c) * WS_ROW + j] = (fold[0][j] + fold[1][j]) + (fold[2][j] + fold[3][j]);
    }
}

// slab element j = (c, t, n).  Four lanes per element walk the slabs k = q, q + 4, ... (loads of independent slabs in
// flight), then the four strided sums are added in a fixed order: deterministic
__global__ __launch_bounds__(4 * WS_ACC) void k_wgrad_small3_finish(float* __restrict__ dwt, const float* __restrict__ partial,
                                                                    int nblk, int C, int N) {
    __shared__ float part[4][WS_ACC];
    const int j = threadIdx.x % WS_ACC, q = threadIdx.x / WS_ACC;
    float a = 0.0f;
#pragma unroll 8
    for (int k = q; k < nblk; k += 4) a += partial[(int64_t)k * WS_ACC + j];
    part[q][j] = a;
    __syncthreads();
    if (q != 0) return;
    const int n = j % WS_NM, t = (j / WS_NM) % 9, c = j / WS_ROW;
    if (c < C && n < N) dwt[((int64_t)t * C + c) * N + n] = (part[0][j] + part[1][j]) + (part[2][j] + part[3][j]);
}

bool wgrad_small_ok(int64_t C, int64_t N, int ksize, int stride, int pad, int transposed) {
    const char* e = std::getenv("SR_WGRAD_SMALL");
    if (e && e[0] == '0') return false;
    return !transposed && ksize == 3 && stride == 1 && pad == 1 && C <= WS_CM && N <= WS_NM;
}

int wgrad_small_blocks(int64_t B, int64_t IH, int64_t IW) {
    const int64_t px = B * IH * IW;
    const int64_t nb = (px + 1023) / 1024;                  // >= 4 pixels per lane: the fold costs a lane ~36 x 6 shuffles
    return (int)(nb < 1 ? 1 : (nb > 256 ? 256 : nb));
}

bool geometry(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW, int64_t OH, int64_t OW,
              int ksize, int stride, int pad, int transposed, int& is, int& GH, int& GW, int& UH,
              int& UW, int& CUc, int& CVc, int& d0) {
    if (B < 0 || C <= 0 || N <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return false;
    if (!(ksize == 3 || ksize == 1) || !(stride == 1 || stride == 2)) return false;
    is = stride;
    if (!transposed) {
        if (OH != (IH + 2 * pad - ksize) / stride + 1 || OW != (IW + 2 * pad - ksize) / stride + 1) return false;
        GH = (int)OH; GW = (int)OW; UH = (int)IH; UW = (int)IW; CUc = (int)C; CVc = (int)N; d0 = -pad;
    } else {
        if (ksize != 3 || stride != 2 || pad != 0 || OH != 2 * IH + 1 || OW != 2 * IW + 1) return false;
        GH = (int)IH; GW = (int)IW; UH = (int)OH; UW = (int)OW; CUc = (int)N; CVc = (int)C; d0 = 0;
    }
    return true;
}

}  // namespace

static bool wgrad_wino_enabled() {
    const char* e = std::getenv("SR_WINOGRAD");          // "0" keeps the direct correlation kernel
    return !(e && e[0] == '0');
}

extern "C" int64_t sr_conv2d_wgrad_scratch_floats(int64_t B, int64_t C, int64_t N, int64_t IH,
                                                  int64_t IW, int64_t OH, int64_t OW, int ksize,
                                                  int stride, int pad, int transposed) {
    int is, GH, GW, UH, UW, CUc, CVc, d0;
    if (!geometry(B, C, N, IH, IW, OH, OW, ksize, stride, pad, transposed, is, GH, GW, UH, UW, CUc, CVc, d0))
        return -1;
    if (wgrad_small_ok(C, N, ksize, stride, pad, transposed)) return (int64_t)wgrad_small_blocks(B, IH, IW) * WS_ACC + 4;
    const Plan pl = make_plan(is, (int)B, CUc, CVc, GH, GW);
    int64_t need = (int64_t)pl.ks * NG_OF(pl.pb) * ksize * ksize * (pl.tiles_u * pl.ut) * (int64_t)(pl.tiles_v * pl.vt) + 4;
    if (!transposed && ksize == 3 && stride == 1 && pad == 1 && wgrad_wino_enabled() &&
        sr_wgrad_wino_eligible(B, C, N, IH, IW, nullptr, nullptr)) {
        const int64_t w = sr_wgrad_wino_scratch_floats(B, C, N, IH, IW);
        need = need > w ? need : w;
    }
    if (!transposed && ksize == 1 && stride == 1 && sr_wgrad_bf16x3_enabled('w')) {
        const int64_t w = sr_wgrad_bf16x3_scratch_floats(B, C, N, IH * IW);
        need = need > w ? need : w;
    }
    if (ksize == 3 && stride == 2 && sr_wgrad_bf16x3_enabled('g')) {
        const int64_t w = sr_wgrad_s2_bf16x3_scratch_floats(B, CUc, CVc, GH, GW);
        need = need > w ? need : w;
    }
    return need;
}

extern "C" int sr_conv2d_wgrad_mfma(float* dwt, const float* x, const float* gy, const float* xscale,
                                    const float* gscale, int64_t B, int64_t C, int64_t N, int64_t IH,
                                    int64_t IW, int64_t OH, int64_t OW, int ksize, int stride, int pad,
                                    int transposed, float* scratch, sr_stream_t stream) {
    int is, GH, GW, UH, UW, CUc, CVc, d0;
    if (!geometry(B, C, N, IH, IW, OH, OW, ksize, stride, pad, transposed, is, GH, GW, UH, UW, CUc, CVc, d0))
        return SR_EINVAL;
    if (!dwt || !x || !gy || !scratch) return SR_EINVAL;
    if (B * C * IH * IW >= (1LL << 31) || B * N * OH * OW >= (1LL << 31)) return SR_ERANGE;
    hipStream_t st = sr_stream(stream);
    if (wgrad_small_ok(C, N, ksize, stride, pad, transposed)) {
        const int nb = B > 0 ? wgrad_small_blocks(B, IH, IW) : 0;
        if (nb > 0) {
            hipLaunchKernelGGL(k_wgrad_small3, dim3(nb, (unsigned)C), dim3(256), 0, st, scratch, x, gy, xscale, gscale, (int)B, (int)C,
                               (int)N, (int)IH, (int)IW);
        }
        hipLaunchKernelGGL(k_wgrad_small3_finish, dim3(1), dim3(4 * WS_ACC), 0, st, dwt, scratch, nb, (int)C, (int)N);
        return sr_launch_status();
    }
    if (!transposed && ksize == 3 && stride == 1 && pad == 1 && wgrad_wino_enabled() &&
        sr_wgrad_wino_eligible(B, C, N, IH, IW, x, gy))
        return sr_wgrad_wino_3x3(dwt, x, gy, xscale, gscale, B, C, N, IH, IW, scratch, st);
    if (!transposed && ksize == 1 && stride == 1 && pad == 0 && B > 0 && sr_wgrad_bf16x3_enabled('w') &&
        sr_wgrad_bf16x3_eligible(B, C, N, IH * IW, x, gy)) {
        // opt-in spike (SR_CONV_SPLIT_BF16=1): split-bf16 matrix path, same partial-slab layout and reduce
        int ks3 = 0, UP3 = 0, VP3 = 0;
        const int rc3 = sr_wgrad_bf16x3_launch(x, gy, xscale, gscale, scratch, B, C, N, IH * IW, &ks3, &UP3, &VP3, st);
        if (rc3 != SR_OK) return rc3;
        ReduceParams r;
        r.partial = scratch; r.out = dwt;
        r.ks = ks3; r.nt = 1; r.UP = UP3; r.VP = VP3; r.CU = CUc; r.CV = CVc;
        r.slab = C * N; r.su = N; r.sv = 1;
        for (int t = 0; t < 9; ++t) r.tmap[t] = t;
        const int64_t total = (int64_t)CUc * CVc;
        launch_wgrad_reduce(r, total, st);
        return sr_launch_status();
    }
    if (ksize == 3 && is == 2 && pad == 0 && sr_wgrad_bf16x3_enabled('g') &&
        sr_wgrad_s2_bf16x3_eligible(B, CUc, CVc, UH, UW, GH, GW, transposed ? x : gy)) {
        // opt-in spike (SR_CONV_SPLIT_BF16=1): the up- / down-sampling layers' weight gradient on the bf16 matrix cores
        int ks3 = 0, UP3 = 0, VP3 = 0;
        const int rc3 = sr_wgrad_s2_bf16x3_launch(transposed ? gy : x, transposed ? x : gy, transposed ? gscale : xscale,
                                                  transposed ? xscale : gscale, scratch, B, CUc, CVc, UH, UW, GH, GW, &ks3,
                                                  &UP3, &VP3, st);
        if (rc3 != SR_OK) return rc3;
        ReduceParams r;
        r.partial = scratch; r.out = dwt;
        r.ks = ks3; r.nt = 9; r.UP = UP3; r.VP = VP3; r.CU = CUc; r.CV = CVc;
        r.slab = C * N;
        r.su = transposed ? 1 : N;
        r.sv = transposed ? N : 1;
        for (int t = 0; t < 9; ++t) r.tmap[t] = t;
        const int64_t total = (int64_t)9 * CUc * CVc;
        launch_wgrad_reduce(r, total, st);
        return sr_launch_status();
    }
    const Plan pl = make_plan(is, (int)B, CUc, CVc, GH, GW);
    WgradParams p;
    p.U = transposed ? gy : x;
    p.V = transposed ? x : gy;
    p.uscale = transposed ? gscale : xscale;
    p.vscale = transposed ? xscale : gscale;
    p.partial = scratch;
    p.B = (int)B; p.CU = CUc; p.CV = CVc; p.UH = UH; p.UW = UW; p.GH = GH; p.GW = GW;
    p.dy0 = p.dx0 = d0;
    p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.tiles_b = pl.tiles_b;
    p.tiles_u = pl.tiles_u; p.tiles_v = pl.tiles_v; p.ks = pl.ks; p.patches_per_slice = pl.pps;
    p.UP = pl.tiles_u * pl.ut; p.VP = pl.tiles_v * pl.vt;
    int rc = SR_OK;
    if (B > 0) {
        if (ksize == 3 && is == 1) rc = launch_wgrad_s1<3, 3>(p, pl, st);
        else if (ksize == 3 && is == 2) rc = launch_wgrad_s2<3, 3>(p, pl, st);
        else if (ksize == 1 && is == 1) rc = launch_wgrad_s1<1, 1>(p, pl, st);
        else rc = launch_wgrad_s2<1, 1>(p, pl, st);
        if (rc != SR_OK) return rc;
    }
    ReduceParams r;
    r.partial = scratch; r.out = dwt;
    r.ks = B > 0 ? pl.ks * NG_OF(pl.pb) : 0; r.nt = ksize * ksize; r.UP = p.UP; r.VP = p.VP; r.CU = CUc; r.CV = CVc;
    r.slab = C * N;
    // dwt is [k*k][C][N]: regular conv has (u, v) = (c, n); transposed has (u, v) = (n, c)
    r.su = transposed ? 1 : N;
    r.sv = transposed ? N : 1;
    for (int t = 0; t < 9; ++t) r.tmap[t] = t;
    const int64_t total = (int64_t)r.nt * CUc * CVc;
    launch_wgrad_reduce(r, total, st);
    return sr_launch_status();
}
