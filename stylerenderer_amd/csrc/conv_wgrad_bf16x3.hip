// SPIKE (opt-in, SR_CONV_SPLIT_BF16=1): weight gradient of the 1x1 convolution on the BF16 matrix cores with fp32-level
// accuracy — every fp32 operand is split into three bf16 pieces and six of the nine cross products are accumulated in
// fp32.
//
//   D[u][v] = sum_{b, p} (uscale[b,u] * U[b,u,p]) * (vscale[b,v] * V[b,v,p])        U = x, V = dL/dy, p = pixel
//
// Why this kernel: the bf16 MFMA (v_mfma_f32_32x32x16_bf16, 16x the rate of v_mfma_f32_32x32x2_f32) wants EIGHT
// consecutive k per lane.  In NCHW the contiguous dimension is the pixel — which is the K dimension of a weight
// gradient, for both operands; the forward / data-gradient convolutions contract over channels (stride H*W) and would
// need a channel-packed activation layout first.  The 1x1 weight gradient (the discriminator's skip convolutions and
// its first layer) is the product's direct kernel where the operand fetch is a plain 32-byte LDS read.
//
// Split (exact): a = h1 + h2 + h3 with h1 = bf16(a), h2 = bf16(a - h1), h3 = a - h1 - h2 (8 + 8 + 8 significand bits:
// both subtractions are exact in fp32 and the last remainder has at most 8 significant bits).  a*b = sum_{i,j} ai*bj;
// the three terms with i + j >= 5 are below 2^-24 |a*b| and dropped: six bf16 MFMAs (192 matrix-pipe cycles per 32x32
// tile and 16 k) instead of eight fp32 MFMAs (512 cycles) — 2.67x fewer, paid for with ~5.5 VALU operations per
// operand element for the split.  Products of bf16 pairs are exact in the fp32 accumulator's input precision, the
// accumulation is fp32 like the fp32 MFMA's: the result differs from the fp32 kernel's by the dropped terms and the
// summation order only (tests/test_conv_gpu.py holds it to the SAME 2e-6 * sum|a*b| bar).
//
// Tiling: workgroup = 128 (u) x 128 (v) channels, 4 waves of 64 x 64 (2 x 2 MFMA tiles), K chunks of 32 pixels of one
// sample, staged through registers into double-buffered LDS (row pitch 36 floats: the eight lanes of a ds_read_b128
// phase hit 32 distinct banks), scales multiplied at staging.  72 KB of LDS: two workgroups per CU.  K slices write
// partial slabs in k_wgrad_mfma's layout; k_wgrad_reduce (conv_wgrad_mfma.hip) sums them in a fixed order.
#include "common.h"
#include "conv_wgrad_bf16x3.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace s1 {
constexpr int UT = 128, VT = 128, KC = 32, PITCH = 36, THREADS = 256;
constexpr int TILE = UT * PITCH;               // floats of one operand tile
constexpr int BUF = 2 * TILE;                  // U tile + V tile
constexpr int LDS_BYTES = 2 * BUF * 4;         // double buffered: 73 728 B
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
}  // namespace s1

struct P3 {
    const float* U;
    const float* V;
    const float* uscale;
    const float* vscale;
    float* partial;            // [ks][UP][VP]
    int B, CU, CV, HW;
    int cps;                   // chunks per sample = ceil(HW / KC)
    int nchunk, per_slice;
    int tiles_u, tiles_v, UP, VP;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Split3 {
    u32x4 h1, h2, h3;          // 8 bf16 each (bit patterns, two per dword: element 2i in the low half)
};

// two floats -> one dword of two round-to-nearest bf16 (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float bf16_hi(unsigned pk) { return __builtin_bit_cast(float, pk & 0xFFFF0000u); }

// a pair of floats -> its three bf16 pieces (see the header: exact three-way split): 11 VALU operations per pair
__device__ __forceinline__ void split2(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = pack_bf16(x0, x1);
    const float r0 = x0 - bf16_lo(p1), r1 = x1 - bf16_hi(p1);
    p2 = pack_bf16(r0, r1);
    const float q0 = r0 - bf16_lo(p2), q1 = r1 - bf16_hi(p2);
    p3 = pack_bf16(q0, q1);
}

__device__ __forceinline__ Split3 split8(const float4 lo, const float4 hi) {
    unsigned a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
    split2(lo.x, lo.y, a0, b0, c0);
    split2(lo.z, lo.w, a1, b1, c1);
    split2(hi.x, hi.y, a2, b2, c2);
    split2(hi.z, hi.w, a3, b3, c3);
    Split3 s;
    s.h1 = u32x4{a0, a1, a2, a3};
    s.h2 = u32x4{b0, b1, b2, b3};
    s.h3 = u32x4{c0, c1, c2, c3};
    return s;
}

__device__ __forceinline__ f32x16 mma(const u32x4 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(s1::THREADS, 2) void k_wgrad1_bf16x3(const P3 p) {
    using namespace s1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    const int tile_uv = bid % (p.tiles_u * p.tiles_v);
    const int slice = bid / (p.tiles_u * p.tiles_v);
    const int u0 = (tile_uv / p.tiles_v) * UT, v0 = (tile_uv % p.tiles_v) * VT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wu = wave >> 1, wv = wave & 1;

    const int first = slice * p.per_slice;
    int last = first + p.per_slice;
    if (last > p.nchunk) last = p.nchunk;

    // staging role of this thread: 4 float4 of the U tile and 4 of the V tile per chunk
    //   item i: row = (tid + 256 i) / 8, quad = (tid + 256 i) % 8
    const int srow = tid >> 3, squad = tid & 7;          // rows srow + 32 i
    float4 stU0, stU1, stU2, stU3, stV0, stV1, stV2, stV3;
    float scU0, scU1, scU2, scU3, scV0, scV1, scV2, scV3;
    // Every load is unconditional (clamped address: a load under a per-lane condition becomes an exec-masked branch) and
    // NOTHING is computed on the loaded values here: the scale is multiplied on when the chunk is written to LDS, after
    // the MFMA block — arithmetic at load time makes the compiler wait for the loads (vmcnt) in front of the MFMAs.
    auto load_one = [&](const float* __restrict__ base, const float* __restrict__ scale, int C, int row, int b, int p0,
                        bool pok, float& sc) -> float4 {
        const int rc = row < C ? row : C - 1;
        const int pc = pok ? p0 : 0;
        sc = scale ? scale[(int64_t)b * C + rc] : 1.0f;
        sc = (pok && row < C) ? sc : 0.0f;
        return *reinterpret_cast<const float4*>(base + ((int64_t)b * C + rc) * p.HW + pc);
    };
    auto load_chunk = [&](int chunk) {
        const int b = chunk / p.cps, p0 = (chunk - b * p.cps) * KC + 4 * squad;
        const bool pok = p0 < p.HW;
        stU0 = load_one(p.U, p.uscale, p.CU, u0 + srow, b, p0, pok, scU0);
        stU1 = load_one(p.U, p.uscale, p.CU, u0 + srow + 32, b, p0, pok, scU1);
        stU2 = load_one(p.U, p.uscale, p.CU, u0 + srow + 64, b, p0, pok, scU2);
        stU3 = load_one(p.U, p.uscale, p.CU, u0 + srow + 96, b, p0, pok, scU3);
        stV0 = load_one(p.V, p.vscale, p.CV, v0 + srow, b, p0, pok, scV0);
        stV1 = load_one(p.V, p.vscale, p.CV, v0 + srow + 32, b, p0, pok, scV1);
        stV2 = load_one(p.V, p.vscale, p.CV, v0 + srow + 64, b, p0, pok, scV2);
        stV3 = load_one(p.V, p.vscale, p.CV, v0 + srow + 96, b, p0, pok, scV3);
    };
    auto scaled = [](float4 a, float sc) { a.x *= sc; a.y *= sc; a.z *= sc; a.w *= sc; return a; };
    auto store_chunk = [&](float* dst) {
        float* d = dst + srow * PITCH + 4 * squad;
        *reinterpret_cast<float4*>(d) = scaled(stU0, scU0);
        *reinterpret_cast<float4*>(d + 32 * PITCH) = scaled(stU1, scU1);
        *reinterpret_cast<float4*>(d + 64 * PITCH) = scaled(stU2, scU2);
        *reinterpret_cast<float4*>(d + 96 * PITCH) = scaled(stU3, scU3);
        *reinterpret_cast<float4*>(d + TILE) = scaled(stV0, scV0);
        *reinterpret_cast<float4*>(d + TILE + 32 * PITCH) = scaled(stV1, scV1);
        *reinterpret_cast<float4*>(d + TILE + 64 * PITCH) = scaled(stV2, scV2);
        *reinterpret_cast<float4*>(d + TILE + 96 * PITCH) = scaled(stV3, scV3);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if (first < last) {
        load_chunk(first);
        store_chunk(smem);
    }
    int buf = 0;
    for (int chunk = first; chunk < last; ++chunk) {
        __syncthreads();                                   // buffer `buf` written; the other one no longer read
        // branch-free staging stream (a branch makes the compiler's vmcnt bookkeeping conservative): after the last chunk
        // it re-stages that chunk into the idle buffer.  The loads are pinned here, their consumers behind the MFMAs.
        load_chunk(chunk + 1 < last ? chunk + 1 : chunk);   // global loads fly under the MFMA block below
        __builtin_amdgcn_sched_barrier(0);
        const float* sU = smem + buf * BUF;
        const float* sV = sU + TILE;
#pragma unroll
        for (int s = 0; s < KC / 16; ++s) {
            const int kofs = 16 * s + 8 * half;
            Split3 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* pa = sU + (wu * 64 + i * 32 + l31) * PITCH + kofs;
                a[i] = split8(*reinterpret_cast<const float4*>(pa), *reinterpret_cast<const float4*>(pa + 4));
                const float* pb = sV + (wv * 64 + i * 32 + l31) * PITCH + kofs;
                b[i] = split8(*reinterpret_cast<const float4*>(pb), *reinterpret_cast<const float4*>(pb + 4));
            }
            // term by term over the four tiles: consecutive MFMAs never accumulate into the same registers
#define SR_TERM(HA, HB)                                      \
    acc[0][0] = mma(a[0].HA, b[0].HB, acc[0][0]);            \
    acc[0][1] = mma(a[0].HA, b[1].HB, acc[0][1]);            \
    acc[1][0] = mma(a[1].HA, b[0].HB, acc[1][0]);            \
    acc[1][1] = mma(a[1].HA, b[1].HB, acc[1][1]);
            SR_TERM(h3, h1) SR_TERM(h1, h3) SR_TERM(h2, h2) SR_TERM(h2, h1) SR_TERM(h1, h2) SR_TERM(h1, h1)
#undef SR_TERM
        }
        __builtin_amdgcn_sched_barrier(0);
        store_chunk(smem + (buf ^ 1) * BUF);
        buf ^= 1;
    }

    // partial[slice][u][v]; C/D layout: column (v) = lane & 31, row (u) = (r & 3) + 8 (r >> 2) + 4 half
    float* dst = p.partial + (int64_t)slice * p.UP * p.VP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = u0 + wu * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int v = v0 + wv * 64 + j * 32 + l31;
                dst[(int64_t)u * p.VP + v] = acc[i][j][r];
            }
}


// ---- stride-2 3x3 weight gradient (the up- / down-sampling convolutions), split-bf16 ---------------------------------
//   D[tap][u][v] = sum_{b, j, i} (us[b,u] U[b,u,2j+ky,2i+kx]) * (vs[b,v] V[b,v,j,i])       tap = 3 ky + kx
// U has 2G + 1 pixels per side (pad-0 stride-2 layers: windows never leave the image), V lives on the G x G grid.
// K = pixels again: V is read as it lies.  Of U, the 8 grid positions i .. i+7 of a lane touch the 17 consecutive
// columns 2i .. 2i+16 of window row 2j + ky, and the three kx variants are the even columns, the odd columns and the
// even columns shifted by one.  The staging writes every window row DE-INTERLEAVED into LDS (even columns | odd
// columns: a different LDS address per lane, free), so that kx = 0 and kx = 1 are plain 32-byte reads whose split
// pieces already are MFMA fragments, and kx = 2 is kx = 0's pieces moved by one bf16 (4 v_perm per piece).
// Tile: workgroup = 64 (u) x 128 (v) channels, 8 waves of 32 x 32 x 9 taps (144 accumulator registers: TWO waves per
// SIMD, so one wave's operand preparation — ~380 VALU operations per 16 k — runs under the other's 54 MFMAs).
// K chunk = 32 grid positions of one grid row: U patch [64][3 rows][33 even | 32 odd] + V [128][32], staged through
// registers, double buffered (152 KB).  v1 of this kernel (4 waves of 32 x 64, one wave per SIMD, interleaved rows +
// 108 v_perm per k-step) ran at 1.15-1.29x the fp32 kernel: the compiler serialises preparation and MFMAs at 490
// registers.
namespace s2 {
constexpr int UT = 64, VT = 128, KC = 32, THREADS = 512, NWAVE = 8;
constexpr int PU = 76;                       // U row pitch: 3 * 76 = 228 = 4 (mod 32) floats between channels
constexpr int ODD = 36;                      // odd columns start here inside a row (evens: 0 .. 32)
constexpr int UCH = 3 * PU;                  // floats per U channel
constexpr int PV = 36;
constexpr int OFF_V = UT * UCH;              // 14 592
constexpr int BUF = OFF_V + VT * PV;         // 19 200 floats
constexpr int LDS_BYTES = 2 * BUF * 4;       // 153 600
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int U_ITEMS = UT * 3 / NWAVE;      // (channel, row) pairs per wave: 24, one 64-column load each
constexpr int V_ITEMS = VT * KC / 4 / THREADS;   // float4 per thread: 2
}  // namespace s2

struct P9 {
    const float* U;
    const float* V;
    const float* uscale;
    const float* vscale;
    float* partial;            // [ks][9][UP][VP]
    int B, CU, CV, UH, UW, GH, GW;
    int segs;                  // chunks per grid row = GW / 32
    int nchunk, per_slice;
    int tiles_u, tiles_v, UP, VP;
};

// (hi half of a, lo half of b): the bf16 pair that straddles two packed dwords
__device__ __forceinline__ unsigned perm_mid(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040302u); }

__device__ __forceinline__ u32x4 shift1(const u32x4 d, unsigned d4) {
    return u32x4{perm_mid(d.x, d.y), perm_mid(d.y, d.z), perm_mid(d.z, d.w), perm_mid(d.w, d4)};
}

__global__ __launch_bounds__(512) void k_wgrad_s2_bf16x3(const P9 p) {
    using namespace s2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    const int tile_uv = bid % (p.tiles_u * p.tiles_v);
    const int slice = bid / (p.tiles_u * p.tiles_v);
    const int u0 = (tile_uv / p.tiles_v) * UT, v0 = (tile_uv % p.tiles_v) * VT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wu = wave >> 2, wv = wave & 3;

    const int first = slice * p.per_slice;
    int last = first + p.per_slice;
    if (last > p.nchunk) last = p.nchunk;
    const int plane_u = p.UH * p.UW, plane_v = p.GH * p.GW;

    // ---- staging (whole channel tiles only: CU % 64 == 0, CV % 128 == 0 — see sr_wgrad_s2_bf16x3_eligible).
    // U: wave w stages channels 8 w .. 8 w + 7, three rows each, 64 columns per load (24 loads of one dword per lane;
    // the per-item address is ONE running VGPR offset — 24 scalar bases per call site would not fit the SGPR file), and
    // thread t < 192 the 65th column of (channel t / 3, row t % 3).  Column c lands at c / 2 (even) or ODD + c / 2.
    // V: thread t stages float4 (row, quad) = ((t + 512 k) / 8, (t + 512 k) % 8).
    float stU[U_ITEMS], stU65 = 0.0f, sc65 = 1.0f, scv[V_ITEMS];
    float4 stV[V_ITEMS];
    const int d_row = p.UW, d_ch = plane_u - 2 * p.UW;
    const int t65 = tid < UT * 3 ? tid : 0;
    const int off65 = (t65 / 3) * plane_u + (t65 % 3) * p.UW + 64;
    const int offv0 = (tid >> 3) * plane_v + 4 * (tid & 7);
    const int lds_col = (lane & 1) ? ODD + (lane >> 1) : (lane >> 1);
    auto load_chunk = [&](int chunk) {
        const int seg = chunk % p.segs;
        const int j = (chunk / p.segs) % p.GH;
        const int b = chunk / (p.segs * p.GH);
        const int i0 = seg * KC;
        const float* ub = p.U + ((int64_t)b * p.CU + u0) * plane_u + (2 * j) * p.UW + 2 * i0;
        const float* usb = p.uscale ? p.uscale + (int64_t)b * p.CU + u0 : nullptr;
        int off = wave * 8 * plane_u + lane;
        // raw loads only: the scales are multiplied on in store_chunk, after the MFMA block (arithmetic here would make
        // the compiler drain vmcnt in front of the MFMAs: the global-load latency of every chunk exposed)
        sc65 = usb ? usb[t65 / 3] : 1.0f;
#pragma unroll
        for (int k = 0; k < U_ITEMS; ++k) {
            stU[k] = ub[off];
            off += (k % 3 == 2) ? d_ch : d_row;
        }
        stU65 = ub[off65];
        const float* vb = p.V + ((int64_t)b * p.CV + v0) * plane_v + j * p.GW + i0;
        const float* vsb = p.vscale ? p.vscale + (int64_t)b * p.CV + v0 : nullptr;
        int offv = offv0;
#pragma unroll
        for (int k = 0; k < V_ITEMS; ++k) scv[k] = 1.0f;
        if (vsb) {
#pragma unroll
            for (int k = 0; k < V_ITEMS; ++k) scv[k] = vsb[(tid >> 3) + 64 * k];
        }
#pragma unroll
        for (int k = 0; k < V_ITEMS; ++k) {
            stV[k] = *reinterpret_cast<const float4*>(vb + offv);
            offv += 64 * plane_v;
        }
    };
    auto store_chunk = [&](float* dst, int chunk) {
        // the wave-uniform U scales are (scalar-)loaded here, not kept in registers across the MFMA block
        const int b = chunk / (p.segs * p.GH);
        float scu[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) scu[c] = 1.0f;
        if (p.uscale) {
            const float* usb = p.uscale + (int64_t)b * p.CU + u0 + wave * 8;
#pragma unroll
            for (int c = 0; c < 8; ++c) scu[c] = usb[c];
        }
        float* du = dst + wave * 8 * UCH + lds_col;
#pragma unroll
        for (int k = 0; k < U_ITEMS; ++k) du[(k / 3) * UCH + (k % 3) * PU] = stU[k] * scu[k / 3];
        if (tid < UT * 3) dst[(t65 / 3) * UCH + (t65 % 3) * PU + 32] = stU65 * sc65;       // column 64 = even #32
        float* dv = dst + OFF_V + (tid >> 3) * PV + 4 * (tid & 7);
#pragma unroll
        for (int k = 0; k < V_ITEMS; ++k) {
            float4 val = stV[k];
            val.x *= scv[k]; val.y *= scv[k]; val.z *= scv[k]; val.w *= scv[k];
            *reinterpret_cast<float4*>(dv + 64 * k * PV) = val;
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    if (first < last) {
        load_chunk(first);
        store_chunk(smem, first);
    }
    int buf = 0;
    for (int chunk = first; chunk < last; ++chunk) {
        __syncthreads();
        const int cn = chunk + 1 < last ? chunk + 1 : chunk;          // branch-free: the last chunk is re-staged
        load_chunk(cn);
        __builtin_amdgcn_sched_barrier(0);
        const float* sU = smem + buf * BUF + (wu * 32 + l31) * UCH;
        const float* sV = smem + buf * BUF + OFF_V + (wv * 32 + l31) * PV;
#pragma unroll 1
        for (int s = 0; s < KC / 16; ++s) {
            const int kofs = 16 * s + 8 * half;
            const float* pb = sV + kofs;
            const Split3 bf = split8(*reinterpret_cast<const float4*>(pb), *reinterpret_cast<const float4*>(pb + 4));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* pe = sU + ky * PU + kofs;                       // even columns e[kofs .. kofs + 8]
                const float* po = pe + ODD;                                  // odd columns  o[kofs .. kofs + 7]
                const Split3 ev = split8(*reinterpret_cast<const float4*>(pe), *reinterpret_cast<const float4*>(pe + 4));
                const Split3 od = split8(*reinterpret_cast<const float4*>(po), *reinterpret_cast<const float4*>(po + 4));
                unsigned e8a, e8b, e8c;
                split2(pe[8], 0.0f, e8a, e8b, e8c);
                const u32x4 s1 = shift1(ev.h1, e8a), s2v = shift1(ev.h2, e8b), s3 = shift1(ev.h3, e8c);
                // the three taps of the row take turns: consecutive MFMAs never accumulate into the same registers (a
                // chain of six dependent MFMAs per tap stalls on the 8-pass result latency: SQ_WAIT_INST_ANY 38 %)
                const int t0 = ky * 3;
                acc[t0] = mma(ev.h3, bf.h1, acc[t0]);
                acc[t0 + 1] = mma(od.h3, bf.h1, acc[t0 + 1]);
                acc[t0 + 2] = mma(s3, bf.h1, acc[t0 + 2]);
                acc[t0] = mma(ev.h1, bf.h3, acc[t0]);
                acc[t0 + 1] = mma(od.h1, bf.h3, acc[t0 + 1]);
                acc[t0 + 2] = mma(s1, bf.h3, acc[t0 + 2]);
                acc[t0] = mma(ev.h2, bf.h2, acc[t0]);
                acc[t0 + 1] = mma(od.h2, bf.h2, acc[t0 + 1]);
                acc[t0 + 2] = mma(s2v, bf.h2, acc[t0 + 2]);
                acc[t0] = mma(ev.h2, bf.h1, acc[t0]);
                acc[t0 + 1] = mma(od.h2, bf.h1, acc[t0 + 1]);
                acc[t0 + 2] = mma(s2v, bf.h1, acc[t0 + 2]);
                acc[t0] = mma(ev.h1, bf.h2, acc[t0]);
                acc[t0 + 1] = mma(od.h1, bf.h2, acc[t0 + 1]);
                acc[t0 + 2] = mma(s1, bf.h2, acc[t0 + 2]);
                acc[t0] = mma(ev.h1, bf.h1, acc[t0]);
                acc[t0 + 1] = mma(od.h1, bf.h1, acc[t0 + 1]);
                acc[t0 + 2] = mma(s1, bf.h1, acc[t0 + 2]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        store_chunk(smem + (buf ^ 1) * BUF, cn);
        buf ^= 1;
    }

    // partial[slice][tap][u][v]
    float* dst = p.partial + (int64_t)slice * 9 * p.UP * p.VP;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = u0 + wu * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int v = v0 + wv * 32 + l31;
            dst[((int64_t)t * p.UP + u) * p.VP + v] = acc[t][r];
        }
}

}  // namespace

// SR_CONV_SPLIT_BF16=1: all four kernel families; or a subset by letter (probes: which family moves a result):
//   w  1x1 weight gradient    g  stride-2 3x3 weight gradient    c  stride-2 3x3 convolution    t  its transposed form
bool sr_wgrad_bf16x3_enabled(char kind) {
    const char* e = std::getenv("SR_CONV_SPLIT_BF16");
    if (!e || !e[0] || e[0] == '0') return false;
    if (e[0] == '1') return true;
    for (; *e; ++e)
        if (*e == kind) return true;
    return false;
}

bool sr_wgrad_bf16x3_eligible(int64_t B, int64_t CU, int64_t CV, int64_t HW, const void* u, const void* v) {
    return B > 0 && CU > 0 && CV > 0 && HW > 0 && HW % 4 == 0 && B * CU * HW < (1LL << 31) && B * CV * HW < (1LL << 31) &&
           ((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
}

static void plan3(int64_t B, int64_t CU, int64_t CV, int64_t HW, P3& p) {
    p.tiles_u = (int)sr_ceil_div(CU, s1::UT);
    p.tiles_v = (int)sr_ceil_div(CV, s1::VT);
    p.UP = p.tiles_u * s1::UT;
    p.VP = p.tiles_v * s1::VT;
    p.cps = (int)sr_ceil_div(HW, s1::KC);
    p.nchunk = (int)(B * p.cps);
    const int tiles_uv = p.tiles_u * p.tiles_v;
    int ks = (2 * SR_NUM_CU + tiles_uv - 1) / tiles_uv;          // two workgroups per CU
    if (ks > p.nchunk) ks = p.nchunk;
    if (ks < 1) ks = 1;
    p.per_slice = (p.nchunk + ks - 1) / ks;
}

int sr_wgrad_bf16x3_slices(int64_t B, int64_t CU, int64_t CV, int64_t HW) {
    P3 p;
    plan3(B, CU, CV, HW, p);
    return (p.nchunk + p.per_slice - 1) / p.per_slice;
}

int64_t sr_wgrad_bf16x3_scratch_floats(int64_t B, int64_t CU, int64_t CV, int64_t HW) {
    P3 p;
    plan3(B, CU, CV, HW, p);
    return (int64_t)sr_wgrad_bf16x3_slices(B, CU, CV, HW) * p.UP * p.VP + 4;
}

// Launches the partial-slab kernel; returns the number of slices and the padded extents for the caller's reduce.
int sr_wgrad_bf16x3_launch(const float* U, const float* V, const float* uscale, const float* vscale, float* partial,
                           int64_t B, int64_t CU, int64_t CV, int64_t HW, int* ks, int* UP, int* VP, hipStream_t st) {
    P3 p;
    plan3(B, CU, CV, HW, p);
    p.U = U; p.V = V; p.uscale = uscale; p.vscale = vscale; p.partial = partial;
    p.B = (int)B; p.CU = (int)CU; p.CV = (int)CV; p.HW = (int)HW;
    *ks = (p.nchunk + p.per_slice - 1) / p.per_slice;
    *UP = p.UP;
    *VP = p.VP;
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad1_bf16x3), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  s1::LDS_BYTES);
        configured = true;
    }
    hipLaunchKernelGGL(k_wgrad1_bf16x3, dim3((unsigned)(p.tiles_u * p.tiles_v * *ks)), dim3(s1::THREADS), s1::LDS_BYTES, st, p);
    return sr_launch_status();
}

// ---- stride-2 3x3 -----------------------------------------------------------------------------------------------
bool sr_wgrad_s2_bf16x3_eligible(int64_t B, int64_t CU, int64_t CV, int64_t UH, int64_t UW, int64_t GH, int64_t GW,
                                 const void* v) {
    return B > 0 && GW % s2::KC == 0 && CU % s2::UT == 0 && CV % s2::VT == 0 && UH == 2 * GH + 1 && UW == 2 * GW + 1 &&
           B * CU * UH * UW < (1LL << 31) && B * CV * GH * GW < (1LL << 31) && (reinterpret_cast<uintptr_t>(v) & 15) == 0;
}

static void plan9(int64_t B, int64_t CU, int64_t CV, int64_t GH, int64_t GW, P9& p) {
    p.tiles_u = (int)sr_ceil_div(CU, s2::UT);
    p.tiles_v = (int)sr_ceil_div(CV, s2::VT);
    p.UP = p.tiles_u * s2::UT;
    p.VP = p.tiles_v * s2::VT;
    p.segs = (int)(GW / s2::KC);
    p.nchunk = (int)(B * GH * p.segs);
    const int tiles_uv = p.tiles_u * p.tiles_v;
    int ks = (SR_NUM_CU + tiles_uv - 1) / tiles_uv;              // one workgroup per CU
    if (ks > p.nchunk) ks = p.nchunk;
    if (ks < 1) ks = 1;
    p.per_slice = (p.nchunk + ks - 1) / ks;
}

int64_t sr_wgrad_s2_bf16x3_scratch_floats(int64_t B, int64_t CU, int64_t CV, int64_t GH, int64_t GW) {
    if (GW % s2::KC != 0 || CU % s2::UT != 0 || CV % s2::VT != 0) return 0;
    P9 p;
    plan9(B, CU, CV, GH, GW, p);
    const int ks = (p.nchunk + p.per_slice - 1) / p.per_slice;
    return (int64_t)ks * 9 * p.UP * p.VP + 4;
}

int sr_wgrad_s2_bf16x3_launch(const float* U, const float* V, const float* uscale, const float* vscale, float* partial,
                              int64_t B, int64_t CU, int64_t CV, int64_t UH, int64_t UW, int64_t GH, int64_t GW, int* ks,
                              int* UP, int* VP, hipStream_t st) {
    P9 p;
    plan9(B, CU, CV, GH, GW, p);
    p.U = U; p.V = V; p.uscale = uscale; p.vscale = vscale; p.partial = partial;
    p.B = (int)B; p.CU = (int)CU; p.CV = (int)CV; p.UH = (int)UH; p.UW = (int)UW; p.GH = (int)GH; p.GW = (int)GW;
    *ks = (p.nchunk + p.per_slice - 1) / p.per_slice;
    *UP = p.UP;
    *VP = p.VP;
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_s2_bf16x3),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, s2::LDS_BYTES);
        configured = true;
    }
    hipLaunchKernelGGL(k_wgrad_s2_bf16x3, dim3((unsigned)(p.tiles_u * p.tiles_v * *ks)), dim3(s2::THREADS), s2::LDS_BYTES,
                       st, p);
    return sr_launch_status();
}
