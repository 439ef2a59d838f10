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
// K loop in chunks of 8 tiles (a 16x2 pixel strip):
//   * the strip's x halo (64 channels x 4 rows x 24 columns, 16-byte aligned window) and gy pixels
//     (64 channels x 2 rows x 16 columns) arrive by LDS-DMA, four chunks deep; every channel occupies
//     25 (resp. 9) float4 slots = 24 (8) payload + 1 hole, so that the 32 lanes of a half-wave, which
//     own 32 different channels, hit different bank groups with ds_read_b128 / ds_read_b64;
//   * BOTH transforms run in registers, directly into the MFMA operands: lane (channel = l & 31,
//     k = l >> 5) computes B^T d B of its own (channel, tile) — 8 ds_read_b128 + 32 VALU ops per
//     k-step (packed fp32 where the data allows) — and A dY A^T of its own (output channel, tile) —
//     2 ds_read_b64 + 12 VALU ops.  No transformed operand ever touches LDS.  The transforms of k-step
//     s+1 (and the chunk's 10 DMA instructions) are pinned into the four 4-MFMA slots of k-step s.
//   * k-step s pairs tiles (t, t + 2) on the two k-lanes so both halves read with the same
//     compile-time element pattern (tile columns 3 + 2t .. 6 + 2t of the aligned window).
#include "common.h"
#include "conv_wino.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

__device__ __attribute__((aligned(16))) const float g_wgw_zero[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int XS = 25;                      // float4 slots per channel of the x strip (4 rows x 6 + hole)
constexpr int GS = 9;                       // float4 slots per channel of the gy strip (2 rows x 4 + hole)
constexpr int X_INSTR = 64 * XS / 64;       // 25 wave-instructions
constexpr int G_INSTR = 64 * GS / 64;       // 9
constexpr int X_FLOATS = X_INSTR * 256;     // 6400
constexpr int G_FLOATS = G_INSTR * 256;     // 2304
constexpr int BUF = X_FLOATS + G_FLOATS;    // 8704 floats = 34 KB per chunk
constexpr int NBUF = 4;
constexpr int PAD = 4 * 256;                // landing zone of the surplus DMA instructions (never read)
constexpr int X_PER_WAVE = (X_INSTR + 3) / 4;   // 7
constexpr int G_PER_WAVE = (G_INSTR + 3) / 4;   // 3
constexpr int MAX_B = 32;                   // scale tables [B][64] x 2 in LDS

struct WgWinoParams {
    const float* x;
    const float* gy;
    const float* xscale;
    const float* gscale;
    float* partial;          // [slices][16][C][N]
    int B, C, N, H, W;
    int tiles_c, tiles_n, slices, chunks_per_slice, chunks_total;
    int cty, ctx;            // chunk grid per sample: H/2 x W/16
};

__device__ __forceinline__ void bt_row4(const float (&d)[16], float (&o)[16], int q, float s) {
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        t[j] = q == 0 ? d[j] - d[8 + j] : q == 1 ? d[4 + j] + d[8 + j] : q == 2 ? d[8 + j] - d[4 + j]
                                                                      : d[4 + j] - d[12 + j];
    o[4 * q + 0] = (t[0] - t[2]) * s;
    o[4 * q + 1] = (t[1] + t[2]) * s;
    o[4 * q + 2] = (t[2] - t[1]) * s;
    o[4 * q + 3] = (t[1] - t[3]) * s;
}

__global__ __launch_bounds__(256) void k_wgrad_wino(const WgWinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const pad = smem + NBUF * BUF;
    float* const tabx = pad + PAD;                    // [B][64] style of this channel block
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

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wc = wave >> 1, wn = wave & 1;

    for (int i = tid; i < p.B * 64; i += 256) {
        const int b = i >> 6, ch = i & 63;
        tabx[i] = p.xscale ? p.xscale[(int64_t)b * p.C + c0 + ch] : 1.0f;
        tabg[i] = p.gscale ? p.gscale[(int64_t)b * p.N + n0 + ch] : 1.0f;
    }

    // ---- DMA descriptors.  Slot s = 64 j + lane of the chunk image; every wave issues the same number
    // of instructions (the surplus ones land in a pad region nobody reads).  Holes and pad slots may
    // hold anything, only out-of-image halo elements need the zero line: flag bits 1 top row, 2 bottom
    // row, 4 left column group, 8 right column group are matched against the strip's position.
    int xd_off[X_PER_WAVE], xd_flag[X_PER_WAVE];
#pragma unroll
    for (int i = 0; i < X_PER_WAVE; ++i) {
        const int s = 64 * (wave + 4 * i) + lane;
        const int c = s / XS, rem = s % XS;
        const int r = rem / 6, q = rem % 6;
        const bool payload = wave + 4 * i < X_INSTR && rem < XS - 1;
        xd_off[i] = payload ? (c * p.H + (r - 1)) * p.W + 4 * q - 4 : 0;
        xd_flag[i] = payload ? (r == 0 ? 1 : 0) | (r == 3 ? 2 : 0) | (q == 0 ? 4 : 0) | (q == 5 ? 8 : 0) : 0;
    }
    int gd_off[G_PER_WAVE];
#pragma unroll
    for (int i = 0; i < G_PER_WAVE; ++i) {
        const int s = 64 * (wave + 4 * i) + lane;
        const int n = s / GS, rem = s % GS;
        gd_off[i] = (wave + 4 * i < G_INSTR && rem < GS - 1) ? (n * p.H + rem / 4) * p.W + 4 * (rem % 4) : 0;
    }
    const int64_t plane = (int64_t)p.H * p.W;

    // strip coordinates of the chunk the next fetch brings in (advanced incrementally, clamped at the last
    // chunk of the tensor so that a surplus fetch at the end of a slice stays in bounds)
    int d_txc = k_beg % p.ctx, d_ty = (k_beg / p.ctx) % p.cty, d_b = k_beg / (p.ctx * p.cty);
    const float* d_xo;
    const float* d_go;
    int d_edge;
    float* d_dst;
    auto dma_begin = [&](int buf) {          // uniform part of one chunk fetch
        const int y0 = 2 * d_ty, x0 = 16 * d_txc;
        d_xo = p.x + ((int64_t)d_b * p.C + c0) * plane + (int64_t)y0 * p.W + x0;
        d_go = p.gy + ((int64_t)d_b * p.N + n0) * plane + (int64_t)y0 * p.W + x0;
        d_edge = (d_ty == 0 ? 1 : 0) | (d_ty == p.cty - 1 ? 2 : 0) | (d_txc == 0 ? 4 : 0) | (d_txc == p.ctx - 1 ? 8 : 0);
        d_dst = smem + buf * BUF;
        const bool last = d_txc == p.ctx - 1 && d_ty == p.cty - 1 && d_b == p.B - 1;
        if (!last && ++d_txc == p.ctx) {
            d_txc = 0;
            if (++d_ty == p.cty) { d_ty = 0; ++d_b; }
        }
    };
    auto dma_x1 = [&](int i) {
        const float* src = (xd_flag[i] & d_edge) ? g_wgw_zero : d_xo + xd_off[i];
        __builtin_amdgcn_global_load_lds(
            (gptr_t)src, (lptr_t)(wave + 4 * i < X_INSTR ? d_dst + (wave + 4 * i) * 256 : pad + wave * 256), 16, 0, 0);
    };
    auto dma_g1 = [&](int i) {
        __builtin_amdgcn_global_load_lds(
            (gptr_t)(d_go + gd_off[i]),
            (lptr_t)(wave + 4 * i < G_INSTR ? d_dst + X_FLOATS + (wave + 4 * i) * 256 : pad + wave * 256), 16, 0, 0);
    };
    auto dma_next = [&](int buf) {
        dma_begin(buf);
#pragma unroll
        for (int i = 0; i < X_PER_WAVE; ++i) dma_x1(i);
#pragma unroll
        for (int i = 0; i < G_PER_WAVE; ++i) dma_g1(i);
    };

    f32x16 acc[16];
#pragma unroll
    for (int pos = 0; pos < 16; ++pos)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pos][r] = 0.0f;

    // per-lane read bases inside a chunk buffer (floats); k-lane 1 reads one float4 slot further right
    const int xb = (wc * 32 + l31) * (XS * 4) + half * 4;
    const int gb = X_FLOATS + (wn * 32 + l31) * (GS * 4) + half * 4;
    const int tab_c = wc * 32 + l31, tab_n = wn * 32 + l31;

    // k-step ks of a chunk pairs tiles t = (ks & 1) + 4 (ks >> 1) (k-lane 0) and t + 2 (k-lane 1): columns
    // 3 + 2t .. 6 + 2t of the aligned window = elements 3..6 (even ks) / 1..4 (odd ks) of two float4 slots.
    // The raw operands are fetched with explicit ds_read_b128 / ds_read_b64 (inline asm: left to itself
    // the compiler fetches only the 4 live elements per row as ds_read2_b32, which lands the 32 lanes of a
    // half-wave — 32 channels, 100 floats apart — on 8 banks; whole 16-byte slots at 25-slot pitch are
    // conflict free).  lds_wait() carries the registers as operands, so every use is ordered behind it.
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    struct Raw { f4 a[4], b[4]; f2 g0, g1; };
    auto load_raw = [&](unsigned xaddr, unsigned gaddr, int ks, Raw& R) {
        const int xo = ((ks >> 1) * 2 + (ks & 1)) * 16, go = ((ks & 1) + 4 * (ks >> 1)) * 8;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(R.a[r]) : "v"(xaddr), "i"(r * 96 + xo));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(R.b[r]) : "v"(xaddr), "i"(r * 96 + xo + 16));
        }
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(R.g0) : "v"(gaddr), "i"(go));
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(R.g1) : "v"(gaddr), "i"(go + 64));
    };
    auto lds_wait = [](Raw& R) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(R.a[0]), "+v"(R.a[1]), "+v"(R.a[2]), "+v"(R.a[3]), "+v"(R.b[0]), "+v"(R.b[1]), "+v"(R.b[2]),
                       "+v"(R.b[3]), "+v"(R.g0), "+v"(R.g1));
    };
    // Transforms with packed fp32 ops where the data allows (a non-MFMA instruction costs the matrix pipe an
    // issue slot).  B^T d B, row q: t0..t3 scalar (the two operands of each come from
    // different 16-byte loads), then with P = (t1, t2), Q = (t0, t3):
    //   (o1, o2) = (P.lo + P.hi, P.hi - P.lo)        (o0, -o3) = (Q.lo - P.hi, Q.hi - P.lo)
    // one v_pk_add_f32 each (op_sel / neg modifiers), one v_pk_mul_f32 each for the style (the second with
    // neg_hi to undo the sign).  A dY A^T: both stages packed on the (p, q) / (r, s) pairs as loaded.
    auto pk_mul = [](f2 a, f2 b) { f2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    auto pk_mul_pn = [](f2 a, f2 b) {            // (a.lo * b.lo, -(a.hi * b.hi))
        f2 r;
        asm("v_pk_mul_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        return r;
    };
    auto pk_add = [](f2 a, f2 b) { f2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    auto pk_sub = [](f2 a, f2 b) {
        f2 r;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        return r;
    };
    auto pk_sum_diff = [](f2 w) {                // (w.lo + w.hi, w.lo - w.hi)
        f2 r;
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(w));
        return r;
    };
    auto pk_nsum_ndiff = [](f2 w) {              // (-w.lo - w.hi, -w.lo + w.hi)
        f2 r;
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[1,1] neg_hi:[1,0]" : "=v"(r) : "v"(w));
        return r;
    };
    auto pk_o12 = [](f2 P) {                     // (P.lo + P.hi, P.hi - P.lo)
        f2 r;
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(P));
        return r;
    };
    auto pk_o03 = [](f2 Q, f2 P) {               // (Q.lo - P.hi, Q.hi - P.lo)
        f2 r;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(Q), "v"(P));
        return r;
    };
    struct ZTmp { f2 w0, w1, w2, w3; };
    auto unpack_d = [](const Raw& R, int ks, float (&d)[16]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if ((ks & 1) == 0) { d[4 * r] = R.a[r].w; d[4 * r + 1] = R.b[r].x; d[4 * r + 2] = R.b[r].y; d[4 * r + 3] = R.b[r].z; }
            else { d[4 * r] = R.a[r].y; d[4 * r + 1] = R.a[r].z; d[4 * r + 2] = R.a[r].w; d[4 * r + 3] = R.b[r].x; }
        }
    };
    auto v_row = [&](const float (&d)[16], int q, f2 s2, float (&V)[16]) {
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            t[j] = q == 0 ? d[j] - d[8 + j] : q == 1 ? d[4 + j] + d[8 + j] : q == 2 ? d[8 + j] - d[4 + j]
                                                                          : d[4 + j] - d[12 + j];
        f2 P, Q;
        P.x = t[1]; P.y = t[2];
        Q.x = t[0]; Q.y = t[3];
        const f2 a = pk_mul(pk_o12(P), s2);          // (o1, o2) * s
        const f2 b = pk_mul_pn(pk_o03(Q, P), s2);    // (o0, o3) * s
        V[4 * q + 0] = b.x;
        V[4 * q + 1] = a.x;
        V[4 * q + 2] = a.y;
        V[4 * q + 3] = b.y;
    };
    auto z_begin = [&](const Raw& R, f2 g2, ZTmp& T) {
        T.w0 = pk_mul(R.g0, g2);                     // (p, q) * sg
        T.w3 = pk_mul(R.g1, g2);                     // (r, s) * sg   (row 3 is its negative)
        T.w1 = pk_add(T.w0, T.w3);
        T.w2 = pk_sub(T.w0, T.w3);
    };
    auto z_row = [&](f2 w, int i, float (&Z)[16]) {  // rows 0..2: (a, a + b, a - b, -b)
        const f2 sd = pk_sum_diff(w);
        Z[4 * i + 0] = w.x;
        Z[4 * i + 1] = sd.x;
        Z[4 * i + 2] = sd.y;
        Z[4 * i + 3] = -w.y;
    };
    auto z_row3 = [&](f2 w, float (&Z)[16]) {        // row 3 from (r, s): (-r, -r - s, -r + s, s)
        const f2 nd = pk_nsum_ndiff(w);
        Z[12] = -w.x;
        Z[13] = nd.x;
        Z[14] = nd.y;
        Z[15] = w.y;
    };
    // piece i (0..3) of the transforms of one k-step: pinned into the four MFMA slots of the previous one
    auto transform_piece = [&](const float (&d)[16], const Raw& R, ZTmp& T, int i, f2 s2, f2 g2, float (&V)[16],
                               float (&Z)[16]) {
        v_row(d, i, s2, V);
        if (i == 0) z_begin(R, g2, T);
        if (i == 1) { z_row(T.w0, 0, Z); z_row(T.w1, 1, Z); }
        if (i == 2) { z_row(T.w2, 2, Z); z_row3(T.w3, Z); }
    };
    auto transform = [&](const Raw& R, int ks, float sx, float sg, float (&V)[16], float (&Z)[16]) {
        float d[16];
        unpack_d(R, ks, d);
        f2 s2, g2;
        s2.x = s2.y = sx;
        g2.x = g2.y = sg;
        ZTmp T;
#pragma unroll
        for (int i = 0; i < 4; ++i) transform_piece(d, R, T, i, s2, g2, V, Z);
    };

    // ---- pipeline.  Four chunk buffers; at the top of iteration kk chunks kk and kk+1 are complete in
    // LDS (every wave drained its own DMAs of them before the barrier), chunk kk+2 is in flight and chunk
    // kk+3 is issued into the buffer of chunk kk-1.  The operands of k-step 0 of chunk kk+1 are produced
    // under the MFMAs of the last k-step of chunk kk, so the matrix pipe never waits for a transform.
    // Raw s_barrier, not __syncthreads: the release fence of the latter drains vmcnt to 0 and with it the
    // chunks that are meant to stay in flight.  Cross-wave data is DMA-written only: "my DMAs landed"
    // (vmcnt) + "my LDS reads retired" (lgkmcnt) before the barrier is the whole protocol; the asm memory
    // clobbers keep the compiler from moving LDS accesses across it.
    const unsigned lds0 = (unsigned)(__SIZE_TYPE__)(lptr_t)smem;       // LDS byte address of the buffers
    const int per_sample = p.ctx * p.cty;
    int c_left = per_sample - (k_beg % per_sample), c_b = k_beg / per_sample;   // sample of the chunk being fetched
    auto scales_next = [&](float& sx, float& sg) {       // scales of the next chunk whose operands are built
        const int b = min(c_b, p.B - 1);
        sx = tabx[b * 64 + tab_c];
        sg = tabg[b * 64 + tab_n];
        if (--c_left == 0) { c_left = per_sample; ++c_b; }
    };
    float V[2][16], Z[2][16];
    Raw R;
    float sx, sg;
    dma_next(0);
    dma_next(1);
    dma_next(2);
    asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    scales_next(sx, sg);
#ifdef WGW_NO_XFORM
#pragma unroll
    for (int i = 0; i < 16; ++i) { V[0][i] = V[1][i] = sx + i; Z[0][i] = Z[1][i] = sg - i; }
#else
    load_raw(lds0 + (unsigned)xb * 4u, lds0 + (unsigned)gb * 4u, 0, R);
    lds_wait(R);
    transform(R, 0, sx, sg, V[0], Z[0]);
#endif
    int cur = 0;
    for (int kk = k_beg; kk < k_end; ++kk) {
        asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // chunk kk+3 (a surplus fetch past the slice lands in a free buffer): its 10 DMA instructions are
        // spread over the four k-steps below, under the MFMAs; no branch in the loop body
        dma_begin((cur + 3) & 3);
        const unsigned xaddr = lds0 + (unsigned)(cur * BUF + xb) * 4u, gaddr = lds0 + (unsigned)(cur * BUF + gb) * 4u;
        const int nxt = (cur + 1) & 3;
        const unsigned xaddr_n = lds0 + (unsigned)(nxt * BUF + xb) * 4u, gaddr_n = lds0 + (unsigned)(nxt * BUF + gb) * 4u;
        float sxn, sgn;
        scales_next(sxn, sgn);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#ifndef WGW_NO_XFORM
            // LDS reads of the next k-step (of the next chunk after the last one) in flight under the MFMAs
            if (ks + 1 < 4) load_raw(xaddr, gaddr, ks + 1, R);
            else load_raw(xaddr_n, gaddr_n, 0, R);
#endif
            __builtin_amdgcn_sched_barrier(0);
            // two MFMAs cover the LDS latency of the reads above, then the wait; after that four slots of
            // MFMAs, each with one piece of the next k-step's transforms (and its share of the DMA issue)
            // pinned into it
#pragma unroll
            for (int pos = 0; pos < 2; ++pos)
                acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[ks & 1][pos], Z[ks & 1][pos], acc[pos], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifndef WGW_NO_XFORM
            lds_wait(R);
            float dn[16];
            unpack_d(R, ks + 1 < 4 ? ks + 1 : 0, dn);
            f2 s2, g2;
            s2.x = s2.y = (ks + 1 < 4) ? sx : sxn;
            g2.x = g2.y = (ks + 1 < 4) ? sg : sgn;
            ZTmp T;
#endif
#pragma unroll
            for (int slot = 0; slot < 4; ++slot) {
#pragma unroll
                for (int pos = (slot == 0 ? 2 : 4 * slot); pos < 4 * slot + 4; ++pos)
                    acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[ks & 1][pos], Z[ks & 1][pos], acc[pos], 0, 0, 0);
#ifndef WGW_NO_XFORM
                transform_piece(dn, R, T, slot, s2, g2, V[(ks + 1) & 1], Z[(ks + 1) & 1]);
#endif
#ifndef WGW_NO_DMA
                if (slot > 0) {
                    const int i = 3 * ks + slot - 1;             // 10 DMA instructions over ks 0..3, slots 1..3
                    if (i < X_PER_WAVE) dma_x1(i);
                    else if (i < X_PER_WAVE + G_PER_WAVE) dma_g1(i - X_PER_WAVE);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        sx = sxn;
        sg = sgn;
        cur = nxt;
    }
    // surplus fetches are still landing in this workgroup's LDS: drain them before the wave can retire
    __builtin_amdgcn_s_waitcnt(0x0F70);

    // ---- partial dU slab of this slice: rows = channels (r & 3) + 8 (r >> 2) + 4 half, cols = l31
    float* out = p.partial + (int64_t)slice * 16 * p.C * p.N;
#pragma unroll
    for (int pos = 0; pos < 16; ++pos)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            out[((int64_t)pos * p.C + c) * p.N + n0 + wn * 32 + l31] = acc[pos][r];
        }
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

bool sr_wgrad_wino_eligible(int64_t B, int64_t C, int64_t N, int64_t H, int64_t W, const void* x, const void* gy) {
    if (B <= 0 || B > MAX_B || C % 64 != 0 || N % 64 != 0 || H % 2 != 0 || W % 16 != 0) return false;
    if (B * C * H * W >= (1LL << 31) || B * N * H * W >= (1LL << 31)) return false;
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
    const int lds = (NBUF * BUF + PAD + 2 * MAX_B * 64) * 4;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_wino), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return SR_EINVAL;
        }
        configured = true;
    }
    const int64_t blocks = (int64_t)p.tiles_c * p.tiles_n * p.slices;
    hipLaunchKernelGGL(k_wgrad_wino, dim3((unsigned)blocks), dim3(256), lds, st, p);
    const int64_t cn = C * N;
    hipLaunchKernelGGL(k_wgrad_wino_finish, dim3((unsigned)sr_ceil_div(cn, 64)), dim3(256), 0, st, dwt, scratch,
                       p.slices, (int)C, (int)N);
    return sr_launch_status();
}
