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
// GEMM per tap:  D[u][v] = sum_k A[u][k] * Bm[k][v],  k = pixel.  A workgroup owns a (UT x VT)
// tile of channels for ALL taps of the window (each staged U halo patch feeds every tap), a wave
// owns 32 x 32 x taps = 9 accumulator tiles (144 VGPRs).  K is split over workgroups
// (grid = channel tiles x K slices ~ 2 x 256 CUs); slices write fp32 partial slabs that a second
// kernel sums in a fixed order (deterministic; no float atomics) while applying the output layout.
// LDS: channel-major planes with ODD pitch, so the 32 lanes of an operand fetch (32 channels, same
// pixel) hit 32 different banks.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

template <int IS, int TY, int TX, int PW, int PH, int PB, int UT, int VT>
struct WG {
    static constexpr int NT = TY * TX;
    static constexpr int EH = (PH - 1) * IS + TY;
    static constexpr int EW = (PW - 1) * IS + TX;
    static constexpr int UPL0 = PB * EH * EW;
    static constexpr int UPL = UPL0 + ((UPL0 % 2 == 0) ? 1 : 0);   // odd plane pitch
    static constexpr int NPIX = PB * PH * PW;
    static constexpr int VPL = NPIX + 1;                           // NPIX is even -> odd pitch
    static constexpr int U_ELEMS = UT * UPL0, V_ELEMS = VT * NPIX;
    static constexpr int U_ITERS = (U_ELEMS + 255) / 256, V_ITERS = (V_ELEMS + 255) / 256;
    static constexpr int WU = UT / 32, WV = VT / 32;
};

template <int IS, int TY, int TX, int PW, int PH, int PB, int UT, int VT>
__global__ __launch_bounds__(256, 1) void k_wgrad_mfma(const WgradParams p) {
    using G = WG<IS, TY, TX, PW, PH, PB, UT, VT>;
    static_assert(G::WU * G::WV == 4, "4 waves per workgroup");
    static_assert(PW % 2 == 0, "pixel pairs run along x");
    __shared__ float s_u[UT * G::UPL];
    __shared__ float s_v[VT * G::VPL];

    int bid = blockIdx.x;
    const int tile_uv = bid % (p.tiles_u * p.tiles_v);
    const int slice = bid / (p.tiles_u * p.tiles_v);
    const int u0 = (tile_uv / p.tiles_v) * UT, v0 = (tile_uv % p.tiles_v) * VT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wu = wave / G::WV, wv = wave % G::WV;
    const int a_base = (wu * 32 + l31) * G::UPL + half * IS;
    const int b_base = (wv * 32 + l31) * G::VPL + half;

    const int npatch = p.tiles_x * p.tiles_y * p.tiles_b;
    const int first = slice * p.patches_per_slice;
    int last = first + p.patches_per_slice;
    if (last > npatch) last = npatch;

    const int64_t plane_u = (int64_t)p.UH * p.UW, plane_v = (int64_t)p.GH * p.GW;

    float u_reg[G::U_ITERS], v_reg[G::V_ITERS];

    auto fetch = [&](int patch) {
        const int tx_i = patch % p.tiles_x;
        const int ty_i = (patch / p.tiles_x) % p.tiles_y;
        const int tb_i = patch / (p.tiles_x * p.tiles_y);
        const int gy0 = ty_i * PH, gx0 = tx_i * PW, b0 = tb_i * PB;
        const int iy0 = gy0 * IS + p.dy0, ix0 = gx0 * IS + p.dx0;
#pragma unroll
        for (int it = 0; it < G::U_ITERS; ++it) {
            const int e = tid + it * 256;
            const int col = e % G::EW, r = (e / G::EW) % G::EH, pb = (e / (G::EW * G::EH)) % PB;
            const int u = u0 + e / G::UPL0;
            const int gy = iy0 + r, gx = ix0 + col, b = b0 + pb;
            float val = 0.0f;
            if (e < G::U_ELEMS && u < p.CU && b < p.B && gy >= 0 && gy < p.UH && gx >= 0 && gx < p.UW) {
                val = p.U[((int64_t)b * p.CU + u) * plane_u + (int64_t)gy * p.UW + gx];
                if (p.uscale) val *= p.uscale[(int64_t)b * p.CU + u];
            }
            u_reg[it] = val;
        }
#pragma unroll
        for (int it = 0; it < G::V_ITERS; ++it) {
            const int e = tid + it * 256;
            const int px = e % PW, py = (e / PW) % PH, pb = (e / (PW * PH)) % PB;
            const int v = v0 + e / G::NPIX;
            const int gy = gy0 + py, gx = gx0 + px, b = b0 + pb;
            float val = 0.0f;
            if (e < G::V_ELEMS && v < p.CV && b < p.B && gy < p.GH && gx < p.GW) {
                val = p.V[((int64_t)b * p.CV + v) * plane_v + (int64_t)gy * p.GW + gx];
                if (p.vscale) val *= p.vscale[(int64_t)b * p.CV + v];
            }
            v_reg[it] = val;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < G::U_ITERS; ++it) {
            const int e = tid + it * 256;
            if (e < G::U_ELEMS) s_u[(e / G::UPL0) * G::UPL + e % G::UPL0] = u_reg[it];
        }
#pragma unroll
        for (int it = 0; it < G::V_ITERS; ++it) {
            const int e = tid + it * 256;
            if (e < G::V_ELEMS) s_v[(e / G::NPIX) * G::VPL + e % G::NPIX] = v_reg[it];
        }
    };

    f32x16 acc[G::NT];
#pragma unroll
    for (int t = 0; t < G::NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    if (first < last) fetch(first);
    for (int patch = first; patch < last; ++patch) {
        __syncthreads();
        commit();
        __syncthreads();
        if (patch + 1 < last) fetch(patch + 1);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int py = 0; py < PH; ++py)
#pragma unroll
                for (int qx = 0; qx < PW / 2; ++qx) {
                    const float bv = s_v[b_base + (pb * PH + py) * PW + 2 * qx];
                    const int ua = a_base + (pb * G::EH + py * IS) * G::EW + 2 * qx * IS;
#pragma unroll
                    for (int ty = 0; ty < TY; ++ty)
#pragma unroll
                        for (int tx = 0; tx < TX; ++tx) {
                            const float av = s_u[ua + ty * G::EW + tx];
                            acc[ty * TX + tx] =
                                __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ty * TX + tx], 0, 0, 0);
                        }
                }
    }

    // partial[slice][tap][u][v]; C/D layout: column (v) = lane & 31, row (u) = (r&3) + 8*(r>>2) + 4*half
    float* dst = p.partial + (int64_t)slice * G::NT * p.UP * p.VP;
#pragma unroll
    for (int t = 0; t < G::NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = u0 + wu * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int v = v0 + wv * 32 + l31;
            dst[((int64_t)t * p.UP + u) * p.VP + v] = acc[t][r];
        }
}

// out[wslab(t) * CU*CV-sized slab ... ] — sums the K slices in a fixed order and writes the
// requested layout: element (t, u, v) goes to out[tmap[t] * slab + u * su + v * sv].
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
        for (int s = 0; s < p.ks; ++s) acc += src[s * stride_s];
        p.out[p.tmap[t] * p.slab + u * p.su + v * p.sv] = acc;
    }
}

struct Plan {
    int pw, ph, pb, ut, vt;
    int tiles_x, tiles_y, tiles_b, tiles_u, tiles_v, ks, pps;
};

Plan make_plan(int is, int B, int CU, int CV, int GH, int GW) {
    Plan pl;
    if (GW > 16) { pl.pw = 32; pl.ph = 2; pl.pb = 1; }
    else if (GW > 8) { pl.pw = 16; pl.ph = 4; pl.pb = 1; }
    else if (GW > 4) { pl.pw = 8; pl.ph = 8; pl.pb = 1; }
    else { pl.pw = 4; pl.ph = 4; pl.pb = 4; }
    if (is == 1) { pl.ut = 64; pl.vt = 64; }
    else { pl.ut = 32; pl.vt = 128; }
    pl.tiles_x = (GW + pl.pw - 1) / pl.pw;
    pl.tiles_y = (GH + pl.ph - 1) / pl.ph;
    pl.tiles_b = (B + pl.pb - 1) / pl.pb;
    pl.tiles_u = (CU + pl.ut - 1) / pl.ut;
    pl.tiles_v = (CV + pl.vt - 1) / pl.vt;
    const int npatch = pl.tiles_x * pl.tiles_y * pl.tiles_b;
    const int tiles_uv = pl.tiles_u * pl.tiles_v;
    int ks = (2 * SR_NUM_CU + tiles_uv - 1) / tiles_uv;
    if (ks > npatch) ks = npatch;
    if (ks < 1) ks = 1;
    pl.pps = (npatch + ks - 1) / ks;
    pl.ks = (npatch + pl.pps - 1) / pl.pps;
    return pl;
}

template <int IS, int TY, int TX, int UT, int VT>
int launch_wgrad(WgradParams& p, const Plan& pl, hipStream_t st) {
    const dim3 grid((unsigned)(pl.tiles_u * pl.tiles_v * pl.ks)), block(256);
    if (pl.pw == 32) hipLaunchKernelGGL((k_wgrad_mfma<IS, TY, TX, 32, 2, 1, UT, VT>), grid, block, 0, st, p);
    else if (pl.pw == 16) hipLaunchKernelGGL((k_wgrad_mfma<IS, TY, TX, 16, 4, 1, UT, VT>), grid, block, 0, st, p);
    else if (pl.pw == 8) hipLaunchKernelGGL((k_wgrad_mfma<IS, TY, TX, 8, 8, 1, UT, VT>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((k_wgrad_mfma<IS, TY, TX, 4, 4, 4, UT, VT>), grid, block, 0, st, p);
    return sr_launch_status();
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

extern "C" int64_t sr_conv2d_wgrad_scratch_floats(int64_t B, int64_t C, int64_t N, int64_t IH,
                                                  int64_t IW, int64_t OH, int64_t OW, int ksize,
                                                  int stride, int pad, int transposed) {
    int is, GH, GW, UH, UW, CUc, CVc, d0;
    if (!geometry(B, C, N, IH, IW, OH, OW, ksize, stride, pad, transposed, is, GH, GW, UH, UW, CUc, CVc, d0))
        return -1;
    const Plan pl = make_plan(is, (int)B, CUc, CVc, GH, GW);
    return (int64_t)pl.ks * ksize * ksize * (pl.tiles_u * pl.ut) * (int64_t)(pl.tiles_v * pl.vt) + 4;
}

extern "C" int sr_conv2d_wgrad_mfma(float* dwt, const float* x, const float* gy, const float* xscale,
                                    const float* gscale, int64_t B, int64_t C, int64_t N, int64_t IH,
                                    int64_t IW, int64_t OH, int64_t OW, int ksize, int stride, int pad,
                                    int transposed, float* scratch, sr_stream_t stream) {
    int is, GH, GW, UH, UW, CUc, CVc, d0;
    if (!geometry(B, C, N, IH, IW, OH, OW, ksize, stride, pad, transposed, is, GH, GW, UH, UW, CUc, CVc, d0))
        return SR_EINVAL;
    if (!dwt || !x || !gy || !scratch) return SR_EINVAL;
    hipStream_t st = sr_stream(stream);
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
        if (ksize == 3 && is == 1) rc = launch_wgrad<1, 3, 3, 64, 64>(p, pl, st);
        else if (ksize == 3 && is == 2) rc = launch_wgrad<2, 3, 3, 32, 128>(p, pl, st);
        else if (ksize == 1 && is == 1) rc = launch_wgrad<1, 1, 1, 64, 64>(p, pl, st);
        else rc = launch_wgrad<2, 1, 1, 32, 128>(p, pl, st);
        if (rc != SR_OK) return rc;
    }
    ReduceParams r;
    r.partial = scratch; r.out = dwt;
    r.ks = B > 0 ? pl.ks : 0; r.nt = ksize * ksize; r.UP = p.UP; r.VP = p.VP; r.CU = CUc; r.CV = CVc;
    r.slab = C * N;
    // dwt is [k*k][C][N]: regular conv has (u, v) = (c, n); transposed has (u, v) = (n, c)
    r.su = transposed ? 1 : N;
    r.sv = transposed ? N : 1;
    for (int t = 0; t < 9; ++t) r.tmap[t] = t;
    const int64_t total = (int64_t)r.nt * CUc * CVc;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(sr_stream_grid(total, 256)), dim3(256), 0, st, r);
    return sr_launch_status();
}
