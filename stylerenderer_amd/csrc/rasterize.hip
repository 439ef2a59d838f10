// Deterministic z-buffer triangle rasterizer for gfx950, bit-exact against the reference's
// SEQUENTIAL CPU loops (reference op/rasterize.cpp:21-67, arithmetic op/rasterize.h:10-228).
//
// The reference's CUDA kernel (op/rasterize.cu:40-83) is racy: atomicMax on the depth followed by
// a non-atomic re-read and unordered stores of weights/ids (op/rasterize.h:143-154), launched as
// only `b` blocks on the legacy default stream.  This file is a different algorithm:
//
//   pass 1  k_depth_keys   one lane per (sample, triangle), ~b*nf lanes (>> 256 CUs x 64): exact
//           triangle setup, bbox walk, and ONE 64-bit atomicMax per covered pixel on the packed key
//           (orderable(z) << 32) | (0xFFFFFFFE - t).  max key == "largest z, ties -> lowest
//           triangle id" == the outcome of the CPU's in-order `if (zB < z)` sweep.
//   pass 2  k_resolve      one lane per pixel: decode the winner, redo its setup with the same
//           device functions (IEEE ops, no contraction => same bits as pass 1 and as x86), write
//           ids (+ nv*batch), weights, depth and, optionally, interpolated attributes.  Every
//           pixel is written, so no pre-initialised outputs and no float atomics are needed.
//   fp64    a 64-bit depth cannot share a word with the id: pass 1a atomicMax(orderable(z)),
//           pass 1b atomicMin(t) among lanes whose depth equals the maximum, then pass 2.
//
// Bit-exactness rules: compiled with -ffp-contract=off (x86-64 baseline has no
// FMA), correctly-rounded fp32 division (hipcc default, kept explicit in the build), f32
// subnormals preserved (hipcc default kernel mode), `x - .5` evaluated in fp32 (equal to the
// reference's double-literal form for every float input), (h, w) passed swapped into the setup
// exactly like the reference's call sites, float->int64 casts with x86 semantics.
//
//   large triangles: a lane whose bounding box exceeds BIG_BOX pixels parks its triangle in an LDS queue and
//           the whole workgroup walks such boxes together afterwards (same per-pixel arithmetic, so the same
//           bits) — one lane per triangle is right at ~1 px per triangle (a 50k-triangle face at 256^2) and
//           would serialise on a few screen-filling triangles.
//
// Backward: k_dcoeff (API parity with rasterize_gpu_backward, coalesced through LDS), and the fused gradient
// of the autograd Function as a deterministic two-phase GATHER — no float atomics, run-to-run identical:
//   k_first_pix  one lane per pixel: integer atomicMin of the row-major pixel index into a per-(sample, triangle)
//                table — the first pixel a triangle won in box order, its LEADER (order-independent, deterministic);
//   k_grad_pix   one lane per PIXEL (coalesced winner-map reads); the leaders of a workgroup's 256 pixels are
//                compacted through LDS, then each sums, in box order, d/d(3 vertices) and d/d(3 attribute rows) over
//                the triangle's pixels (winner map written by k_resolve) into the triangle's three corner records;
//   k_grad_big   the large triangles k_depth_keys listed: one workgroup each, fixed-order tree over the lanes;
//   k_grad_vert  one lane per (sample, vertex): sums the corner blocks of its incident triangles in the fixed
//                order of a per-topology incidence list (corner-major, ascending triangle id).
// (The reference builds a COO matrix per call and runs sparse.mm, op/rasterize.py:46-77; round 1 of this repo
// scattered with float atomics.)
//
// Host path: the same setup / shading functions are __host__ __device__; sr_rasterize_*_cpu_* run the
// reference's sequential loops (op/rasterize.cpp:21-95) for CPU tensors, which the reference's extension also
// accepts (op/rasterize.cpp:126-150).  Device tensors never take it.
#include <float.h>
#include <stdlib.h>

#include <vector>

#include "common.h"

#define SR_HD __host__ __device__ __forceinline__

namespace {

constexpr int BIG_BOX = 64;     // bounding boxes with more pixels than this are walked by the whole workgroup

template <typename R> struct Lim;
template <> struct Lim<float> {
    static __host__ __device__ float lowest() { return -FLT_MAX; }
};
template <> struct Lim<double> {
    static __host__ __device__ double lowest() { return -DBL_MAX; }
};

// Triangle state held as named scalars (never arrays: runtime-indexed arrays would be demoted
// to scratch memory).  p*: vertices (x, y in screen space after setup, z untouched);
// e0..e2 edge constants, e3..e5 d/dx, e6..e8 d/dy; area = |signed area sum|.
template <typename R>
struct Tri {
    R p0, p1, p2, p3, p4, p5, p6, p7, p8;
    R e0, e1, e2, e3, e4, e5, e6, e7, e8;
    R area;
    int x0, x1, y0, y1;
};

// (int64_t) cast as x86-64 cvtts[sd]2si performs it: NaN / out of range -> INT64_MIN.
template <typename R>
SR_HD long long to_i64_x86(R f) {
    if (f >= (R)-9223372036854775808.0 && f < (R)9223372036854775808.0) return (long long)f;
    return (long long)0x8000000000000000ULL;
}

template <typename R> SR_HD R r_ceil(R x);
template <> SR_HD float r_ceil<float>(float x) { return ceilf(x); }
template <> SR_HD double r_ceil<double>(double x) { return ceil(x); }
template <typename R> SR_HD R r_floor(R x);
template <> SR_HD float r_floor<float>(float x) { return floorf(x); }
template <> SR_HD double r_floor<double>(double x) { return floor(x); }

// One vertex to screen space; false = rejected by the perspective near test.
template <typename R>
SR_HD bool to_screen(R& x, R& y, const R z, R sw, R sh, bool perspective, R eps) {
    if (perspective) {
        if (z >= -eps) return false;
        x = x / -z;
        y = y / -z;
    }
    const R sx = (1 + x) * sw / 2;
    const R sy = (1 - y) * sh / 2;
    x = sx - (R)0.5;
    y = sy - (R)0.5;
    return true;
}

template <typename R>
SR_HD void grow(R& lo, R& hi, R q) {
    // `if (lo > q) lo = q; else if (hi < q) hi = q;` as value selects (reference op/rasterize.h:27-34)
    const bool below = lo > q;
    const bool above = !below && (hi < q);
    lo = below ? q : lo;
    hi = above ? q : hi;
}

// Triangle setup.  sw / sh are what the reference's barycentric() receives as (w, h): the callers
// pass (h_arg, w_arg).  Returns false when the triangle is rejected.
template <typename R>
SR_HD bool tri_setup(Tri<R>& t, long long sw, long long sh, bool perspective,
                                          R eps) {
    const R fw = (R)sw, fh = (R)sh;
    if (!to_screen<R>(t.p0, t.p1, t.p2, fw, fh, perspective, eps)) return false;
    if (!to_screen<R>(t.p3, t.p4, t.p5, fw, fh, perspective, eps)) return false;
    if (!to_screen<R>(t.p6, t.p7, t.p8, fw, fh, perspective, eps)) return false;
    R lo_u = t.p0, hi_u = t.p0, lo_v = t.p1, hi_v = t.p1;
    grow<R>(lo_u, hi_u, t.p3);
    grow<R>(lo_v, hi_v, t.p4);
    grow<R>(lo_u, hi_u, t.p6);
    grow<R>(lo_v, hi_v, t.p7);
    long long x0 = to_i64_x86<R>(r_ceil<R>(lo_u)), x1 = to_i64_x86<R>(r_floor<R>(hi_u));
    long long y0 = to_i64_x86<R>(r_ceil<R>(lo_v)), y1 = to_i64_x86<R>(r_floor<R>(hi_v));
    if (x0 < 0) x0 = 0;
    if (x1 > sw - 1) x1 = sw - 1;
    if (y0 < 0) y0 = 0;
    if (y1 > sh - 1) y1 = sh - 1;
    if (x1 < x0 || y1 < y0) return false;
    t.x0 = (int)x0; t.x1 = (int)x1; t.y0 = (int)y0; t.y1 = (int)y1;

    R m0 = t.p3 * t.p7, m1 = t.p4 * t.p6;
    t.e0 = m0 - m1;
    m0 = t.p1 * t.p6; m1 = t.p0 * t.p7;
    t.e1 = m0 - m1;
    m0 = t.p0 * t.p4; m1 = t.p1 * t.p3;
    t.e2 = m0 - m1;
    R det = t.e0 + t.e1;
    det = det + t.e2;
    if (det > eps) return false;
    t.e3 = t.p4 - t.p7;
    t.e4 = t.p7 - t.p1;
    t.e5 = t.p1 - t.p4;
    t.e6 = t.p6 - t.p3;
    t.e7 = t.p0 - t.p6;
    t.e8 = t.p3 - t.p0;
    if (det < 0) {
        t.e0 = -t.e0; t.e1 = -t.e1; t.e2 = -t.e2;
        t.e3 = -t.e3; t.e4 = -t.e4; t.e5 = -t.e5;
        t.e6 = -t.e6; t.e7 = -t.e7; t.e8 = -t.e8;
        t.area = -det;
    } else {
        t.area = det;
    }
    return true;
}

template <typename R>
SR_HD void edge_values(const Tri<R>& t, R px, R py, R& c0, R& c1, R& c2) {
    R gx = t.e3 * px, gy = t.e6 * py;
    R a = t.e0 + gx;
    c0 = a + gy;
    gx = t.e4 * px; gy = t.e7 * py;
    a = t.e1 + gx;
    c1 = a + gy;
    gx = t.e5 * px; gy = t.e8 * py;
    a = t.e2 + gx;
    c2 = a + gy;
}

#define SR_SEL3(a0, a1, a2, n) ((n) == 0 ? (a0) : ((n) == 1 ? (a1) : (a2)))

// Launders a value through an empty asm so that a later select between such values cannot be
// folded back into a runtime-indexed load of the triangle struct (which would force the whole
// struct into scratch memory).  Emits no instruction.
SR_HD float opaque(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
    return x;
}
SR_HD double opaque(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
    return x;
}

// Barycentric weights from the edge values; false = pixel outside.
template <typename R>
SR_HD bool pixel_weights(const Tri<R>& t, R px, R py, R& c0, R& c1, R& c2, R eps) {
    if (c0 < -eps || c1 < -eps || c2 < -eps) return false;
    if (t.area > eps) {
        R s = c0 + c1;
        s = s + c2;
        c0 = c0 / s;
        c1 = c1 / s;
        c2 = c2 / s;
        return true;
    }
    // zero-area triangle: longest edge (segment) or a point (reference op/rasterize.h:87-121).
    const R E3 = opaque(t.e3), E4 = opaque(t.e4), E5 = opaque(t.e5);
    const R E6 = opaque(t.e6), E7 = opaque(t.e7), E8 = opaque(t.e8);
    const R P0 = opaque(t.p0), P1 = opaque(t.p1), P3 = opaque(t.p3);
    const R P4 = opaque(t.p4), P6 = opaque(t.p6), P7 = opaque(t.p7);
    R l0, l1, l2;
    {
        R a = E3 * E3, b = E6 * E6;
        l0 = a + b;
        a = E4 * E4; b = E7 * E7;
        l1 = a + b;
        a = E5 * E5; b = E8 * E8;
        l2 = a + b;
    }
    l0 = opaque(l0); l1 = opaque(l1); l2 = opaque(l2);
    int i = (l0 > l1) ? 0 : 1;
    const R li_len = (i == 0) ? l0 : l1;
    i = (li_len > l2) ? i : 2;
    const R lmax = (i == 2) ? l2 : li_len;
    const int j = (i == 2) ? 0 : i + 1, k = (j == 2) ? 0 : j + 1;
    const R e3i = SR_SEL3(E3, E4, E5, i), e6i = SR_SEL3(E6, E7, E8, i);
    R ci, cj, ck;
    bool inside;
    if (lmax > eps) {
        const R pkx = SR_SEL3(P0, P3, P6, k), pky = SR_SEL3(P1, P4, P7, k);
        const R pjx = SR_SEL3(P0, P3, P6, j), pjy = SR_SEL3(P1, P4, P7, j);
        R a = -(px - pkx) * e6i;
        R b = (py - pky) * e3i;
        const R lj = a + b;
        a = (px - pjx) * e6i;
        b = (py - pjy) * e3i;
        const R lk = a - b;
        const R li = lj + lk;
        ci = 0;
        cj = lj / li;
        ck = lk / li;
        inside = cj >= -eps && ck >= -eps;
    } else {
        const R pix_ = SR_SEL3(P0, P3, P6, i), piy_ = SR_SEL3(P1, P4, P7, i);
        ci = 1;
        cj = 0;
        ck = 0;
        const R dx = px - pix_, dy = py - piy_;
        const R a = dx * dx, b = dy * dy;
        inside = (a + b) < eps;
    }
    ci = opaque(ci); cj = opaque(cj); ck = opaque(ck);
    c0 = (i == 0) ? ci : ((j == 0) ? cj : ck);
    c1 = (i == 1) ? ci : ((j == 1) ? cj : ck);
    c2 = (i == 2) ? ci : ((j == 2) ? cj : ck);
    return inside;
}

template <typename R>
SR_HD bool pixel_depth(const Tri<R>& t, R& c0, R& c1, R& c2, bool perspective,
                                            R eps, R& z) {
    if (perspective) {
        c0 = c0 / t.p2;
        c1 = c1 / t.p5;
        c2 = c2 / t.p8;
        R s = c0 + c1;
        s = s + c2;
        if (s >= -eps) return false;
        c0 = c0 * s;
        c1 = c1 * s;
        c2 = c2 * s;
        z = s;
    } else {
        const R a = c0 * t.p2, b = c1 * t.p5, d = c2 * t.p8;
        const R s = a + b;
        z = s + d;
    }
    return true;
}

// Total order on floats as unsigned integers; -0 is folded onto +0 so that equal depths tie.
__device__ __forceinline__ unsigned ord32(float z) {
    unsigned u = __float_as_uint(z);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long ord64(double z) {
    unsigned long long u = (unsigned long long)__double_as_longlong(z);
    if (u == 0x8000000000000000ULL) u = 0ULL;
    return (u & 0x8000000000000000ULL) ? ~u : (u | 0x8000000000000000ULL);
}
__host__ __device__ inline unsigned long long key_init_f32() {
    // orderable(-FLT_MAX) = ~0xFF7FFFFF = 0x00800000
    return ((unsigned long long)0x00800000u << 32) | 0xFFFFFFFFull;
}
__host__ __device__ inline unsigned long long key_init_f64() { return 0x0010000000000000ULL; }

// Three consecutive values with ONE memory instruction (global_load_dwordx3 for fp32: element alignment is all a
// global load needs).  The gather kernels are bound by the number of divergent memory instructions a wave issues — the
// texture-address unit walks a fully divergent load one lane per cycle, whatever its width — not by bytes.
template <typename R>
SR_HD void load3(const R* __restrict__ p, R& a, R& b, R& c) {
    R q[3];
    __builtin_memcpy(q, p, 3 * sizeof(R));
    a = q[0]; b = q[1]; c = q[2];
}

template <typename R>
SR_HD bool load_tri(Tri<R>& t, const R* __restrict__ vs,
                                         const long long* __restrict__ fs, long long ti,
                                         long long nv, long long& i0, long long& i1, long long& i2) {
    i0 = fs[3 * ti];
    i1 = fs[3 * ti + 1];
    i2 = fs[3 * ti + 2];
    if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= nv || i1 >= nv || i2 >= nv) return false;
    load3<R>(vs + 3 * i0, t.p0, t.p1, t.p2);
    load3<R>(vs + 3 * i1, t.p3, t.p4, t.p5);
    load3<R>(vs + 3 * i2, t.p6, t.p7, t.p8);
    return true;
}

// (also resets the gradient state of the call when there is one: the big-triangle counter, the leader table — left
// EMPTY by this path — and the state word that tells the gradient pass so: 0 = table filled but not built)
__global__ __launch_bounds__(256) void k_fill_u64(unsigned long long* p, unsigned long long v,
                                                  long long n, int* counter, int* first, long long n_first) {
    if (counter && blockIdx.x == 0 && threadIdx.x == 0) {
        *counter = 0;
        if (first) first[n_first] = 0;
    }
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
    if (first)
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_first; i += stride)
            first[i] = 0x7FFFFFFF;
}
__global__ __launch_bounds__(256) void k_fill_u32(unsigned* p, unsigned v, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// Weights + depth of triangle `t` at pixel (x, y); false = not covered.  The ONE per-pixel evaluation every
// pass uses (depth keys, resolve, gradient walk, host loops), so they all see the same bits.
template <typename R>
SR_HD bool shade(const Tri<R>& t, int x, int y, bool perspective, R eps, R& c0, R& c1, R& c2, R& z) {
    const R px = (R)x, py = (R)y;
    edge_values<R>(t, px, py, c0, c1, c2);
    if (!pixel_weights<R>(t, px, py, c0, c1, c2, eps)) return false;
    return pixel_depth<R>(t, c0, c1, c2, perspective, eps, z);
}

// MODE 0: fp32 packed key.  MODE 1: fp64 depth max.  MODE 2: fp64 lowest id among depth-maxima.
template <typename R, int MODE>
__device__ __forceinline__ void depth_test(const Tri<R>& t, int x, int y, long long w, long long hw,
                                           bool perspective, R eps, unsigned long long* __restrict__ ks,
                                           unsigned* __restrict__ tm, unsigned ti) {
    const long long pix = x + (long long)y * w;
    if (pix >= hw) return;                  // the reference would write out of bounds (w > h)
    R c0, c1, c2, z;
    if (!shade<R>(t, x, y, perspective, eps, c0, c1, c2, z)) return;
    if (!(z == z)) return;                  // NaN never passes `zB < z`
    if (MODE == 0) {
        const unsigned long long key =
            ((unsigned long long)ord32((float)z) << 32) | (unsigned long long)(0xFFFFFFFEu - ti);
        if (key > ks[pix]) atomicMax(&ks[pix], key);
    } else if (MODE == 1) {
        const unsigned long long key = ord64((double)z);
        if (key > ks[pix]) atomicMax(&ks[pix], key);
    } else {
        const unsigned long long key = ord64((double)z);
        if (key == ks[pix] && key > key_init_f64()) atomicMin(&tm[pix], ti);
    }
}

template <typename R, int MODE>
__device__ __forceinline__ void depth_keys_body(long long b, long long nv, long long nf, long long h, long long w,
                                                bool repeat_v, bool repeat_f, bool perspective,
                                                const R* __restrict__ v, const long long* __restrict__ f,
                                                unsigned long long* __restrict__ keys, unsigned* __restrict__ tmin,
                                                int* __restrict__ big, R eps) {
    __shared__ int s_nbig;
    __shared__ int s_big[256];
    if (threadIdx.x == 0) s_nbig = 0;
    __syncthreads();
    const long long hw = h * w;
    const long long base = (long long)blockIdx.x * 256;
    {
        const long long g = base + threadIdx.x;
        if (g < b * nf) {
            const long long s = g / nf, ti = g - s * nf;
            const R* vs = repeat_v ? v : v + s * nv * 3;
            const long long* fs = repeat_f ? f : f + s * nf * 3;
            Tri<R> t;
            long long i0, i1, i2;
            if (load_tri<R>(t, vs, fs, ti, nv, i0, i1, i2) && tri_setup<R>(t, h, w, perspective, eps)) {
                const long long box = (long long)(t.x1 - t.x0 + 1) * (t.y1 - t.y0 + 1);
                if (box > BIG_BOX) {
                    s_big[atomicAdd(&s_nbig, 1)] = threadIdx.x;        // walked by the workgroup below
                    // ... and remembered for the gradient pass (list order is irrelevant: one row each)
                    if (MODE != 2 && big) big[1 + atomicAdd(big, 1)] = (int)g;
                } else {
                    for (int y = t.y0; y <= t.y1; ++y)
                        for (int x = t.x0; x <= t.x1; ++x)
                            depth_test<R, MODE>(t, x, y, w, hw, perspective, eps, keys + s * hw, tmin + s * hw,
                                                (unsigned)ti);
                }
            }
        }
    }
    __syncthreads();
    const int nbig = s_nbig;
    for (int q = 0; q < nbig; ++q) {
        const long long g = base + s_big[q];
        const long long s = g / nf, ti = g - s * nf;
        const R* vs = repeat_v ? v : v + s * nv * 3;
        const long long* fs = repeat_f ? f : f + s * nf * 3;
        Tri<R> t;
        long long i0, i1, i2;
        load_tri<R>(t, vs, fs, ti, nv, i0, i1, i2);
        tri_setup<R>(t, h, w, perspective, eps);
        const int bw = t.x1 - t.x0 + 1;
        const long long npx = (long long)bw * (t.y1 - t.y0 + 1);
        for (long long p = threadIdx.x; p < npx; p += 256) {
            const int yy = (int)(p / bw);
            depth_test<R, MODE>(t, t.x0 + (int)(p - (long long)yy * bw), t.y0 + yy, w, hw, perspective, eps,
                                keys + s * hw, tmin + s * hw, (unsigned)ti);
        }
    }
}

template <typename R, int MODE>
__global__ __launch_bounds__(256) void k_depth_keys(long long b, long long nv, long long nf,
                                                    long long h, long long w, bool repeat_v,
                                                    bool repeat_f, bool perspective,
                                                    const R* __restrict__ v,
                                                    const long long* __restrict__ f,
                                                    unsigned long long* __restrict__ keys,
                                                    unsigned* __restrict__ tmin, int* __restrict__ big,
                                                    R eps) {
    depth_keys_body<R, MODE>(b, nv, nf, h, w, repeat_v, repeat_f, perspective, v, f, keys, tmin, big, eps);
}

template <typename R>
__device__ __forceinline__ void resolve_body(long long b, long long nv, long long nf, long long h, long long w,
                                             bool repeat_v, bool repeat_f, bool perspective, const R* __restrict__ v,
                                             const long long* __restrict__ f,
                                             const unsigned long long* __restrict__ keys,
                                             const unsigned* __restrict__ tmin, long long* __restrict__ index,
                                             R* __restrict__ coeff, R* __restrict__ zbuf, const R* __restrict__ tex,
                                             long long tex_c, R* __restrict__ attr, int* __restrict__ win, R eps,
                                             bool chw) {
    const long long hw = h * w;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= b * hw) return;
    const long long s = g / hw, pix = g - s * hw;
    long long ti = -1;
    if (sizeof(R) == 4) {
        const unsigned long long key = keys[g];
        if (key != key_init_f32()) ti = (long long)(0xFFFFFFFEu - (unsigned)(key & 0xFFFFFFFFull));
    } else {
        const unsigned tm = tmin[g];
        if (tm != 0xFFFFFFFFu) ti = tm;
    }
    long long i0 = 0, i1 = 0, i2 = 0;
    R c0 = 0, c1 = 0, c2 = 0;
    R z = Lim<R>::lowest();
    if (ti >= 0) {
        const R* vs = repeat_v ? v : v + s * nv * 3;
        const long long* fs = repeat_f ? f : f + s * nf * 3;
        Tri<R> t;
        load_tri<R>(t, vs, fs, ti, nv, i0, i1, i2);
        tri_setup<R>(t, h, w, perspective, eps);
        const int y = (int)(pix / w), x = (int)(pix - (long long)y * w);
        shade<R>(t, x, y, perspective, eps, c0, c1, c2, z);
        const long long shift = repeat_v ? 0 : nv * s;
        i0 += shift; i1 += shift; i2 += shift;
    }
    if (index) {
        index[3 * g] = i0;
        index[3 * g + 1] = i1;
        index[3 * g + 2] = i2;
    }
    if (coeff) {
        coeff[3 * g] = c0;
        coeff[3 * g + 1] = c1;
        coeff[3 * g + 2] = c2;
    }
    if (zbuf) zbuf[g] = z;
    if (win) win[g] = (int)ti;
    if (attr) {
        for (long long ch = 0; ch < tex_c; ++ch) {
            const R a0 = tex[i0 * tex_c + ch] * c0;
            const R a1 = tex[i1 * tex_c + ch] * c1;
            const R a2 = tex[i2 * tex_c + ch] * c2;
            const R s01 = a0 + a1;
            attr[chw ? (s * tex_c + ch) * hw + pix : g * tex_c + ch] = s01 + a2;     // [b,c,h,w] or [b,h,w,c]
        }
    }
}

template <typename R>
__global__ __launch_bounds__(256) void k_resolve(long long b, long long nv, long long nf, long long h,
                                                 long long w, bool repeat_v, bool repeat_f,
                                                 bool perspective, const R* __restrict__ v,
                                                 const long long* __restrict__ f,
                                                 const unsigned long long* __restrict__ keys,
                                                 const unsigned* __restrict__ tmin,
                                                 long long* __restrict__ index, R* __restrict__ coeff,
                                                 R* __restrict__ zbuf, const R* __restrict__ tex,
                                                 long long tex_c, R* __restrict__ attr,
                                                 int* __restrict__ win, R eps, bool chw) {
    resolve_body<R>(b, nv, nf, h, w, repeat_v, repeat_f, perspective, v, f, keys, tmin, index, coeff, zbuf, tex, tex_c,
                    attr, win, eps, chw);
}

// ---- the same mesh at several resolutions in three launches (GeneratorWithMap: one normal map per synthesis
// resolution, reference model.py:255-262 — seven calls of three launches each at 256^2, every one ~5 us at the training
// and inversion batch sizes).  blockIdx.y selects the level; its extents and buffers come from a table passed by value.
// The bodies are the single-level kernels' (same expressions, same order: same bits).
constexpr int RASTER_MAX_LEVELS = SR_RASTER_MAX_LEVELS;
struct RasterLevels {
    int n;
    int res_h[RASTER_MAX_LEVELS], res_w[RASTER_MAX_LEVELS];
    unsigned long long* keys[RASTER_MAX_LEVELS];
    int* big[RASTER_MAX_LEVELS];              // gradient state of the level (or NULL)
    int* win[RASTER_MAX_LEVELS];
    float* attr[RASTER_MAX_LEVELS];
};

__global__ __launch_bounds__(256) void k_fill_levels(const RasterLevels t, long long b, long long nf) {
    const int l = blockIdx.y;
    const long long n = b * t.res_h[l] * t.res_w[l];
    unsigned long long* p = t.keys[l];
    int* counter = t.big[l];
    int* first = (counter && t.win[l]) ? counter + 1 + b * nf : nullptr;
    const long long n_first = b * nf;
    if (counter && blockIdx.x == 0 && threadIdx.x == 0) {
        *counter = 0;
        if (first) first[n_first] = 0;
    }
    const long long stride = (long long)gridDim.x * blockDim.x;
    const unsigned long long v = key_init_f32();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
    if (first)
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_first; i += stride) first[i] = 0x7FFFFFFF;
}

__global__ __launch_bounds__(256) void k_depth_keys_levels(const RasterLevels t, long long b, long long nv, long long nf,
                                                           bool repeat_v, bool repeat_f, bool perspective,
                                                           const float* __restrict__ v, const long long* __restrict__ f,
                                                           float eps) {
    const int l = blockIdx.y;
    depth_keys_body<float, 0>(b, nv, nf, t.res_h[l], t.res_w[l], repeat_v, repeat_f, perspective, v, f, t.keys[l], nullptr,
                              t.big[l], eps);
}

__global__ __launch_bounds__(256) void k_resolve_levels(const RasterLevels t, long long b, long long nv, long long nf,
                                                        bool repeat_v, bool repeat_f, bool perspective,
                                                        const float* __restrict__ v, const long long* __restrict__ f,
                                                        const float* __restrict__ tex, long long tex_c, float eps,
                                                        bool chw) {
    const int l = blockIdx.y;
    resolve_body<float>(b, nv, nf, t.res_h[l], t.res_w[l], repeat_v, repeat_f, perspective, v, f, t.keys[l], nullptr, nullptr,
                        nullptr, nullptr, tex, tex_c, t.attr[l], t.win[l], eps, chw);
}

// ---- fp32 forward with LDS-staged triangle tiles (square images) ------------------------------------------------
// The global-key path above pays 8 B/pixel of key buffer (fill, atomics in L2, re-read) and walks bounding boxes one
// lane per triangle (lanes of a wave wait for the largest box).  The tiled path:
//   k_tile_bin     one lane per (sample, triangle): the SAME load + setup; accepted small triangles (box <= BIG_BOX
//                  pixels, hence <= 4 tiles) are appended to the list of every 32x32 screen tile their box touches
//                  (wave-aggregated counter atomics); large boxes, and entries beyond a tile's capacity, go to the
//                  sample's `wide` list that every tile of the sample scans;
//   k_tile_raster  one workgroup per (sample, tile): the depth keys of the tile live in LDS (8 KB, ds_max_u64).  List
//                  entries are set up 256 at a time into LDS records; the (triangle, pixel-of-clipped-box) pairs are
//                  enumerated by a block prefix sum so that EVERY lane shades one pixel per step whatever the box
//                  sizes are; then the tile is resolved from LDS and each output is written once.
// Same `tri_setup` / `shade` device functions, same packed key (max z, ties -> lowest id): same bits as the global
// path and the CPU loops.  No key buffer, no fill pass, no second kernel for the resolve.
constexpr int TILE = 32;            // screen tile edge (pixels)
constexpr int TILE_CAP = 2048;      // list entries per (sample, tile); overflow spills to the sample's wide list
constexpr int REC_F = 13;           // floats per LDS triangle record: z0 z1 z2, e0..e8, area (+ 4 integer fields)

// The per-triangle work of these kernels is vector-ALU bound (k_depth_keys: ~1 100 VALU instructions, 80 us for 3.2 M
// lanes), so the tiled kernels use slimmer device-only forms of `tri_setup` — the SAME floating-point expressions in
// the same order (same bits), without what costs instructions and changes no result:
//   * float -> int64 conversion with x86 semantics is ~20 instructions of emulation, four times per triangle; the
//     bounding box only ever meets [0, res - 1], where a SATURATED 32-bit conversion compares identically (NaN and
//     values >= 2^63 behave as "very negative", like cvttss2si's INT64_MIN);
//   * the projection mode is a template parameter (no division code in the orthographic kernels);
//   * indices are 32-bit, the sample comes from blockIdx.y;
//   * the resolve step needs the edge constants only: no bounding box, no rejection tests.
__device__ __forceinline__ int sat_i32_x86(float f) {
    const int i = (int)f;                                   // v_cvt_i32_f32: saturates, NaN -> 0
    return (f >= 9223372036854775808.0f || f != f) ? (int)0x80000000 : i;
}

template <bool PERSP>
__device__ __forceinline__ bool screen3(Tri<float>& t, float fres, float eps) {
    if (!to_screen<float>(t.p0, t.p1, t.p2, fres, fres, PERSP, eps)) return false;
    if (!to_screen<float>(t.p3, t.p4, t.p5, fres, fres, PERSP, eps)) return false;
    return to_screen<float>(t.p6, t.p7, t.p8, fres, fres, PERSP, eps);
}

__device__ __forceinline__ bool bounds_i32(Tri<float>& t, int res) {
    float lo_u = t.p0, hi_u = t.p0, lo_v = t.p1, hi_v = t.p1;
    grow<float>(lo_u, hi_u, t.p3);
    grow<float>(lo_v, hi_v, t.p4);
    grow<float>(lo_u, hi_u, t.p6);
    grow<float>(lo_v, hi_v, t.p7);
    int x0 = sat_i32_x86(ceilf(lo_u)), x1 = sat_i32_x86(floorf(hi_u));
    int y0 = sat_i32_x86(ceilf(lo_v)), y1 = sat_i32_x86(floorf(hi_v));
    x0 = x0 < 0 ? 0 : x0;
    x1 = x1 > res - 1 ? res - 1 : x1;
    y0 = y0 < 0 ? 0 : y0;
    y1 = y1 > res - 1 ? res - 1 : y1;
    t.x0 = x0; t.x1 = x1; t.y0 = y0; t.y1 = y1;
    return x1 >= x0 && y1 >= y0;
}

// e0..e2 and the signed area sum (tri_setup's `det`)
__device__ __forceinline__ float edge_origin(Tri<float>& t) {
    float m0 = t.p3 * t.p7, m1 = t.p4 * t.p6;
    t.e0 = m0 - m1;
    m0 = t.p1 * t.p6; m1 = t.p0 * t.p7;
    t.e1 = m0 - m1;
    m0 = t.p0 * t.p4; m1 = t.p1 * t.p3;
    t.e2 = m0 - m1;
    float det = t.e0 + t.e1;
    det = det + t.e2;
    return det;
}

__device__ __forceinline__ void edge_slopes(Tri<float>& t, float det) {
    t.e3 = t.p4 - t.p7;
    t.e4 = t.p7 - t.p1;
    t.e5 = t.p1 - t.p4;
    t.e6 = t.p6 - t.p3;
    t.e7 = t.p0 - t.p6;
    t.e8 = t.p3 - t.p0;
    if (det < 0) {
        t.e0 = -t.e0; t.e1 = -t.e1; t.e2 = -t.e2;
        t.e3 = -t.e3; t.e4 = -t.e4; t.e5 = -t.e5;
        t.e6 = -t.e6; t.e7 = -t.e7; t.e8 = -t.e8;
        t.area = -det;
    } else {
        t.area = det;
    }
}

// accept / reject + bounding box (what the binning needs); on a square image of `res` pixels
template <bool PERSP>
__device__ __forceinline__ bool tri_bounds_fast(Tri<float>& t, int res, float eps) {
    if (!screen3<PERSP>(t, (float)res, eps)) return false;
    if (!bounds_i32(t, res)) return false;
    return !(edge_origin(t) > eps);
}

// the whole of tri_setup for a triangle that is known to be accepted
template <bool PERSP>
__device__ __forceinline__ void tri_setup_accepted(Tri<float>& t, int res, float eps, bool want_box) {
    screen3<PERSP>(t, (float)res, eps);
    if (want_box) bounds_i32(t, res);
    edge_slopes(t, edge_origin(t));
}

__device__ __forceinline__ bool load_tri32(Tri<float>& t, const float* __restrict__ vs,
                                           const long long* __restrict__ fs, unsigned ti, unsigned nv,
                                           unsigned& i0, unsigned& i1, unsigned& i2) {
    const long long a = fs[3 * (size_t)ti], bq = fs[3 * (size_t)ti + 1], c = fs[3 * (size_t)ti + 2];
    if ((unsigned long long)a >= nv || (unsigned long long)bq >= nv || (unsigned long long)c >= nv) return false;
    i0 = (unsigned)a; i1 = (unsigned)bq; i2 = (unsigned)c;
    load3<float>(vs + 3 * (size_t)i0, t.p0, t.p1, t.p2);
    load3<float>(vs + 3 * (size_t)i1, t.p3, t.p4, t.p5);
    load3<float>(vs + 3 * (size_t)i2, t.p6, t.p7, t.p8);
    return true;
}

// (also resets the gradient state of the call when there is one: the big-triangle counter, the leader table
// `first`, which k_tile_raster fills while it resolves its tiles, and the state word behind it: 1 = table built by the
// forward pass)
__global__ __launch_bounds__(256) void k_tile_zero(unsigned* __restrict__ p, long long n, int* __restrict__ big,
                                                   int* __restrict__ first, long long n_first) {
    if (big && blockIdx.x == 0 && threadIdx.x == 0) {
        *big = 0;
        if (first) first[n_first] = 1;
    }
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = 0u;
    if (first)
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_first; i += stride) first[i] = 0x7FFFFFFF;
}

// Wave-level aggregation of counter increments, split so that the atomics of several groups are in flight together:
// `wave_group` only votes (no memory operation): the lanes of `active` that share `idx` form a group; returns the mask
// of the caller's group.  The lowest lane of a group issues ONE atomicAdd(count of the group) — every group of the wave
// in the same instruction — and the others fetch the base from it (`wave_slot`).
__device__ __forceinline__ unsigned long long wave_group(unsigned idx, bool active) {
    unsigned long long mine = 0;
    unsigned long long todo = __ballot(active);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int lead = __ffsll((long long)todo) - 1;
        const unsigned key = (unsigned)__shfl((int)idx, lead, SR_WAVE);
        const unsigned long long same = __ballot(active && idx == key) & todo;
        if ((same >> lane) & 1ull) mine = same;
        todo &= ~same;
    }
    return mine;
}
__device__ __forceinline__ unsigned wave_slot(unsigned long long group, unsigned leader_base) {
    const int lane = threadIdx.x & 63;
    const int lead = group ? __ffsll((long long)group) - 1 : lane;
    const unsigned base = (unsigned)__shfl((int)leader_base, lead, SR_WAVE);
    return base + (unsigned)__popcll(group & ((1ull << lane) - 1ull));
}

// grid (ceil(nf / 256), b)
template <bool PERSP>
__global__ __launch_bounds__(256) void k_tile_bin(unsigned nv, unsigned nf, int res, bool repeat_v, bool repeat_f,
                                                  const float* __restrict__ v, const long long* __restrict__ f,
                                                  unsigned* __restrict__ tile_cnt, unsigned* __restrict__ tile_list,
                                                  unsigned* __restrict__ wide_cnt, unsigned* __restrict__ wide_list,
                                                  int* __restrict__ big, int ntx, float eps) {
    const unsigned ti = blockIdx.x * 256 + threadIdx.x;
    const unsigned s = blockIdx.y;
    Tri<float> t;
    bool ok = false;
    if (ti < nf) {
        const float* vs = repeat_v ? v : v + (size_t)s * nv * 3;
        const long long* fs = repeat_f ? f : f + (size_t)s * nf * 3;
        unsigned i0, i1, i2;
        ok = load_tri32(t, vs, fs, ti, nv, i0, i1, i2) && tri_bounds_fast<PERSP>(t, res, eps);
    }
    bool wide = ok && ((long long)(t.x1 - t.x0 + 1) * (t.y1 - t.y0 + 1) > BIG_BOX);
    if (wide && big) big[1 + atomicAdd(big, 1)] = (int)(s * nf + ti);   // (gradient pass: one row each, any order)
    const int ntile = ntx * ntx;
    const bool small_ok = ok && !wide;
    const int tx0 = small_ok ? t.x0 / TILE : 0, tx1 = small_ok ? t.x1 / TILE : 0;
    const int ty0 = small_ok ? t.y0 / TILE : 0, ty1 = small_ok ? t.y1 / TILE : 0;
    const int lane = threadIdx.x & 63;
    // a small box (<= 64 pixels) touches <= 4 tiles: 2 x 2, or up to 3 in a row.  The four common steps vote first and
    // then have their counter atomics in flight together (a returning atomic is a ~microsecond round trip)
    const bool h00 = small_ok, h01 = small_ok && tx0 + 1 <= tx1, h10 = small_ok && ty0 + 1 <= ty1, h11 = h01 && h10;
    const unsigned base_tile = s * ntile + ty0 * ntx + tx0;
    const unsigned t00 = base_tile, t01 = base_tile + 1, t10 = base_tile + ntx, t11 = base_tile + ntx + 1;
    const unsigned long long g00 = wave_group(t00, h00), g01 = wave_group(t01, h01), g10 = wave_group(t10, h10),
                             g11 = wave_group(t11, h11);
    unsigned b00 = 0, b01 = 0, b10 = 0, b11 = 0;
    if (g00 && lane == __ffsll((long long)g00) - 1) b00 = atomicAdd(&tile_cnt[t00], (unsigned)__popcll(g00));
    if (g01 && lane == __ffsll((long long)g01) - 1) b01 = atomicAdd(&tile_cnt[t01], (unsigned)__popcll(g01));
    if (g10 && lane == __ffsll((long long)g10) - 1) b10 = atomicAdd(&tile_cnt[t10], (unsigned)__popcll(g10));
    if (g11 && lane == __ffsll((long long)g11) - 1) b11 = atomicAdd(&tile_cnt[t11], (unsigned)__popcll(g11));
    const unsigned s00 = wave_slot(g00, b00), s01 = wave_slot(g01, b01), s10 = wave_slot(g10, b10),
                   s11 = wave_slot(g11, b11);
#define SR_TILE_PUT(H, T, S)                                                        \
    if (H) {                                                                        \
        if ((S) < (unsigned)TILE_CAP) tile_list[(size_t)(T) * TILE_CAP + (S)] = ti; \
        else wide = true; /* this tile is full: every tile of the sample scans it */ \
    }
    SR_TILE_PUT(h00, t00, s00)
    SR_TILE_PUT(h01, t01, s01)
    SR_TILE_PUT(h10, t10, s10)
    SR_TILE_PUT(h11, t11, s11)
    // third tile of a 1-pixel-high / -wide strip (rare)
    const bool h02 = small_ok && tx0 + 2 <= tx1, h20 = small_ok && ty0 + 2 <= ty1;
    if (__ballot(h02 || h20)) {
        const unsigned t02 = base_tile + 2, t20 = base_tile + 2 * ntx;
        const unsigned long long g02 = wave_group(t02, h02), g20 = wave_group(t20, h20);
        unsigned b02 = 0, b20 = 0;
        if (g02 && lane == __ffsll((long long)g02) - 1) b02 = atomicAdd(&tile_cnt[t02], (unsigned)__popcll(g02));
        if (g20 && lane == __ffsll((long long)g20) - 1) b20 = atomicAdd(&tile_cnt[t20], (unsigned)__popcll(g20));
        const unsigned s02 = wave_slot(g02, b02), s20 = wave_slot(g20, b20);
        SR_TILE_PUT(h02, t02, s02)
        SR_TILE_PUT(h20, t20, s20)
    }
#undef SR_TILE_PUT
    if (__ballot(wide)) {
        const unsigned long long gw = wave_group(s, wide);
        unsigned bw = 0;
        if (gw && lane == __ffsll((long long)gw) - 1) bw = atomicAdd(&wide_cnt[s], (unsigned)__popcll(gw));
        const unsigned slot = wave_slot(gw, bw);
        if (wide) wide_list[(size_t)s * nf + slot] = ti;
    }
}

// The same binning with the counters of a workgroup aggregated in LDS (images of up to BIN_LDS_TILES tiles): every hit is
// one returning LDS atomic (its slot inside the workgroup's share of the tile), then ONE global atomic per tile the
// workgroup touched reserves the share, then the entries are stored.  No wave votes; 256 consecutive triangles of a mesh
// touch a handful of tiles, so the global atomics drop by the number of waves and the list stores of a tile are contiguous.
constexpr int BIN_LDS_TILES = 1024;
template <bool PERSP>
__global__ __launch_bounds__(256) void k_tile_bin_lds(unsigned nv, unsigned nf, int res, bool repeat_v, bool repeat_f,
                                                      const float* __restrict__ v, const long long* __restrict__ f,
                                                      unsigned* __restrict__ tile_cnt, unsigned* __restrict__ tile_list,
                                                      unsigned* __restrict__ wide_cnt, unsigned* __restrict__ wide_list,
                                                      int* __restrict__ big, int ntx, float eps) {
    __shared__ unsigned s_cnt[BIN_LDS_TILES + 1], s_base[BIN_LDS_TILES + 1];       // [ntile] = the sample's wide list
    const int ntile = ntx * ntx;
    for (int i = threadIdx.x; i <= ntile; i += 256) s_cnt[i] = 0u;
    const unsigned ti = blockIdx.x * 256 + threadIdx.x;
    const unsigned s = blockIdx.y;
    Tri<float> t;
    bool ok = false;
    if (ti < nf) {
        const float* vs = repeat_v ? v : v + (size_t)s * nv * 3;
        const long long* fs = repeat_f ? f : f + (size_t)s * nf * 3;
        unsigned i0, i1, i2;
        ok = load_tri32(t, vs, fs, ti, nv, i0, i1, i2) && tri_bounds_fast<PERSP>(t, res, eps);
    }
    const bool wide0 = ok && ((long long)(t.x1 - t.x0 + 1) * (t.y1 - t.y0 + 1) > BIG_BOX);
    if (wide0 && big) big[1 + atomicAdd(big, 1)] = (int)(s * nf + ti);   // (gradient pass: one row each, any order)
    const bool small_ok = ok && !wide0;
    // a small box (<= 64 pixels) touches <= 4 tiles as 2 x 2, or up to 3 in a row / column
    const int tx0 = small_ok ? t.x0 / TILE : 0, tx1 = small_ok ? t.x1 / TILE : -1;
    const int ty0 = small_ok ? t.y0 / TILE : 0, ty1 = small_ok ? t.y1 / TILE : -1;
    __syncthreads();
    int tl[6];
    unsigned slot[6];
    int n_hit = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        // (0,0) (1,0) (0,1) (1,1) (2,0) (0,2): the last two only for one-pixel-high / -wide strips
        const int dx = k == 1 || k == 3 ? 1 : (k == 4 ? 2 : 0), dy = k == 2 || k == 3 ? 1 : (k == 5 ? 2 : 0);
        const bool hit = tx0 + dx <= tx1 && ty0 + dy <= ty1;      // (3 x 2 tiles would need a box of >= 68 pixels)
        tl[k] = hit ? (ty0 + dy) * ntx + tx0 + dx : -1;
        slot[k] = hit ? atomicAdd(&s_cnt[tl[k]], 1u) : 0u;
        n_hit += hit;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ntile; i += 256) {
        const unsigned c = s_cnt[i];
        if (c) s_base[i] = atomicAdd(&tile_cnt[s * ntile + i], c);
    }
    __syncthreads();
    bool wide = wide0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (tl[k] < 0) continue;
        const unsigned pos = s_base[tl[k]] + slot[k];
        if (pos < (unsigned)TILE_CAP) tile_list[(size_t)(s * ntile + tl[k]) * TILE_CAP + pos] = ti;
        else wide = true;                   // this tile is full: every tile of the sample scans the triangle
    }
    const unsigned wslot = wide ? atomicAdd(&s_cnt[ntile], 1u) : 0u;
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt[ntile]) s_base[ntile] = atomicAdd(&wide_cnt[s], s_cnt[ntile]);
    __syncthreads();
    if (wide) wide_list[(size_t)s * nf + s_base[ntile] + wslot] = ti;
}

// Exclusive prefix sum of one int per lane over the 256 lanes; returns the lane's offset, total in `total`.
__device__ __forceinline__ int block_scan_256(int x, int* s_wave, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(incl, off, SR_WAVE);
        if (lane >= off) incl += y;
    }
    __syncthreads();                       // (s_wave may still be read from the previous round)
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    const int w0 = s_wave[0], w1 = s_wave[1], w2 = s_wave[2], w3 = s_wave[3];
    total = w0 + w1 + w2 + w3;
    const int before = (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
    return before + incl - x;
}

template <bool PERSP>
__global__ __launch_bounds__(256) void k_tile_raster(unsigned b, unsigned nv, unsigned nf, int res, bool repeat_v,
                                                     bool repeat_f, const float* __restrict__ v,
                                                     const long long* __restrict__ f,
                                                     const unsigned* __restrict__ tile_cnt,
                                                     const unsigned* __restrict__ tile_list,
                                                     const unsigned* __restrict__ wide_cnt,
                                                     const unsigned* __restrict__ wide_list, int ntx,
                                                     long long* __restrict__ index, float* __restrict__ coeff,
                                                     float* __restrict__ zbuf, const float* __restrict__ tex,
                                                     int tex_c, float* __restrict__ attr, int* __restrict__ win,
                                                     int* __restrict__ first, float eps, bool chw) {
    __shared__ unsigned long long s_key[TILE * TILE];
    __shared__ float s_rec[REC_F][256];
    __shared__ int s_box[4][256];          // clipped box x0, y0, width; triangle id
    __shared__ int s_pre[257];
    __shared__ int s_wave[4];
    const unsigned ntile = (unsigned)(ntx * ntx);
    // all tiles of a sample on one XCD (blockIdx round-robins over the 8 XCDs): its vertices stay in that L2
    const unsigned nblk = b * ntile;
    unsigned unit = blockIdx.x;
    {
        const unsigned per = nblk / SR_NUM_XCD;
        if (per * SR_NUM_XCD == nblk) unit = (blockIdx.x % SR_NUM_XCD) * per + blockIdx.x / SR_NUM_XCD;
    }
    const unsigned s = unit / ntile;
    const int tile = (int)(unit - s * ntile);
    const int ox = (tile % ntx) * TILE, oy = (tile / ntx) * TILE;
    const float* vs = repeat_v ? v : v + (size_t)s * nv * 3;
    const long long* fs = repeat_f ? f : f + (size_t)s * nf * 3;
    for (int i = threadIdx.x; i < TILE * TILE; i += 256) s_key[i] = key_init_f32();
    __syncthreads();
    unsigned n_own = tile_cnt[s * ntile + tile];
    if (n_own > (unsigned)TILE_CAP) n_own = TILE_CAP;
    const unsigned n_wide = wide_cnt[s];
    const unsigned* own = tile_list + (size_t)(s * ntile + tile) * TILE_CAP;
    const unsigned* wl = wide_list + (size_t)s * nf;
    const unsigned n_all = n_own + n_wide;
    for (unsigned c0 = 0; c0 < n_all; c0 += 256) {
        const unsigned e = c0 + threadIdx.x;
        int npx = 0;
        if (e < n_all) {
            const unsigned ti = e < n_own ? own[e] : wl[e - n_own];
            Tri<float> t;
            unsigned i0, i1, i2;
            load_tri32(t, vs, fs, ti, nv, i0, i1, i2);
            tri_setup_accepted<PERSP>(t, res, eps, true);          // accepted in k_tile_bin: same inputs, same bits
            const int x0 = t.x0 > ox ? t.x0 : ox, x1 = t.x1 < ox + TILE - 1 ? t.x1 : ox + TILE - 1;
            const int y0 = t.y0 > oy ? t.y0 : oy, y1 = t.y1 < oy + TILE - 1 ? t.y1 : oy + TILE - 1;
            if (x1 >= x0 && y1 >= y0) npx = (x1 - x0 + 1) * (y1 - y0 + 1);
            const int l = threadIdx.x;
            s_rec[0][l] = t.p2; s_rec[1][l] = t.p5; s_rec[2][l] = t.p8;
            s_rec[3][l] = t.e0; s_rec[4][l] = t.e1; s_rec[5][l] = t.e2; s_rec[6][l] = t.e3; s_rec[7][l] = t.e4;
            s_rec[8][l] = t.e5; s_rec[9][l] = t.e6; s_rec[10][l] = t.e7; s_rec[11][l] = t.e8; s_rec[12][l] = t.area;
            s_box[0][l] = x0; s_box[1][l] = y0; s_box[3][l] = (int)ti;
            // box width and ceil(2^16 / width): item -> row is a multiply + shift (exact: item < 1024, width <= 32)
            // (a wide-list triangle whose clipped box misses this tile has npx = 0 and possibly x1 = x0 - 1: nothing reads
            // its entry, and the division is not evaluated)
            s_box[2][l] = npx > 0 ? (x1 - x0 + 1) | ((65536 + (x1 - x0)) / (x1 - x0 + 1)) << 8 : 0;
        }
        int total;
        const int before = block_scan_256(npx, s_wave, total);
        s_pre[threadIdx.x] = before;
        if (threadIdx.x == 255) s_pre[256] = total;
        __syncthreads();
        for (int item = threadIdx.x; item < total; item += 256) {
            int lo = 0, hi = 256;                       // largest j with s_pre[j] <= item
#pragma unroll
            for (int step = 0; step < 8; ++step) {
                const int mid = (lo + hi) >> 1;
                if (s_pre[mid] <= item) lo = mid; else hi = mid;
            }
            const int j = lo, local = item - s_pre[j];
            const int bwr = s_box[2][j], bw = bwr & 0xFF;
            const int yy = (int)(((unsigned)local * (unsigned)(bwr >> 8)) >> 16), xx = local - yy * bw;
            const int x = s_box[0][j] + xx, y = s_box[1][j] + yy;
            Tri<float> t;
            t.p2 = s_rec[0][j]; t.p5 = s_rec[1][j]; t.p8 = s_rec[2][j];
            t.e0 = s_rec[3][j]; t.e1 = s_rec[4][j]; t.e2 = s_rec[5][j]; t.e3 = s_rec[6][j]; t.e4 = s_rec[7][j];
            t.e5 = s_rec[8][j]; t.e6 = s_rec[9][j]; t.e7 = s_rec[10][j]; t.e8 = s_rec[11][j]; t.area = s_rec[12][j];
            t.p0 = t.p1 = t.p3 = t.p4 = t.p6 = t.p7 = 0.0f;
            if (!(t.area > eps)) {
                // zero-area triangle (segment / point rules of pixel_weights read the screen-space vertices): rare,
                // so the record does not carry them — gathered again, same projection, same bits
                unsigned i0, i1, i2;
                Tri<float> q;
                load_tri32(q, vs, fs, (unsigned)s_box[3][j], nv, i0, i1, i2);
                screen3<PERSP>(q, (float)res, eps);
                t.p0 = q.p0; t.p1 = q.p1; t.p3 = q.p3; t.p4 = q.p4; t.p6 = q.p6; t.p7 = q.p7;
            }
            float c0, c1, c2, z;
            if (shade<float>(t, x, y, PERSP, eps, c0, c1, c2, z) && z == z) {      // NaN never passes `zB < z`
                const unsigned long long key =
                    ((unsigned long long)ord32(z) << 32) | (unsigned long long)(0xFFFFFFFEu - (unsigned)s_box[3][j]);
                unsigned long long* slot = &s_key[(y - oy) * TILE + (x - ox)];
                if (key > *slot) atomicMax(slot, key);
            }
        }
        __syncthreads();                                // records / prefix are rewritten by the next chunk
    }
    __syncthreads();
    // ---- resolve the tile from LDS: every pixel written once ----
    const size_t hw = (size_t)res * res;
    const long long shift = repeat_v ? 0 : (long long)nv * s;
    for (int p = threadIdx.x; p < TILE * TILE; p += 256) {
        const int lx = p & (TILE - 1), ly = p / TILE;
        const int x = ox + lx, y = oy + ly;
        if (x >= res || y >= res) continue;
        const unsigned long long key = s_key[p];
        int ti = -1;
        if (key != key_init_f32()) ti = (int)(0xFFFFFFFEu - (unsigned)(key & 0xFFFFFFFFull));
        unsigned i0 = 0, i1 = 0, i2 = 0;
        float c0 = 0, c1 = 0, c2 = 0, z = Lim<float>::lowest();
        long long shf = 0;
        if (ti >= 0) {
            Tri<float> t;
            load_tri32(t, vs, fs, (unsigned)ti, nv, i0, i1, i2);
            tri_setup_accepted<PERSP>(t, res, eps, false);
            shade<float>(t, x, y, PERSP, eps, c0, c1, c2, z);
            shf = shift;
        }
        const size_t g = (size_t)s * hw + (size_t)y * res + x;
        if (index) {
            index[3 * g] = (long long)i0 + shf;
            index[3 * g + 1] = (long long)i1 + shf;
            index[3 * g + 2] = (long long)i2 + shf;
        }
        if (coeff) {
            coeff[3 * g] = c0;
            coeff[3 * g + 1] = c1;
            coeff[3 * g + 2] = c2;
        }
        if (zbuf) zbuf[g] = z;
        if (win) win[g] = ti;
        if (first && ti >= 0) {
            // leader table of the gradient pass (k_first_pix's job, done here where the tile's winners sit in LDS):
            // first[sample, triangle] = smallest row-major pixel the triangle won.  A pixel whose left or upper
            // neighbour INSIDE the tile has the same winner cannot be the minimum and skips the atomic; tile-border
            // pixels always try (integer minimum: idempotent and order independent).
            const unsigned mine = (unsigned)(key & 0xFFFFFFFFull);
            const bool left = lx > 0 && (unsigned)(s_key[p - 1] & 0xFFFFFFFFull) == mine &&
                              s_key[p - 1] != key_init_f32();
            const bool up = ly > 0 && (unsigned)(s_key[p - TILE] & 0xFFFFFFFFull) == mine &&
                            s_key[p - TILE] != key_init_f32();
            if (!left && !up) atomicMin(&first[(size_t)s * nf + ti], y * res + x);
        }
        if (attr) {
            const size_t r0 = (size_t)((long long)i0 + shf) * tex_c, r1 = (size_t)((long long)i1 + shf) * tex_c,
                         r2 = (size_t)((long long)i2 + shf) * tex_c;
            for (int ch = 0; ch < tex_c; ++ch) {
                const float a0 = tex[r0 + ch] * c0;
                const float a1 = tex[r1 + ch] * c1;
                const float a2 = tex[r2 + ch] * c2;
                const float s01 = a0 + a1;
                attr[chw ? ((size_t)s * tex_c + ch) * hw + (size_t)y * res + x : g * tex_c + ch] = s01 + a2;
            }
        }
    }
}

// scratch layout of the tiled path (all unsigned): tile_cnt [b * ntile] | wide_cnt [b] | tile_list [b * ntile *
// TILE_CAP] | wide_list [b * nf]
inline long long tile_scratch_words(long long b, long long nf, long long hres) {
    const long long ntx = sr_ceil_div(hres, TILE), ntile = ntx * ntx;
    return b * ntile + b + b * ntile * TILE_CAP + b * nf;
}

// d(3 weights)/d(3 vertices x xyz) at one pixel (reference op/rasterize.h:169-228).  `g` = 27 values
// [weight][vertex][component]; returns false (g untouched) for a degenerate triangle.
template <typename R>
SR_HD bool weight_jacobian(const R p[9], R px, R py, R sw, R sh, R g[27],
                                                bool perspective, R eps) {
    const R u = (px * 2 - sw + 1) / sw;
    const R vv = (py * -2 + sh - 1) / sh;
    R e[9], det;
    R m0 = p[3] * p[7], m1 = p[4] * p[6];
    e[0] = m0 - m1;
    m0 = p[1] * p[6]; m1 = p[0] * p[7];
    e[1] = m0 - m1;
    m0 = p[0] * p[4]; m1 = p[1] * p[3];
    e[2] = m0 - m1;
    if (perspective) {
        if (p[2] >= -eps || p[5] >= -eps || p[8] >= -eps) return false;
        const R a = e[0] * p[2], b = e[1] * p[5], d = e[2] * p[8];
        det = a + b;
        det = det + d;
        if (det >= -eps && det <= eps) return false;
        e[0] = -e[0];
        e[1] = -e[1];
        e[2] = -e[2];
        m0 = p[4] * p[8]; m1 = p[5] * p[7]; e[3] = m0 - m1;
        m0 = p[2] * p[7]; m1 = p[1] * p[8]; e[4] = m0 - m1;
        m0 = p[1] * p[5]; m1 = p[2] * p[4]; e[5] = m0 - m1;
        m0 = p[5] * p[6]; m1 = p[3] * p[8]; e[6] = m0 - m1;
        m0 = p[0] * p[8]; m1 = p[2] * p[6]; e[7] = m0 - m1;
        m0 = p[2] * p[3]; m1 = p[0] * p[5]; e[8] = m0 - m1;
    } else {
        det = e[0] + e[1];
        det = det + e[2];
        e[3] = p[4] - p[7];
        e[4] = p[7] - p[1];
        e[5] = p[1] - p[4];
        e[6] = p[6] - p[3];
        e[7] = p[0] - p[6];
        e[8] = p[3] - p[0];
    }
    if (!(det < -eps || det > eps)) return false;
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = e[k] / det;
    R c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const R gx = e[3 + k] * u, gy = e[6 + k] * vv;
        const R a = e[k] + gx;
        c[k] = a + gy;
    }
#pragma unroll
    for (int l = 0; l < 27; ++l) {
        const int wi = l / 9, comp = (l + 1) % 3, vert = (l / 3) % 3;
        g[l] = -c[vert] * e[wi + comp * 3];
    }
    if (perspective) {
        R s = c[0] + c[1];
        s = s + c[2];
#pragma unroll
        for (int l = 0; l < 9; ++l) {
            R tot = g[l] + g[l + 9];
            tot = tot + g[l + 18];
#pragma unroll
            for (int wi = 0; wi < 3; ++wi) {
                const R corr = c[wi] * tot / s;
                if (l % 3 == 2) g[l + wi * 9] = (-g[l + wi * 9] - corr) / s;
                else g[l + wi * 9] = (g[l + wi * 9] - corr) / s;
            }
        }
    } else {
#pragma unroll
        for (int l = 0; l < 9; ++l) g[2 + l * 3] = 0;
    }
    return true;
}

template <typename R>
SR_HD bool load_pixel_tri(const long long* __restrict__ index, long long g,
                                               long long rows, const R* __restrict__ v, R p[9],
                                               long long ids[3]) {
    ids[0] = index[3 * g];
    ids[1] = index[3 * g + 1];
    ids[2] = index[3 * g + 2];
    if (ids[0] == ids[1] || ids[0] == ids[2] || ids[1] == ids[2]) return false;
    if (ids[0] < 0 || ids[1] < 0 || ids[2] < 0 || ids[0] >= rows || ids[1] >= rows || ids[2] >= rows)
        return false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p[k] = v[3 * ids[0] + k];
        p[3 + k] = v[3 * ids[1] + k];
        p[6 + k] = v[3 * ids[2] + k];
    }
    return true;
}

// dcoeff [b*h*w, 27]: 256 pixels per workgroup, staged through LDS so the 27-float records leave
// as fully coalesced rows.
template <typename R>
__global__ __launch_bounds__(256) void k_dcoeff(long long b, long long n, long long h, long long w,
                                                bool perspective, const R* __restrict__ v,
                                                const long long* __restrict__ index,
                                                R* __restrict__ dcoeff, R eps) {
    __shared__ R s_g[256 * 27];
    const long long hw = h * w, total = b * hw;
    const long long base = (long long)blockIdx.x * 256;
    const long long g = base + threadIdx.x;
    R jac[27];
#pragma unroll
    for (int l = 0; l < 27; ++l) jac[l] = 0;
    if (g < total) {
        R p[9];
        long long ids[3];
        if (load_pixel_tri<R>(index, g, n * b, v, p, ids)) {
            const long long pix = g % hw;
            weight_jacobian<R>(p, (R)(pix % w), (R)(pix / w), (R)h, (R)w, jac, perspective, eps);
        }
    }
#pragma unroll
    for (int l = 0; l < 27; ++l) s_g[threadIdx.x * 27 + l] = jac[l];
    __syncthreads();
    const long long remain = total - base;
    const int cnt = (int)((remain < 256 ? remain : 256) * 27);
    for (int i = threadIdx.x; i < cnt; i += 256) dcoeff[base * 27 + i] = s_g[i];
}

// ---- fused gradient, phase 1: per-triangle sums over the pixels the triangle won ----------------------------
// Scratch layout: one 16-byte aligned row of RS floats per (sample, triangle), written whole with float4 stores
// (scattered 4-byte stores cost one L2 transaction each: measured 130 of 217 us):
//   corner k at [k * (NVC + CT)]:  NVC vertex-gradient values (x, y[, z]), then CT attribute-gradient values;
// NVC = 2 orthographic (d/dz is identically zero there), 3 perspective.  A byte per row marks the written ones.
//
// Per-triangle constants of d(weights)/d(vertices) (reference barycentric_grad, op/rasterize.h:169-228).
// Orthographic: dcoeff[wi][vert][comp] = -c[vert] * E[wi + 3*((comp+1)%3)] with the z column zeroed, so the
// 27-entry Jacobian collapses to   g_vert.x = -c[vert] * sum_wi d_wi E[3+wi],  g_vert.y = -c[vert] * sum_wi d_wi E[6+wi]
// (the same products re-associated): 9 triangle constants, ~20 flops per pixel, no 27-register array.
template <typename R, bool PERSP>
struct JacTri;

template <typename R>
struct JacTri<R, false> {
    static constexpr int NV = 6;           // accumulated vertex-gradient values: (vertex, x|y)
    R E0, E1, E2, E3, E4, E5, E6, E7, E8;
    bool ok;
    __device__ __forceinline__ void init(const R p[9], R eps) {
        R m0 = p[3] * p[7], m1 = p[4] * p[6];
        const R e0 = m0 - m1;
        m0 = p[1] * p[6]; m1 = p[0] * p[7];
        const R e1 = m0 - m1;
        m0 = p[0] * p[4]; m1 = p[1] * p[3];
        const R e2 = m0 - m1;
        R det = e0 + e1;
        det = det + e2;
        ok = det < -eps || det > eps;
        E0 = e0 / det; E1 = e1 / det; E2 = e2 / det;
        E3 = (p[4] - p[7]) / det; E4 = (p[7] - p[1]) / det; E5 = (p[1] - p[4]) / det;
        E6 = (p[6] - p[3]) / det; E7 = (p[0] - p[6]) / det; E8 = (p[3] - p[0]) / det;
    }
    __device__ __forceinline__ void add(R (&gv)[NV], R d0, R d1, R d2, R px, R py, R sw, R sh, R eps) const {
        const R u = (px * 2 - sw + 1) / sw;
        const R vv = (py * -2 + sh - 1) / sh;
        const R c0 = (E0 + E3 * u) + E6 * vv;
        const R c1 = (E1 + E4 * u) + E7 * vv;
        const R c2 = (E2 + E5 * u) + E8 * vv;
        const R qx = (d0 * E3 + d1 * E4) + d2 * E5;
        const R qy = (d0 * E6 + d1 * E7) + d2 * E8;
        gv[0] -= c0 * qx; gv[1] -= c0 * qy;
        gv[2] -= c1 * qx; gv[3] -= c1 * qy;
        gv[4] -= c2 * qx; gv[5] -= c2 * qy;
    }
    // value j of vertex k (0: x, 1: y, 2: z)
    static __device__ __forceinline__ R get(const R (&gv)[NV], int k, int j) { return j == 2 ? (R)0 : gv[2 * k + j]; }
};

template <typename R>
struct JacTri<R, true> {
    static constexpr int NV = 9;
    R p[9];
    bool ok;
    __device__ __forceinline__ void init(const R q[9], R) {
#pragma unroll
        for (int i = 0; i < 9; ++i) p[i] = q[i];
        ok = true;
    }
    __device__ __forceinline__ void add(R (&gv)[NV], R d0, R d1, R d2, R px, R py, R sw, R sh, R eps) const {
        R jac[27];
        if (!weight_jacobian<R>(p, px, py, sw, sh, jac, true, eps)) return;
#pragma unroll
        for (int j = 0; j < 9; ++j) gv[j] += d0 * jac[j] + d1 * jac[9 + j] + d2 * jac[18 + j];
    }
    static __device__ __forceinline__ R get(const R (&gv)[NV], int k, int j) { return gv[3 * k + j]; }
};

template <typename R, int CT, bool PERSP>
struct TriAcc {
    R gv[JacTri<R, PERSP>::NV];
    R gt[3 * CT];
    int n;
    __device__ __forceinline__ void clear() {
        n = 0;
#pragma unroll
        for (int j = 0; j < JacTri<R, PERSP>::NV; ++j) gv[j] = 0;
#pragma unroll
        for (int j = 0; j < 3 * CT; ++j) gt[j] = 0;
    }
};

// One pixel of triangle `ti` (skipped unless the triangle won it).
template <typename R, int CT, bool PERSP>
__device__ __forceinline__ void grad_pixel(TriAcc<R, CT, PERSP>& acc, const Tri<R>& t, const JacTri<R, PERSP>& jt,
                                           bool want_v, int x, int y, long long w, long long hw, long long h_arg,
                                           R eps, const int* __restrict__ wins, int ti,
                                           const R* __restrict__ gos, const R* __restrict__ tex0,
                                           const R* __restrict__ tex1, const R* __restrict__ tex2,
                                           int tex_c, int ch0, long long gcs) {
    const long long pix = x + (long long)y * w;
    if (pix >= hw || wins[pix] != ti) return;
    R c0, c1, c2, z;
    shade<R>(t, x, y, PERSP, eps, c0, c1, c2, z);        // same bits as k_resolve used for the interpolation
    // gcs: channel stride of grad_out — 1 for [b,h,w,c] (pixel stride tex_c), h*w for [b,c,h,w] (pixel stride 1)
    const R* go = gos + pix * (gcs == 1 ? tex_c : 1);
    acc.n += 1;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        if (ch0 + j < tex_c) {
            const R gch = go[(ch0 + j) * gcs];
            acc.gt[j] += gch * c0;
            acc.gt[CT + j] += gch * c1;
            acc.gt[2 * CT + j] += gch * c2;
        }
    }
    if (want_v) {
        R d0 = 0, d1 = 0, d2 = 0;
        for (int ch = 0; ch < tex_c; ++ch) {
            const R gch = go[ch * gcs];
            d0 += gch * tex0[ch];
            d1 += gch * tex1[ch];
            d2 += gch * tex2[ch];
        }
        jt.add(acc.gv, d0, d1, d2, (R)x, (R)y, (R)h_arg, (R)w, eps);
    }
}

// The same pixel with the triangle's three attribute rows already in registers (tex_c <= 4: the normal maps of
// the generator): k_grad_pix fetches them together with the vertex positions, one round trip earlier, and once per
// triangle instead of once per pixel.  Same operations in the same order as grad_pixel.
template <typename R, int CT, bool PERSP>
__device__ __forceinline__ void grad_pixel_regs(TriAcc<R, CT, PERSP>& acc, const Tri<R>& t, const JacTri<R, PERSP>& jt,
                                                bool want_v, int x, int y, long long w, long long hw, long long h_arg,
                                                R eps, const int* __restrict__ wins, int ti,
                                                const R* __restrict__ gos, const R (&tx)[3][4], int tex_c, int ch0,
                                                long long gcs) {
    // (the caller walks the pixels of its winner mask: this pixel IS inside the image and won by `ti`)
    const long long pix = x + (long long)y * w;
    R c0, c1, c2, z;
    shade<R>(t, x, y, PERSP, eps, c0, c1, c2, z);
    const R* go = gos + pix * (gcs == 1 ? tex_c : 1);
    R gch[4];
    if (tex_c == 3 && gcs == 1) {
        load3<R>(go, gch[0], gch[1], gch[2]);
        gch[3] = (R)0;
    } else {
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) gch[ch] = ch < tex_c ? go[ch * gcs] : (R)0;
    }
    acc.n += 1;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        if (j < tex_c) {                                 // (tex_c <= 4: one channel chunk, ch0 == 0)
            acc.gt[j] += gch[j] * c0;
            acc.gt[CT + j] += gch[j] * c1;
            acc.gt[2 * CT + j] += gch[j] * c2;
        }
    }
    if (want_v) {
        R d0 = 0, d1 = 0, d2 = 0;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if (ch < tex_c) {
                d0 += gch[ch] * tx[0][ch];
                d1 += gch[ch] * tx[1][ch];
                d2 += gch[ch] * tx[2][ch];
            }
        }
        jt.add(acc.gv, d0, d1, d2, (R)x, (R)y, (R)h_arg, (R)w, eps);
    }
}

__device__ __forceinline__ float sr_shfl_down(float x, int off) { return __shfl_down(x, off, SR_WAVE); }
__device__ __forceinline__ double sr_shfl_down(double x, int off) { return __shfl_down(x, off, SR_WAVE); }

// Fixed-order sum over the 256 lanes of the workgroup (wave tree, then waves 0..3 in order); same value on all lanes.
template <typename R>
__device__ __forceinline__ R block_sum_256(R x, R* s_part) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += sr_shfl_down(x, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = x;
    __syncthreads();
    R tot = s_part[0];
    tot += s_part[1];
    tot += s_part[2];
    tot += s_part[3];
    return tot;
}

// One record per triangle: for each corner k the first four of its CORNER values (d/d(vertex) then d/d(attribute row))
// as an aligned quad at 4 k, the remaining CORNER - 4 behind the three quads.  k_grad_vert, which is bound by the number
// of scattered load instructions it issues (every lane of a wave hits a different line), fetches a corner with one
// 16-byte load + one scalar instead of five scalars.
template <int CT, bool PERSP>
struct RowShape {
    static constexpr int NVC = PERSP ? 3 : 2;
    static constexpr int CORNER = NVC + CT;
    static constexpr int TAIL = CORNER > 4 ? CORNER - 4 : 0;     // values of a corner behind its quad
    static constexpr int RS = (12 + 3 * TAIL + 3) / 4 * 4;       // values per record, multiple of 4
    static __host__ __device__ constexpr int at(int k, int j) { return j < 4 ? 4 * k + j : 12 + k * TAIL + (j - 4); }
};

template <typename R, int CT, bool PERSP>
__device__ __forceinline__ void store_row(const TriAcc<R, CT, PERSP>& acc, R* __restrict__ tg,
                                          long long rowid) {
    using S = RowShape<CT, PERSP>;
    R vals[S::RS];
#pragma unroll
    for (int i = 0; i < S::RS; ++i) vals[i] = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int j = 0; j < S::NVC; ++j) vals[S::at(k, j)] = JacTri<R, PERSP>::get(acc.gv, k, j);
#pragma unroll
        for (int j = 0; j < CT; ++j) vals[S::at(k, S::NVC + j)] = acc.gt[k * CT + j];
    }
    R* row = tg + rowid * S::RS;
    if (sizeof(R) == 4) {
        float4* dst = reinterpret_cast<float4*>(row);
#pragma unroll
        for (int i = 0; i < S::RS / 4; ++i)
            dst[i] = make_float4((float)vals[4 * i], (float)vals[4 * i + 1], (float)vals[4 * i + 2], (float)vals[4 * i + 3]);
    } else {
        double2* dst = reinterpret_cast<double2*>(row);
#pragma unroll
        for (int i = 0; i < S::RS / 2; ++i) dst[i] = make_double2((double)vals[2 * i], (double)vals[2 * i + 1]);
    }
}

// VERTEX-MAJOR corner slots (round 6).  The incidence list of a topology is sorted by vertex, so position e of the list
// (0 <= e < 3 nf) is a slot that belongs to ONE vertex, and the slots of a vertex — and of consecutive vertices — are
// contiguous.  Phase 1 writes the CORNER values of corner k of triangle f straight into slot e = slot_of[k * nf + f]
// (the inverse permutation of the list) instead of a 64-byte record per triangle; phase 2 then reads one contiguous span
// per vertex, consecutive lanes consecutive spans, instead of chasing six scattered records: same values, same summation
// order, a coalesced stream in place of a divergent gather.  Layout: quad plane Q[b * 3 nf] (the first four values of a
// corner, one 16-byte store / load) and TAIL scalar planes T[j][b * 3 nf] for the values behind it.
template <typename R, int CT, bool PERSP>
__device__ __forceinline__ void store_slots(const TriAcc<R, CT, PERSP>& acc, R* __restrict__ quad, R* __restrict__ tail,
                                            long long plane, long long slot_base, const int* __restrict__ slot_of,
                                            long long nf, long long ti) {
    using S = RowShape<CT, PERSP>;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        R vals[S::CORNER < 4 ? 4 : S::CORNER];
#pragma unroll
        for (int j = 0; j < 4; ++j) vals[j] = 0;
#pragma unroll
        for (int j = 0; j < S::NVC; ++j) vals[j] = JacTri<R, PERSP>::get(acc.gv, k, j);
#pragma unroll
        for (int j = 0; j < CT; ++j) vals[S::NVC + j] = acc.gt[k * CT + j];
        const long long e = slot_base + slot_of[k * nf + ti];
        if (sizeof(R) == 4) {
            reinterpret_cast<float4*>(quad)[e] = make_float4((float)vals[0], (float)vals[1], (float)vals[2], (float)vals[3]);
        } else {
            double2* dst = reinterpret_cast<double2*>(quad) + 2 * e;
            dst[0] = make_double2((double)vals[0], (double)vals[1]);
            dst[1] = make_double2((double)vals[2], (double)vals[3]);
        }
#pragma unroll
        for (int j = 0; j < S::TAIL; ++j) tail[j * plane + e] = vals[4 + j];
    }
}

// slot_of[adj[e]] = e: the inverse of a topology's incidence list (one int per corner; built per call when the caller
// does not hand a cached table over)
__global__ __launch_bounds__(256) void k_slot_inverse(long long n3, const int* __restrict__ adj, int* __restrict__ slot_of,
                                                      long long adj_bstride) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n3) return;
    const long long s = blockIdx.y;
    slot_of[s * n3 + adj[s * adj_bstride + e]] = (int)e;
}

// Large triangles (the list k_depth_keys recorded): one workgroup per triangle walks the box together; each lane
// sums its pixels in order, then a fixed-order tree over the 256 lanes.
template <typename R, int CT, bool PERSP>
__device__ __forceinline__ void grad_big_body(long long nv, long long nf, long long h, long long w, bool repeat_f,
                                              const R* __restrict__ v, const R* __restrict__ tex, int tex_c, int ch0,
                                              const long long* __restrict__ f, const int* __restrict__ win,
                                              const int* __restrict__ big, const R* __restrict__ grad_out, bool want_v,
                                              R* __restrict__ tg, R eps, bool chw, const int* __restrict__ slot_of,
                                              long long slot_bs, R* __restrict__ tail, long long plane) {
    __shared__ R s_part[4];
    const long long hw = h * w;
    const int count = big[0];
    for (int q = blockIdx.x; q < count; q += gridDim.x) {
        const long long g = big[1 + q];
        const long long s = g / nf, ti = g - s * nf;
        const R* vs = v + s * nv * 3;
        const long long* fs = repeat_f ? f : f + s * nf * 3;
        Tri<R> t;
        long long i0, i1, i2;
        load_tri<R>(t, vs, fs, ti, nv, i0, i1, i2);
        const R praw[9] = {t.p0, t.p1, t.p2, t.p3, t.p4, t.p5, t.p6, t.p7, t.p8};
        JacTri<R, PERSP> jt;
        jt.init(praw, eps);
        tri_setup<R>(t, h, w, PERSP, eps);
        TriAcc<R, CT, PERSP> acc;
        acc.clear();
        const bool distinct = want_v && jt.ok && i0 != i1 && i0 != i2 && i1 != i2;
        const R* tb = tex + s * nv * tex_c;
        const int bw = t.x1 - t.x0 + 1;
        const long long npx = (long long)bw * (t.y1 - t.y0 + 1);
        for (long long p = threadIdx.x; p < npx; p += 256) {
            const int yy = (int)(p / bw);
            grad_pixel<R, CT, PERSP>(acc, t, jt, distinct, t.x0 + (int)(p - (long long)yy * bw), t.y0 + yy, w, hw, h,
                                     eps, win + s * hw, (int)ti, grad_out + s * hw * tex_c, tb + i0 * tex_c,
                                     tb + i1 * tex_c, tb + i2 * tex_c, tex_c, ch0, chw ? hw : 1);
        }
        const R cnt = block_sum_256<R>((R)acc.n, s_part);
#pragma unroll
        for (int j = 0; j < JacTri<R, PERSP>::NV; ++j) acc.gv[j] = block_sum_256<R>(acc.gv[j], s_part);
#pragma unroll
        for (int j = 0; j < 3 * CT; ++j) acc.gt[j] = block_sum_256<R>(acc.gt[j], s_part);
        if (threadIdx.x == 0 && cnt > 0) {
            if (slot_of) store_slots<R, CT, PERSP>(acc, tg, tail, plane, s * 3 * nf, slot_of + s * slot_bs, nf, ti);
            else store_row<R, CT, PERSP>(acc, tg, g);
        }
    }
}

template <typename R, int CT, bool PERSP>
__global__ __launch_bounds__(256) void k_grad_big(long long nv, long long nf, long long h, long long w,
                                                  bool repeat_f, const R* __restrict__ v,
                                                  const R* __restrict__ tex, int tex_c, int ch0,
                                                  const long long* __restrict__ f, const int* __restrict__ win,
                                                  const int* __restrict__ big, const R* __restrict__ grad_out,
                                                  bool want_v, R* __restrict__ tg,
                                                  R eps, bool chw, const int* __restrict__ slot_of, long long slot_bs,
                                                  R* __restrict__ tail, long long plane) {
    grad_big_body<R, CT, PERSP>(nv, nf, h, w, repeat_f, v, tex, tex_c, ch0, f, win, big, grad_out, want_v, tg, eps, chw,
                                slot_of, slot_bs, tail, plane);
}

// first[s * nf + t] = the smallest row-major pixel index triangle t of sample s won (INT_MAX: none).  Row-major order
// over the image restricted to a triangle's box IS box order, so that pixel is the triangle's leader.
// A pixel whose left or upper neighbour was won by the same triangle cannot be the minimum and skips the atomic
// (most of a triangle's pixels: the table sees little more than one update per visible triangle).
// `built` (the state word the forward pass left behind the table): non-zero = the tiled forward has already built
// the table, every workgroup leaves at once.  The decision is the FORWARD's record, not a predicate re-evaluated at
// backward time (environment flips, a different heuristic input: the table would be read uninitialised).
__device__ __forceinline__ void first_pix_body(long long total, long long hw, long long w, long long nf,
                                               const int* __restrict__ win, int* __restrict__ first,
                                               const int* __restrict__ built) {
    if (*built) return;
    // grid-stride: the launch is capped at a few thousand workgroups, so that the common case (table already built by
    // the tiled forward: every workgroup leaves at the line above) costs ~1 us instead of retiring b*h*w/256 of them
    const long long stride = (long long)gridDim.x * 256;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
        const int ti = win[g];
        if (ti < 0) continue;
        const long long s = g / hw, pix = g - s * hw;
        const long long x = pix % w;
        if (x > 0 && win[g - 1] == ti) continue;
        if (pix >= w && win[g - w] == ti) continue;
        atomicMin(&first[s * nf + ti], (int)pix);                // integer minimum: order independent
    }
}

__global__ __launch_bounds__(256) void k_first_pix(long long total, long long hw, long long w, long long nf,
                                                   const int* __restrict__ win, int* __restrict__ first,
                                                   const int* __restrict__ built) {
    first_pix_body(total, hw, w, nf, win, first, built);
}

// Small triangles: one lane per (sample, TRIANGLE).  The leader table says in ONE coalesced load whether the triangle
// won a pixel at all (culled, hidden and sub-pixel triangles leave at once) and where its first pixel is; the lane
// then sums the triangle's pixels in box order and writes the records.  Consecutive lanes own consecutive triangles:
// the index loads and the row stores are coalesced, and because visibility is spatially coherent on a mesh the waves are
// mostly dense or empty.  (History at config[3]: one lane per pixel with a 16-probe leader test, 202 us; leaders
// compacted per workgroup through LDS, 122-129 us; a dense leader list, 86 us + 600 us for its single append
// counter; this form needs neither list nor counter.)
template <typename R, int CT, bool PERSP>
__device__ __forceinline__ void grad_pix_body(long long b, long long nv, long long nf, long long h, long long w,
                                              bool repeat_f, const R* __restrict__ v, const R* __restrict__ tex, int tex_c,
                                              int ch0, const long long* __restrict__ f, const int* __restrict__ win,
                                              const int* __restrict__ first, unsigned long long* __restrict__ valid,
                                              const R* __restrict__ grad_out, bool want_v, R* __restrict__ tg, R eps,
                                              bool chw, const int* __restrict__ slot_of, long long slot_bs,
                                              R* __restrict__ tail, long long plane) {
    __shared__ int s_row[256], s_pix[256];
    __shared__ int s_cnt[4];
    const long long hw = h * w;
    const long long row0 = (long long)blockIdx.x * 256;
    int lead_pix;
    long long row;
    {
        const long long rq = row0 + threadIdx.x;
        const int pq = rq < b * nf ? first[rq] : 0x7FFFFFFF;
        // one bit per record, "this triangle won a pixel: its record is written" (by this kernel or by k_grad_big) —
        // what k_grad_vert tests before it reads a record: a wave's 64 consecutive rows are one word
        const unsigned long long won = __ballot(pq != 0x7FFFFFFF);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0 && rq < b * nf) valid[rq >> 6] = won;                    // (waves beyond the last row own no word)
        // the winning triangles of the workgroup's 256 rows, compacted: the long part below runs in dense waves
        if (lane == 0) s_cnt[wv] = __popcll(won);
        __syncthreads();
        int base = 0;
        for (int i = 0; i < wv; ++i) base += s_cnt[i];
        const int n_won = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (pq != 0x7FFFFFFF) {
            const int slot = base + __popcll(won & ((1ull << lane) - 1ull));
            s_row[slot] = threadIdx.x;
            s_pix[slot] = pq;
        }
        __syncthreads();
        if ((int)threadIdx.x >= n_won) return;
        row = row0 + s_row[threadIdx.x];
        lead_pix = s_pix[threadIdx.x];
    }
    const long long s = row / nf;
    const int ti = (int)(row - s * nf);
    const long long pix = lead_pix;
    const int py = (int)(pix / w), px = (int)(pix - (long long)py * w);
    const R* vs = v + s * nv * 3;
    const long long* fs = repeat_f ? f : f + s * nf * 3;
    Tri<R> t;
    long long i0, i1, i2;
    load_tri<R>(t, vs, fs, ti, nv, i0, i1, i2);
    // attribute rows of the three corners: in flight together with the vertex positions (up to four channels)
    const R* tb = tex + s * nv * tex_c;
    const bool tex_regs = tex_c <= 4;
    R tx[3][4];
    if (tex_c == 3) {                               // the normal maps of the generator: one instruction per corner
        load3<R>(tb + i0 * 3, tx[0][0], tx[0][1], tx[0][2]);
        load3<R>(tb + i1 * 3, tx[1][0], tx[1][1], tx[1][2]);
        load3<R>(tb + i2 * 3, tx[2][0], tx[2][1], tx[2][2]);
        tx[0][3] = tx[1][3] = tx[2][3] = (R)0;
    } else {
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const bool have = tex_regs && ch < tex_c;
            tx[0][ch] = have ? tb[i0 * tex_c + ch] : (R)0;
            tx[1][ch] = have ? tb[i1 * tex_c + ch] : (R)0;
            tx[2][ch] = have ? tb[i2 * tex_c + ch] : (R)0;
        }
    }
    const R praw[9] = {t.p0, t.p1, t.p2, t.p3, t.p4, t.p5, t.p6, t.p7, t.p8};
    if constexpr (sizeof(R) == 4) {
        // a triangle that won a pixel was accepted by the forward pass: on a square image the short setup (no
        // rejection tests, no 64-bit conversion emulation) gives the same box and the same edge constants
        if (h == w && h < 0x40000000LL) tri_setup_accepted<PERSP>(t, (int)h, eps, true);
        else tri_setup<R>(t, h, w, PERSP, eps);
    } else {
        tri_setup<R>(t, h, w, PERSP, eps);
    }
    if ((long long)(t.x1 - t.x0 + 1) * (t.y1 - t.y0 + 1) > BIG_BOX) return;      // k_grad_big owns it
    const int* wins = win + s * hw;
    // which pixels of the box did this triangle win?  All winner-map loads are issued together (a loop that
    // leaves at the first hit would serialise one memory round trip per pixel).  Bit index = box order.
    const int bw = t.x1 - t.x0 + 1, bh = t.y1 - t.y0 + 1;
    unsigned long long mask = 0;
    if (bw <= 4 && bh <= 4) {
        int got[16];
        if (t.x0 + 3 < w && t.x0 + 3 + (long long)(t.y0 + 3) * w < hw) {
            // a row of the box = four consecutive ints inside the image row: one 16-byte load per row (4-byte aligned)
            typedef int i4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
            for (int yy = 0; yy < 4; ++yy) {
                const i4u r = *reinterpret_cast<const i4u*>(wins + t.x0 + (long long)(t.y0 + yy) * w);
                got[4 * yy + 0] = (0 < bw && yy < bh) ? r.x : -1;
                got[4 * yy + 1] = (1 < bw && yy < bh) ? r.y : -1;
                got[4 * yy + 2] = (2 < bw && yy < bh) ? r.z : -1;
                got[4 * yy + 3] = (3 < bw && yy < bh) ? r.w : -1;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int xx = i & 3, yy = i >> 2;
                const long long q = (t.x0 + xx) + (long long)(t.y0 + yy) * w;
                got[i] = (xx < bw && yy < bh && q < hw) ? wins[q] : -1;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (got[i] == ti) mask |= 1ull << ((i >> 2) * bw + (i & 3));
    } else {
        int p = 0;
        for (int y = t.y0; y <= t.y1; ++y)
            for (int x = t.x0; x <= t.x1; ++x, ++p) {
                const long long q = x + (long long)y * w;
                if (q < hw && wins[q] == ti) mask |= 1ull << p;
            }
    }
    const int self = (py - t.y0) * bw + (px - t.x0);
    if (mask & ((1ull << self) - 1ull)) return;          // an earlier pixel of the box leads this triangle
    JacTri<R, PERSP> jt;
    jt.init(praw, eps);
    TriAcc<R, CT, PERSP> acc;
    acc.clear();
    const bool distinct = want_v && jt.ok && i0 != i1 && i0 != i2 && i1 != i2;
    while (mask) {                                        // ascending bit = box order
        const int p = __ffsll((long long)mask) - 1;
        mask &= mask - 1ull;
        const int yy = p / bw;
        if (tex_regs)
            grad_pixel_regs<R, CT, PERSP>(acc, t, jt, distinct, t.x0 + (p - yy * bw), t.y0 + yy, w, hw, h, eps, wins, ti,
                                          grad_out + s * hw * tex_c, tx, tex_c, ch0, chw ? hw : 1);
        else
            grad_pixel<R, CT, PERSP>(acc, t, jt, distinct, t.x0 + (p - yy * bw), t.y0 + yy, w, hw, h, eps, wins, ti,
                                     grad_out + s * hw * tex_c, tb + i0 * tex_c, tb + i1 * tex_c, tb + i2 * tex_c, tex_c,
                                     ch0, chw ? hw : 1);
    }
    if (slot_of) store_slots<R, CT, PERSP>(acc, tg, tail, plane, s * 3 * nf, slot_of + s * slot_bs, nf, ti);
    else store_row<R, CT, PERSP>(acc, tg, row);
}

template <typename R, int CT, bool PERSP>
__global__ __launch_bounds__(256) void k_grad_pix(long long b, long long nv, long long nf, long long h,
                                                  long long w, bool repeat_f, const R* __restrict__ v,
                                                  const R* __restrict__ tex, int tex_c, int ch0,
                                                  const long long* __restrict__ f, const int* __restrict__ win,
                                                  const int* __restrict__ first,
                                                  unsigned long long* __restrict__ valid,
                                                  const R* __restrict__ grad_out, bool want_v,
                                                  R* __restrict__ tg, R eps, bool chw, const int* __restrict__ slot_of,
                                                  long long slot_bs, R* __restrict__ tail, long long plane) {
    grad_pix_body<R, CT, PERSP>(b, nv, nf, h, w, repeat_f, v, tex, tex_c, ch0, f, win, first, valid, grad_out, want_v, tg,
                                eps, chw, slot_of, slot_bs, tail, plane);
}

// ---- fused gradient, phase 2: per-vertex sum over its incident corners, in incidence-list order ---------------
// One step of the gather: U list entries (already in registers; -1 = none) -> validity words -> records, every level's
// loads in flight together (unconditional loads from clamped addresses, masked afterwards); the sums keep list order.
template <typename R, int CT, bool PERSP, int U>
__device__ __forceinline__ void vert_gather(const int (&idx)[U], long long s, long long nf, const R* __restrict__ tg,
                                            const unsigned long long* __restrict__ valid, R (&av)[3], R (&at)[CT]) {
    using S = RowShape<CT, PERSP>;
    constexpr int NC = S::CORNER < 4 ? 4 : S::CORNER;
    int kk[U];
    long long row[U];
    unsigned long long word[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = idx[u] < 0 ? 0 : idx[u];
        kk[u] = i >= 2 * (int)nf ? 2 : (i >= (int)nf ? 1 : 0);
        row[u] = s * nf + (i - kk[u] * (int)nf);
        word[u] = valid[row[u] >> 6];
    }
    R c[U][NC];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        ok[u] = idx[u] >= 0 && ((word[u] >> (row[u] & 63)) & 1ull);      // won a pixel: the record exists
        const R* rec = tg + (ok[u] ? row[u] : s * nf) * S::RS;
        if (sizeof(R) == 4) {
            const float4 q = *reinterpret_cast<const float4*>(rec + 4 * kk[u]);
            c[u][0] = (R)q.x; c[u][1] = (R)q.y; c[u][2] = (R)q.z; c[u][3] = (R)q.w;
        } else {
            const double2 q0 = *reinterpret_cast<const double2*>(rec + 4 * kk[u]);
            const double2 q1 = *reinterpret_cast<const double2*>(rec + 4 * kk[u] + 2);
            c[u][0] = (R)q0.x; c[u][1] = (R)q0.y; c[u][2] = (R)q1.x; c[u][3] = (R)q1.y;
        }
#pragma unroll
        for (int j = 4; j < S::CORNER; ++j) c[u][j] = rec[12 + kk[u] * S::TAIL + (j - 4)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
#pragma unroll
        for (int j = 0; j < S::NVC; ++j) av[j] += c[u][j];
#pragma unroll
        for (int j = 0; j < CT; ++j) at[j] += c[u][S::NVC + j];
    }
}

// The same step over vertex-major slots: entry u of the step is list position e0 + u, i.e. slot e0 + u — the records of a
// lane are one contiguous span and the spans of a wave's lanes follow each other.
// EAGER: the slot loads do not wait for the validity words (their address is the list position, known as soon as the
// offsets are): three dependent memory levels instead of four, at the price of fetching the unwritten slots of hidden
// triangles too — taken for small batches, where the pass is bound by that chain and not by bytes (inversion: batch 1).
template <typename R, int CT, bool PERSP, int U, bool EAGER>
__device__ __forceinline__ void vert_gather_slots(const int (&idx)[U], long long e0, long long s, long long nf,
                                                  const R* __restrict__ quad, const R* __restrict__ tail, long long plane,
                                                  const unsigned long long* __restrict__ valid, R (&av)[3], R (&at)[CT]) {
    using S = RowShape<CT, PERSP>;
    constexpr int NC = S::CORNER < 4 ? 4 : S::CORNER;
    long long row[U];
    unsigned long long word[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = idx[u] < 0 ? 0 : idx[u];
        const int kk = i >= 2 * (int)nf ? 2 : (i >= (int)nf ? 1 : 0);
        row[u] = s * nf + (i - kk * (int)nf);
        word[u] = valid[row[u] >> 6];
    }
    R c[U][NC];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        ok[u] = idx[u] >= 0 && ((word[u] >> (row[u] & 63)) & 1ull);      // won a pixel: the slot was written
        const long long e = s * 3 * nf + (EAGER ? (idx[u] >= 0 ? e0 + u : e0) : (ok[u] ? e0 + u : 0));
        if (sizeof(R) == 4) {
            const float4 q = reinterpret_cast<const float4*>(quad)[e];
            c[u][0] = (R)q.x; c[u][1] = (R)q.y; c[u][2] = (R)q.z; c[u][3] = (R)q.w;
        } else {
            const double2 q0 = reinterpret_cast<const double2*>(quad)[2 * e];
            const double2 q1 = reinterpret_cast<const double2*>(quad)[2 * e + 1];
            c[u][0] = (R)q0.x; c[u][1] = (R)q0.y; c[u][2] = (R)q1.x; c[u][3] = (R)q1.y;
        }
#pragma unroll
        for (int j = 4; j < S::CORNER; ++j) c[u][j] = tail[(j - 4) * plane + e];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
#pragma unroll
        for (int j = 0; j < S::NVC; ++j) av[j] += c[u][j];
#pragma unroll
        for (int j = 0; j < CT; ++j) at[j] += c[u][S::NVC + j];
    }
}

// 85 us at config[3] with one corner per step, 70 with six in flight.  (Round 4 tried the incidence list as fixed-width
// rows — one aligned 32-byte load per vertex instead of offsets -> entries, a dependent level less: no change, 69-74 us;
// the kernel is bound by the number of divergent L1 accesses, ~14 per lane at 48 % TCP utilisation, not by the chain.)
// The sums of one (sample, vertex) over its corner list [e0, e1) from ONE gradient state (tg / valid / tail): what
// k_grad_vert stores, and what k_grad_vert_levels adds up over the resolutions of a pyramid.
template <typename R, int CT, bool PERSP>
__device__ __forceinline__ void vert_sum(long long s, long long nf, const int* __restrict__ ad, int e0, int e1,
                                         const R* __restrict__ tg, const unsigned long long* __restrict__ valid, int slots,
                                         const R* __restrict__ tail, long long plane, R (&av)[3], R (&at)[CT]) {
    av[0] = av[1] = av[2] = 0;
#pragma unroll
    for (int j = 0; j < CT; ++j) at[j] = 0;
    constexpr int U = 6;                        // (the valence of an interior vertex of a triangulated grid)
    // A vertex of high valence (the two poles of the formula mesh: 192 corners; a fan anywhere) would be ONE lane walking
    // 32 dependent steps while its wave waits — at batch 1 those two lanes were the whole duration of the kernel (38 us for
    // 24 770 vertices).  Such a vertex is summed by its wave instead: lane l takes list positions e0 + l, e0 + l + 64, ...
    // in order, then a fixed-order tree over the lanes (deterministic; the order differs from the serial one by rounding).
    constexpr int WIDE = 4 * U;
    const bool wide = e1 - e0 > WIDE;
    if (!wide) {
        for (int base = e0; base < e1; base += U) {
            int idx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) idx[u] = base + u < e1 ? ad[base + u] : -1;
            if (slots == 2) vert_gather_slots<R, CT, PERSP, U, true>(idx, base, s, nf, tg, tail, plane, valid, av, at);
            else if (slots) vert_gather_slots<R, CT, PERSP, U, false>(idx, base, s, nf, tg, tail, plane, valid, av, at);
            else vert_gather<R, CT, PERSP, U>(idx, s, nf, tg, valid, av, at);
        }
    }
    unsigned long long wm = __ballot(wide);
    const int lane = threadIdx.x & 63;
    while (wm) {
        const int src = __ffsll((long long)wm) - 1;
        wm &= wm - 1ull;
        const int we0 = __shfl(e0, src, SR_WAVE), we1 = __shfl(e1, src, SR_WAVE);
        R pv[3] = {0, 0, 0};
        R pt[CT];
#pragma unroll
        for (int j = 0; j < CT; ++j) pt[j] = 0;
        for (int e = we0 + lane; e < we1; e += 64) {
            const int idx[1] = {ad[e]};
            if (slots) vert_gather_slots<R, CT, PERSP, 1, false>(idx, e, s, nf, tg, tail, plane, valid, pv, pt);
            else vert_gather<R, CT, PERSP, 1>(idx, s, nf, tg, valid, pv, pt);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j) pv[j] += sr_shfl_down(pv[j], o);
#pragma unroll
            for (int j = 0; j < CT; ++j) pt[j] += sr_shfl_down(pt[j], o);
        }
        // the sums sit in lane 0; hand them to the lane that owns the vertex
#pragma unroll
        for (int j = 0; j < 3; ++j) pv[j] = __shfl(pv[j], 0, SR_WAVE);
#pragma unroll
        for (int j = 0; j < CT; ++j) pt[j] = __shfl(pt[j], 0, SR_WAVE);
        if (lane == src) {
#pragma unroll
            for (int j = 0; j < 3; ++j) av[j] = pv[j];
#pragma unroll
            for (int j = 0; j < CT; ++j) at[j] = pt[j];
        }
    }
}

template <typename R, int CT, bool PERSP>
__global__ __launch_bounds__(256) void k_grad_vert(long long nv, long long nf, const int* __restrict__ adj_off,
                                                   const int* __restrict__ adj, long long off_bstride,
                                                   long long adj_bstride, const R* __restrict__ tg,
                                                   const unsigned long long* __restrict__ valid, int tex_c, int ch0,
                                                   R* __restrict__ grad_v, R* __restrict__ grad_tex, int slots,
                                                   const R* __restrict__ tail, long long plane, bool accumulate) {
    const long long vert = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long s = blockIdx.y;
    const bool live = vert < nv;
    R av[3];
    R at[CT];
    const int* off = adj_off + s * off_bstride;
    const int* ad = adj + s * adj_bstride;
    const int e0 = live ? off[vert] : 0, e1 = live ? off[vert + 1] : 0;
    vert_sum<R, CT, PERSP>(s, nf, ad, e0, e1, tg, valid, slots, tail, plane, av, at);
    if (!live) return;
    if (grad_v && ch0 == 0) {
        // accumulate (SR_RASTER_GRAD_ACC): the buffers hold the gradient of earlier calls — the same mesh rasterised at
        // several resolutions sums its gradients here, in call order, instead of in one tensor addition per call
        R* o = grad_v + (s * nv + vert) * 3;
        o[0] = accumulate ? o[0] + av[0] : av[0];
        o[1] = accumulate ? o[1] + av[1] : av[1];
        o[2] = accumulate ? o[2] + av[2] : av[2];
    }
    if (grad_tex) {
        R* o = grad_tex + (s * nv + vert) * tex_c + ch0;
#pragma unroll
        for (int j = 0; j < CT; ++j)
            if (ch0 + j < tex_c) o[j] = accumulate ? o[j] + at[j] : at[j];
    }
}

// ---- gradient of a pyramid (the same mesh at several resolutions, RasterLevels above) in four launches ------------------
// blockIdx.y (k_first_pix / k_grad_big / k_grad_pix) = level; k_grad_vert_levels sums the levels' per-vertex sums in table
// order — the value SR_RASTER_GRAD_ACC calls in that order produce, bit for bit (each level's sum is formed on its own,
// then added to the running total).  fp32, <= 4 attribute channels, vertex-major slots from the caller's inverse table.
struct GradLevels {
    int n;
    int res_h[RASTER_MAX_LEVELS], res_w[RASTER_MAX_LEVELS];
    const int* win[RASTER_MAX_LEVELS];
    int* big[RASTER_MAX_LEVELS];                  // [count | ids b*nf | leader table b*nf | state word]
    const float* grad_out[RASTER_MAX_LEVELS];
    float* tg[RASTER_MAX_LEVELS];                 // per-level scratch: corner slots (quad plane | tail planes) ...
    unsigned long long* valid[RASTER_MAX_LEVELS]; // ... and the valid words behind them
};

__global__ __launch_bounds__(256) void k_first_pix_levels(const GradLevels t, long long b, long long nf) {
    const int l = blockIdx.y;
    const long long hw = (long long)t.res_h[l] * t.res_w[l];
    int* first = t.big[l] + 1 + b * nf;
    first_pix_body(b * hw, hw, t.res_w[l], nf, t.win[l], first, first + b * nf);
}

template <int CT, bool PERSP>
__global__ __launch_bounds__(256) void k_grad_big_levels(const GradLevels t, long long b, long long nv, long long nf,
                                                         bool repeat_f, const float* __restrict__ v,
                                                         const float* __restrict__ tex, int tex_c,
                                                         const long long* __restrict__ f, bool want_v, float eps, bool chw,
                                                         const int* __restrict__ slot_of, long long slot_bs) {
    const int l = blockIdx.y;
    grad_big_body<float, CT, PERSP>(nv, nf, t.res_h[l], t.res_w[l], repeat_f, v, tex, tex_c, 0, f, t.win[l], t.big[l],
                                    t.grad_out[l], want_v, t.tg[l], eps, chw, slot_of, slot_bs, t.tg[l] + b * nf * 12,
                                    b * 3 * nf);
}

template <int CT, bool PERSP>
__global__ __launch_bounds__(256) void k_grad_pix_levels(const GradLevels t, long long b, long long nv, long long nf,
                                                         bool repeat_f, const float* __restrict__ v,
                                                         const float* __restrict__ tex, int tex_c,
                                                         const long long* __restrict__ f, bool want_v, float eps, bool chw,
                                                         const int* __restrict__ slot_of, long long slot_bs) {
    const int l = blockIdx.y;
    grad_pix_body<float, CT, PERSP>(b, nv, nf, t.res_h[l], t.res_w[l], repeat_f, v, tex, tex_c, 0, f, t.win[l],
                                    t.big[l] + 1 + b * nf, t.valid[l], t.grad_out[l], want_v, t.tg[l], eps, chw, slot_of,
                                    slot_bs, t.tg[l] + b * nf * 12, b * 3 * nf);
}

template <int CT, bool PERSP>
__global__ __launch_bounds__(256) void k_grad_vert_levels(const GradLevels t, long long b, long long nv, long long nf,
                                                          const int* __restrict__ adj_off, const int* __restrict__ adj,
                                                          long long off_bstride, long long adj_bstride, int tex_c,
                                                          float* __restrict__ grad_v, float* __restrict__ grad_tex,
                                                          int slots) {
    const long long vert = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long s = blockIdx.y;
    const bool live = vert < nv;
    const int* off = adj_off + s * off_bstride;
    const int* ad = adj + s * adj_bstride;
    const int e0 = live ? off[vert] : 0, e1 = live ? off[vert + 1] : 0;
    float tv[3] = {0.0f, 0.0f, 0.0f}, tt[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) tt[j] = 0.0f;
    for (int l = 0; l < t.n; ++l) {
        float av[3], at[CT];
        vert_sum<float, CT, PERSP>(s, nf, ad, e0, e1, t.tg[l], t.valid[l], slots, t.tg[l] + b * nf * 12, b * 3 * nf, av, at);
        if (l == 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j) tv[j] = av[j];
#pragma unroll
            for (int j = 0; j < CT; ++j) tt[j] = at[j];
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) tv[j] = tv[j] + av[j];
#pragma unroll
            for (int j = 0; j < CT; ++j) tt[j] = tt[j] + at[j];
        }
    }
    if (!live) return;
    if (grad_v) {
        float* o = grad_v + (s * nv + vert) * 3;
        o[0] = tv[0];
        o[1] = tv[1];
        o[2] = tv[2];
    }
    if (grad_tex) {
        float* o = grad_tex + (s * nv + vert) * tex_c;
#pragma unroll
        for (int j = 0; j < CT; ++j)
            if (j < tex_c) o[j] = tt[j];
    }
}

// The tiled path serves fp32 on square images (x and y ranges are swapped by the reference's call sites, SURVEY D8:
// on a non-square image pixel columns beyond the width alias into the next row, which only a global key buffer
// reproduces).  It pays when there are thousands of (sample, tile) workgroups with a few hundred triangles each —
// BASELINE config[3]: 64 samples x 64 tiles, ~400 accepted triangles per tile: 0.116 vs 0.129 ms.  With few tiles (low
// resolution under the same mesh: sub-pixel triangles that the global path rejects in a handful of instructions) or
// few samples (training: 4, inversion: 1) each workgroup walks a long list serially and the global-key path wins by
// 2-5x (scripts/raster_res_probe.py), so the choice is made per call.  SR_RASTER_TILED=1 forces the tiled path (tests),
// =0 the global-key path.
template <typename R>
inline bool tiled_ok(long long, long long, long long, long long) { return false; }
template <>
inline bool tiled_ok<float>(long long b, long long nf, long long h, long long w) {
    const char* e = getenv("SR_RASTER_TILED");               // read per call: tests flip it at run time
    const int mode = !e ? 2 : (e[0] == '0' ? 0 : 1);
    const long long ntx = sr_ceil_div(h, TILE), ntile = ntx * ntx;
    const bool possible = mode != 0 && h == w && nf > 0 && h < 0x40000000LL && b <= 65535 && b * ntile < 0x7FFFFFFFLL &&
                          b * nf < 0x7FFFFFFFLL;
    if (!possible) return false;
    return mode == 1 || (b * ntile >= 4096 && nf <= 1024 * ntile);
}

template <typename R>
int forward_tiled(long long, long long, long long, long long, int, int, int, const R*, const long long*, long long*, R*,
                  R*, R, const R*, long long, R*, int*, int*, bool, void*, hipStream_t) {
    return SR_EINVAL;
}
template <>
int forward_tiled<float>(long long b, long long nv, long long nf, long long hres, int repeat_v, int repeat_f,
                         int perspective, const float* v, const long long* tri, long long* index, float* coeff,
                         float* zbuf, float eps, const float* tex, long long tex_c, float* attr, int* win, int* big,
                         bool chw, void* work, hipStream_t st) {
    const int ntx = (int)sr_ceil_div(hres, TILE);
    const long long ntile = (long long)ntx * ntx;
    unsigned* tile_cnt = reinterpret_cast<unsigned*>(work);
    unsigned* wide_cnt = tile_cnt + b * ntile;
    unsigned* tile_list = wide_cnt + b;
    unsigned* wide_list = tile_list + b * ntile * TILE_CAP;
    // gradient state: big = [count | b * nf big-triangle ids | b * nf leader table | state word]
    int* first = (big && win) ? big + 1 + b * nf : nullptr;
    hipLaunchKernelGGL(k_tile_zero, dim3(sr_stream_grid(first ? b * nf : b * ntile + b, 256)), dim3(256), 0, st, tile_cnt,
                       b * ntile + b, big, first, b * nf);
    const dim3 bin_grid((unsigned)sr_ceil_div(nf, 256), (unsigned)b);
    const char* e_bin = getenv("SR_RASTER_BIN_LDS");                         // =0: the wave-vote binning (A/B, tests)
    const bool bin_lds = !(e_bin && e_bin[0] == '0');
#define SR_TILE_LAUNCH(P)                                                                                              \
    do {                                                                                                               \
        if (ntile <= BIN_LDS_TILES && bin_lds)                                                                         \
            hipLaunchKernelGGL((k_tile_bin_lds<P>), bin_grid, dim3(256), 0, st, (unsigned)nv, (unsigned)nf, (int)hres,  \
                               repeat_v != 0, repeat_f != 0, v, tri, tile_cnt, tile_list, wide_cnt, wide_list, big, ntx, \
                               eps);                                                                                   \
        else                                                                                                           \
            hipLaunchKernelGGL((k_tile_bin<P>), bin_grid, dim3(256), 0, st, (unsigned)nv, (unsigned)nf, (int)hres,      \
                               repeat_v != 0, repeat_f != 0, v, tri, tile_cnt, tile_list, wide_cnt, wide_list, big, ntx, \
                               eps);                                                                                   \
        hipLaunchKernelGGL((k_tile_raster<P>), dim3((unsigned)(b * ntile)), dim3(256), 0, st, (unsigned)b, (unsigned)nv, \
                           (unsigned)nf, (int)hres, repeat_v != 0, repeat_f != 0, v, tri, tile_cnt, tile_list, wide_cnt,  \
                           wide_list, ntx, index, coeff, zbuf, tex, (int)tex_c, attr, win, first, eps, chw);            \
    } while (0)
    if (perspective) SR_TILE_LAUNCH(true);
    else SR_TILE_LAUNCH(false);
#undef SR_TILE_LAUNCH
    return sr_launch_status();
}

template <typename R>
int forward_impl(long long b, long long nv, long long nf, long long h, long long w, int repeat_v,
                 int repeat_f, int perspective, const R* v, const long long* tri, long long* index,
                 R* coeff, R* zbuf, R eps, const R* tex, long long tex_c, R* attr, int* win, int* big,
                 void* work, hipStream_t st) {
    if (b < 0 || nv < 0 || nf < 0 || h <= 0 || w <= 0) return SR_EINVAL;
    if (nf >= 0xFFFFFFFELL || (win && nf >= 0x7FFFFFFFLL) || (big && b * nf >= 0x7FFFFFFFLL)) return SR_ERANGE;
    if (b == 0) return SR_OK;
    if (!work || (nf > 0 && (!v || !tri))) return SR_EINVAL;
    if (attr && (!tex || tex_c <= 0)) return SR_EINVAL;
    // `perspective` carries the flags of the call: bit 0 = perspective projection, bit 1 (SR_RASTER_CHW) = attribute
    // maps channel-major [b, c, h, w] instead of the reference's [b, h, w, c]
    const bool chw = (perspective & SR_RASTER_CHW) != 0;
    perspective &= 1;
    const long long npix = b * h * w;
    if (eps < 0) eps = -eps;
    if (tiled_ok<R>(b, nf, h, w)) return forward_tiled(b, nv, nf, h, repeat_v, repeat_f, perspective, v, tri, index, coeff,
                                                       zbuf, eps, tex, tex_c, attr, win, big, chw, work, st);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(work);
    unsigned* tmin = reinterpret_cast<unsigned*>(keys + npix);
    const bool is64 = sizeof(R) == 8;
    hipLaunchKernelGGL(k_fill_u64, dim3(sr_stream_grid(npix, 256)), dim3(256), 0, st, keys,
                       is64 ? key_init_f64() : key_init_f32(), npix, big, (big && win) ? big + 1 + b * nf : nullptr,
                       b * nf);
    if (is64)
        hipLaunchKernelGGL(k_fill_u32, dim3(sr_stream_grid(npix, 256)), dim3(256), 0, st, tmin,
                           0xFFFFFFFFu, npix);
    const long long ntri = b * nf;
    if (ntri > 0) {
        const unsigned grid = (unsigned)sr_ceil_div(ntri, 256);
        if (!is64) {
            hipLaunchKernelGGL((k_depth_keys<R, 0>), dim3(grid), dim3(256), 0, st, b, nv, nf, h, w,
                               repeat_v != 0, repeat_f != 0, perspective != 0, v, tri, keys, tmin, big, eps);
        } else {
            hipLaunchKernelGGL((k_depth_keys<R, 1>), dim3(grid), dim3(256), 0, st, b, nv, nf, h, w,
                               repeat_v != 0, repeat_f != 0, perspective != 0, v, tri, keys, tmin, big, eps);
            hipLaunchKernelGGL((k_depth_keys<R, 2>), dim3(grid), dim3(256), 0, st, b, nv, nf, h, w,
                               repeat_v != 0, repeat_f != 0, perspective != 0, v, tri, keys, tmin, big, eps);
        }
    }
    hipLaunchKernelGGL((k_resolve<R>), dim3((unsigned)sr_ceil_div(npix, 256)), dim3(256), 0, st, b, nv, nf,
                       h, w, repeat_v != 0, repeat_f != 0, perspective != 0, v, tri, keys, tmin, index,
                       coeff, zbuf, tex, tex_c, attr, win, eps, chw);
    return sr_launch_status();
}

template <typename R>
int backward_impl(long long b, long long n, long long h, long long w, int perspective, const R* v,
                  const long long* index, R* dcoeff, R eps, hipStream_t st) {
    if (b < 0 || n < 0 || h < 0 || w < 0) return SR_EINVAL;
    const long long total = b * h * w;
    if (total == 0) return SR_OK;
    if (!v || !index || !dcoeff) return SR_EINVAL;
    if (eps < 0) eps = -eps;
    hipLaunchKernelGGL((k_dcoeff<R>), dim3((unsigned)sr_ceil_div(total, 256)), dim3(256), 0, st, b, n, h,
                       w, perspective != 0, v, index, dcoeff, eps);
    return sr_launch_status();
}

// floats per scratch row for the widest chunk (<= 4 channels) and either projection: 3 * (3 + 4) rounded up to 4
inline long long grad_row_floats() { return 24; }

template <typename R, int CT, bool PERSP>
void grad_launch(long long b, long long nv, long long nf, long long h, long long w, bool repeat_f, const R* v,
                 const R* tex, int tex_c, int ch0, const long long* tri, const int* win, const int* big,
                 const R* grad_out, const int* adj_off, const int* adj, long long off_bs, long long adj_bs,
                 R* grad_v, R* grad_tex, R eps, R* tg, const int* first, unsigned long long* valid, bool chw,
                 const int* slot_of, long long slot_bs, R* tail, bool accumulate, hipStream_t st) {
    const bool want_v = grad_v != nullptr && ch0 == 0;
    const long long plane = b * 3 * nf;
    // few lanes: the gather is a latency chain, read the slots eagerly (SR_RASTER_GRAD_EAGER=0 / 1 forces)
    const char* ee = getenv("SR_RASTER_GRAD_EAGER");
    const bool eager = ee ? ee[0] == '1' : b * nv < 8 * 32768;
    hipLaunchKernelGGL((k_grad_big<R, CT, PERSP>), dim3(SR_NUM_CU * 2), dim3(256), 0, st, nv, nf, h, w, repeat_f, v,
                       tex, tex_c, ch0, tri, win, big, grad_out, want_v, tg, eps, chw, slot_of, slot_bs, tail, plane);
    if (b * nf > 0)
        hipLaunchKernelGGL((k_grad_pix<R, CT, PERSP>), dim3((unsigned)sr_ceil_div(b * nf, 256)), dim3(256), 0, st, b,
                           nv, nf, h, w, repeat_f, v, tex, tex_c, ch0, tri, win, first, valid, grad_out, want_v, tg, eps,
                           chw, slot_of, slot_bs, tail, plane);
    hipLaunchKernelGGL((k_grad_vert<R, CT, PERSP>), dim3((unsigned)sr_ceil_div(nv, 256), (unsigned)b), dim3(256), 0,
                       st, nv, nf, adj_off, adj, off_bs, adj_bs, tg, valid, tex_c, ch0, grad_v, grad_tex,
                       slot_of == nullptr ? 0 : (eager ? 2 : 1), tail, plane, accumulate);
}

template <typename R>
int grad_impl(long long b, long long nv, long long nf, long long h, long long w, int repeat_f, int perspective,
              const R* v, const R* tex, long long tex_c, const long long* tri, const int* win, const int* big,
              const R* grad_out, const int* adj_off, const int* adj, long long off_bs, long long adj_bs,
              const int* adj_slot, R* grad_v, R* grad_tex, R eps, void* work, hipStream_t st) {
    if (b < 0 || nv < 0 || nf < 0 || h < 0 || w < 0 || tex_c <= 0) return SR_EINVAL;
    const bool chw = (perspective & SR_RASTER_CHW) != 0;        // grad_out [b, c, h, w] (see forward_impl)
    const bool accumulate = (perspective & SR_RASTER_GRAD_ACC) != 0;
    perspective &= 1;
    if (b == 0 || nv == 0 || (!grad_v && !grad_tex)) return SR_OK;
    if (b > 65535 || nf >= 0x7FFFFFFFLL / 3 || tex_c > 0x7FFFFFFF) return SR_ERANGE;
    if (!v || !tex || !grad_out || !adj_off || !work || (nf > 0 && (!tri || !adj || !win || !big))) return SR_EINVAL;
    if (eps < 0) eps = -eps;
    if (h * w >= 0x7FFFFFFFLL) return SR_ERANGE;
    R* tg = reinterpret_cast<R*>(work);
    if (b * h * w >= 0x7FFFFFFFLL) return SR_ERANGE;
    // (records: 24 floats each, so the two tables behind them stay 8-byte aligned)
    unsigned long long* valid = reinterpret_cast<unsigned long long*>(tg + b * nf * grad_row_floats());
    // vertex-major slots (default; SR_RASTER_GRAD_SLOTS=0: the per-triangle records of rounds 2-5): the same 24 values per
    // triangle hold the quad plane (12) and up to three tail planes (3 each); the inverse incidence table sits behind the
    // valid words
    const char* slots_env = getenv("SR_RASTER_GRAD_SLOTS");
    const bool use_slots = nf > 0 && !(slots_env && slots_env[0] == '0');
    R* tail = tg + b * nf * 12;
    const int* slot_of = nullptr;
    long long slot_bs = 0;
    if (use_slots && adj_slot) {                               // the caller's cached inverse table
        slot_of = adj_slot;
        slot_bs = adj_bs;
    } else if (use_slots) {
        int* built = reinterpret_cast<int*>(valid + (b * nf + 63) / 64 + 1);
        const long long topo = adj_bs ? b : 1;                 // per-sample topologies carry a batch stride
        slot_bs = adj_bs ? 3 * nf : 0;
        hipLaunchKernelGGL(k_slot_inverse, dim3((unsigned)sr_ceil_div(3 * nf, 256), (unsigned)topo), dim3(256), 0, st, 3 * nf,
                           adj, built, adj_bs);
        slot_of = built;
    }
    // leaders of all triangles, once per call (the attribute-channel chunks below share them): the table behind the
    // big-triangle list of the forward call's gradient state.  The tiled forward has built it (state word 1); the
    // global-key forward has only filled it (state word 0) and k_first_pix completes it here, in place (idempotent: a
    // second backward over the same state repeats the same integer minima).  Which of the two happened is read from
    // the state the FORWARD wrote, on the device — never re-derived from tiled_ok() at backward time.
    int* first = const_cast<int*>(big) + 1 + b * nf;
#ifndef SR_ABL_NOFIRST      // (ablation build: timing of the gate itself; only valid after a tiled forward)
    if (b * nf > 0)
#else
    if (false)
#endif
        hipLaunchKernelGGL(k_first_pix, dim3((unsigned)std::min<long long>(sr_ceil_div(b * h * w, 256), 1024)), dim3(256), 0,
                           st, b * h * w, h * w, w, nf, win, first, first + b * nf);
    // attribute channels in chunks of <= 4 register accumulators; the vertex gradient rides with chunk 0
    for (long long ch0 = 0; ch0 < (grad_tex ? tex_c : 1); ch0 += 4) {
        const long long ct = tex_c - ch0 < 4 ? tex_c - ch0 : 4;
#define SR_GRAD_ARGS                                                                                              \
    b, nv, nf, h, w, repeat_f != 0, v, tex, (int)tex_c, (int)ch0, tri, win, big, grad_out, adj_off, adj, off_bs, \
        adj_bs, grad_v, grad_tex, eps, tg, first, valid, chw, slot_of, slot_bs, tail, accumulate, st
#define SR_GRAD_CASE(CT)                                          \
    do {                                                          \
        if (perspective) grad_launch<R, CT, true>(SR_GRAD_ARGS);  \
        else grad_launch<R, CT, false>(SR_GRAD_ARGS);             \
    } while (0)
        if (ct == 1) SR_GRAD_CASE(1);
        else if (ct == 2) SR_GRAD_CASE(2);
        else if (ct == 3) SR_GRAD_CASE(3);
        else SR_GRAD_CASE(4);
#undef SR_GRAD_CASE
#undef SR_GRAD_ARGS
    }
    return sr_launch_status();
}

template <int CT, bool PERSP>
void grad_levels_launch(const GradLevels& t, long long b, long long nv, long long nf, bool repeat_f, const float* v,
                        const float* tex, int tex_c, const long long* tri, const int* adj_off, const int* adj,
                        long long off_bs, long long adj_bs, const int* slot_of, float* grad_v, float* grad_tex, float eps,
                        bool chw, long long max_pix, hipStream_t st) {
    const bool want_v = grad_v != nullptr;
    const char* ee = getenv("SR_RASTER_GRAD_EAGER");
    const bool eager = ee ? ee[0] == '1' : b * nv < 8 * 32768;
    const unsigned n = (unsigned)t.n;
    hipLaunchKernelGGL(k_first_pix_levels, dim3((unsigned)std::min<long long>(sr_ceil_div(max_pix, 256), 1024), n), dim3(256),
                       0, st, t, b, nf);
    hipLaunchKernelGGL((k_grad_big_levels<CT, PERSP>), dim3(SR_NUM_CU * 2, n), dim3(256), 0, st, t, b, nv, nf, repeat_f, v,
                       tex, tex_c, tri, want_v, eps, chw, slot_of, adj_bs);
    hipLaunchKernelGGL((k_grad_pix_levels<CT, PERSP>), dim3((unsigned)sr_ceil_div(b * nf, 256), n), dim3(256), 0, st, t, b,
                       nv, nf, repeat_f, v, tex, tex_c, tri, want_v, eps, chw, slot_of, adj_bs);
    hipLaunchKernelGGL((k_grad_vert_levels<CT, PERSP>), dim3((unsigned)sr_ceil_div(nv, 256), (unsigned)b), dim3(256), 0, st,
                       t, b, nv, nf, adj_off, adj, off_bs, adj_bs, tex_c, grad_v, grad_tex, eager ? 2 : 1);
}

// ---- host path: the reference's sequential loops (op/rasterize.cpp:21-67, 69-95) on the shared arithmetic ------
template <typename R>
int forward_cpu(long long b, long long nv, long long nf, long long h, long long w, int repeat_v, int repeat_f,
                int perspective, const R* v, const long long* tri, long long* index, R* coeff, R* zbuf, R eps) {
    if (b < 0 || nv < 0 || nf < 0 || h <= 0 || w <= 0) return SR_EINVAL;
    if (b == 0) return SR_OK;
    if (!index || !coeff || (nf > 0 && (!v || !tri))) return SR_EINVAL;
    if (eps < 0) eps = -eps;
    const long long hw = h * w;
    std::vector<R> own;
    if (!zbuf) {
        own.resize((size_t)(b * hw));
        zbuf = own.data();
    }
    for (long long i = 0; i < b * hw; ++i) {
        zbuf[i] = Lim<R>::lowest();
        index[3 * i] = index[3 * i + 1] = index[3 * i + 2] = 0;
        coeff[3 * i] = coeff[3 * i + 1] = coeff[3 * i + 2] = 0;
    }
    for (long long s = 0; s < b; ++s) {
        const R* vs = repeat_v ? v : v + s * nv * 3;
        const long long* fs = repeat_f ? tri : tri + s * nf * 3;
        const long long shift = repeat_v ? 0 : nv * s;
        for (long long ti = 0; ti < nf; ++ti) {
            Tri<R> t;
            long long i0, i1, i2;
            if (!load_tri<R>(t, vs, fs, ti, nv, i0, i1, i2)) continue;
            if (!tri_setup<R>(t, h, w, perspective != 0, eps)) continue;
            for (int y = t.y0; y <= t.y1; ++y)
                for (int x = t.x0; x <= t.x1; ++x) {
                    const long long pix = x + (long long)y * w;
                    if (pix >= hw) continue;
                    R c0, c1, c2, z;
                    if (!shade<R>(t, x, y, perspective != 0, eps, c0, c1, c2, z)) continue;
                    const long long o = s * hw + pix;
                    if (zbuf[o] < z) {                     // in order: later triangles win only when nearer
                        zbuf[o] = z;
                        coeff[3 * o] = c0; coeff[3 * o + 1] = c1; coeff[3 * o + 2] = c2;
                        index[3 * o] = i0 + shift; index[3 * o + 1] = i1 + shift; index[3 * o + 2] = i2 + shift;
                    }
                }
        }
    }
    return SR_OK;
}

template <typename R>
int backward_cpu(long long b, long long n, long long h, long long w, int perspective, const R* v,
                 const long long* index, R* dcoeff, R eps) {
    if (b < 0 || n < 0 || h < 0 || w < 0) return SR_EINVAL;
    const long long hw = h * w, total = b * hw;
    if (total == 0) return SR_OK;
    if (!v || !index || !dcoeff) return SR_EINVAL;
    if (eps < 0) eps = -eps;
    for (long long g = 0; g < total; ++g) {
        R jac[27];
        for (int l = 0; l < 27; ++l) jac[l] = 0;
        R p[9];
        long long ids[3];
        if (load_pixel_tri<R>(index, g, n * b, v, p, ids)) {
            const long long pix = g % hw;
            weight_jacobian<R>(p, (R)(pix % w), (R)(pix / w), (R)h, (R)w, jac, perspective != 0, eps);
        }
        for (int l = 0; l < 27; ++l) dcoeff[g * 27 + l] = jac[l];
    }
    return SR_OK;
}

}  // namespace

extern "C" int64_t sr_rasterize_scratch_bytes(int64_t b, int64_t nf, int64_t h, int64_t w, int is_double) {
    b = b > 0 ? b : 0;
    nf = nf > 0 ? nf : 0;
    const int64_t npix = b * (h > 0 ? h : 0) * (w > 0 ? w : 0);
    int64_t bytes = npix * (is_double ? 12 : 8) + 16;                       // global-key path
    if (!is_double && h == w && h > 0) {
        const int64_t tiled = tile_scratch_words(b, nf, h) * 4 + 16;         // tile lists of the LDS-tiled path
        if (tiled > bytes) bytes = tiled;
    }
    return bytes;
}
extern "C" int64_t sr_rasterize_grad_scratch_bytes(int64_t b, int64_t nf, int64_t tex_c, int is_double) {
    (void)tex_c;
    const int64_t rows = (b > 0 ? b : 0) * (nf > 0 ? nf : 0);
    // records (or corner slots) | valid bits | inverse incidence table (3 ints per row at most)
    return rows * grad_row_floats() * (is_double ? 8 : 4) + ((rows + 63) / 64 + 1) * 8 + rows * 12 + 16;
}

extern "C" int sr_rasterize_forward_f32(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w,
                                        int repeat_v, int repeat_f, int perspective, const float* v,
                                        const int64_t* tri, int64_t* index, float* coeff, float* zbuf,
                                        float eps, const float* tex, int64_t tex_c, float* attr,
                                        int32_t* win, int32_t* big, void* work, sr_stream_t stream) {
    return forward_impl<float>(b, nv, nf, h, w, repeat_v, repeat_f, perspective, v,
                               reinterpret_cast<const long long*>(tri),
                               reinterpret_cast<long long*>(index), coeff, zbuf, eps, tex, tex_c, attr,
                               win, big, work, sr_stream(stream));
}
// n levels of the same mesh in three launches (global-key path, fp32): see k_fill_levels.  A level that the per-call
// dispatcher would hand to the LDS-tiled path makes the call refuse (SR_EINVAL; sr_rasterize_levels_supported says so in
// advance) — the caller then rasterises level by level.
extern "C" int sr_rasterize_levels_supported(int n, int64_t b, int64_t nf, const int64_t* h, const int64_t* w) {
    if (n <= 0 || n > RASTER_MAX_LEVELS || !h || !w || b <= 0 || b > 65535) return 0;
    for (int l = 0; l < n; ++l) {
        if (h[l] <= 0 || w[l] <= 0 || h[l] > 0x7FFFFFFF || w[l] > 0x7FFFFFFF || b * h[l] * w[l] >= 0x7FFFFFFFLL) return 0;
        if (tiled_ok<float>(b, nf, h[l], w[l])) return 0;
    }
    return 1;
}

extern "C" int sr_rasterize_forward_levels_f32(int n, int64_t b, int64_t nv, int64_t nf, const int64_t* h, const int64_t* w,
                                               int repeat_v, int repeat_f, int perspective, const float* v,
                                               const int64_t* tri, float eps, const float* tex, int64_t tex_c,
                                               float* const* attr, int32_t* const* win, int32_t* const* big,
                                               void* const* work, sr_stream_t stream) {
    if (n < 0 || b < 0 || nv < 0 || nf < 0 || tex_c <= 0) return SR_EINVAL;
    if (n == 0 || b == 0) return SR_OK;
    if (!h || !w || !attr || !work || !tex || (nf > 0 && (!v || !tri))) return SR_EINVAL;
    if (!sr_rasterize_levels_supported(n, b, nf, h, w)) return SR_EINVAL;
    const bool chw = (perspective & SR_RASTER_CHW) != 0;
    perspective &= 1;
    if (eps < 0) eps = -eps;
    RasterLevels t;
    t.n = n;
    long long max_pix = 0;
    for (int l = 0; l < n; ++l) {
        if (!attr[l] || !work[l]) return SR_EINVAL;
        t.res_h[l] = (int)h[l];
        t.res_w[l] = (int)w[l];
        t.keys[l] = reinterpret_cast<unsigned long long*>(work[l]);
        t.big[l] = (big && win && big[l] && win[l]) ? big[l] : nullptr;
        t.win[l] = (win && t.big[l]) ? win[l] : nullptr;
        t.attr[l] = attr[l];
        const long long npix = b * h[l] * w[l];
        if (npix > max_pix) max_pix = npix;
    }
    hipStream_t st = sr_stream(stream);
    const long long fill_items = max_pix > b * nf ? max_pix : b * nf;
    hipLaunchKernelGGL(k_fill_levels, dim3(sr_stream_grid(fill_items, 256), (unsigned)n), dim3(256), 0, st, t, (long long)b,
                       (long long)nf);
    if (b * nf > 0)
        hipLaunchKernelGGL(k_depth_keys_levels, dim3((unsigned)sr_ceil_div(b * nf, 256), (unsigned)n), dim3(256), 0, st, t,
                           (long long)b, (long long)nv, (long long)nf, repeat_v != 0, repeat_f != 0, perspective != 0, v,
                           reinterpret_cast<const long long*>(tri), eps);
    hipLaunchKernelGGL(k_resolve_levels, dim3((unsigned)sr_ceil_div(max_pix, 256), (unsigned)n), dim3(256), 0, st, t,
                       (long long)b, (long long)nv, (long long)nf, repeat_v != 0, repeat_f != 0, perspective != 0, v,
                       reinterpret_cast<const long long*>(tri), tex, (long long)tex_c, eps, chw);
    return sr_launch_status();
}
// Gradient of n levels of one mesh (sr_rasterize_forward_levels_f32 / any forward that left win / big per level) in four
// launches; grad_v / grad_tex receive the SUM over the levels, added in table order.  fp32, tex_c <= 4, nf > 0, the cached
// inverse incidence table adj_slot; anything else: SR_EINVAL (the caller accumulates level by level, SR_RASTER_GRAD_ACC).
extern "C" int sr_rasterize_grad_levels_f32(int n, int64_t b, int64_t nv, int64_t nf, const int64_t* h, const int64_t* w,
                                            int repeat_f, int perspective, const float* v, const float* tex, int64_t tex_c,
                                            const int64_t* tri, const int32_t* const* win, int32_t* const* big,
                                            const float* const* grad_out, const int32_t* adj_off, const int32_t* adj,
                                            int64_t off_bstride, int64_t adj_bstride, const int32_t* adj_slot,
                                            float* grad_v, float* grad_tex, float eps, void* const* work,
                                            sr_stream_t stream) {
    if (n <= 0 || n > RASTER_MAX_LEVELS || b <= 0 || nv <= 0 || nf <= 0 || tex_c <= 0 || tex_c > 4) return SR_EINVAL;
    if (!h || !w || !v || !tex || !tri || !win || !big || !grad_out || !adj_off || !adj || !adj_slot || !work ||
        (!grad_v && !grad_tex))
        return SR_EINVAL;
    if (b > 65535 || nf >= 0x7FFFFFFFLL / 3) return SR_ERANGE;
    const bool chw = (perspective & SR_RASTER_CHW) != 0;
    perspective &= 1;
    if (eps < 0) eps = -eps;
    GradLevels t;
    t.n = n;
    long long max_pix = 0;
    for (int l = 0; l < n; ++l) {
        if (h[l] <= 0 || w[l] <= 0 || !win[l] || !big[l] || !grad_out[l] || !work[l]) return SR_EINVAL;
        if (b * h[l] * w[l] >= 0x7FFFFFFFLL) return SR_ERANGE;
        t.res_h[l] = (int)h[l];
        t.res_w[l] = (int)w[l];
        t.win[l] = win[l];
        t.big[l] = big[l];
        t.grad_out[l] = grad_out[l];
        t.tg[l] = reinterpret_cast<float*>(work[l]);
        t.valid[l] = reinterpret_cast<unsigned long long*>(t.tg[l] + b * nf * grad_row_floats());
        if (b * h[l] * w[l] > max_pix) max_pix = b * h[l] * w[l];
    }
    hipStream_t st = sr_stream(stream);
    const long long* tri_ll = reinterpret_cast<const long long*>(tri);
#define SR_GL_ARGS t, b, nv, nf, repeat_f != 0, v, tex, (int)tex_c, tri_ll, adj_off, adj, off_bstride, adj_bstride, adj_slot, \
                   grad_v, grad_tex, eps, chw, max_pix, st
#define SR_GL_CASE(CT)                                             \
    do {                                                           \
        if (perspective) grad_levels_launch<CT, true>(SR_GL_ARGS); \
        else grad_levels_launch<CT, false>(SR_GL_ARGS);            \
    } while (0)
    if (tex_c == 1) SR_GL_CASE(1);
    else if (tex_c == 2) SR_GL_CASE(2);
    else if (tex_c == 3) SR_GL_CASE(3);
    else SR_GL_CASE(4);
#undef SR_GL_CASE
#undef SR_GL_ARGS
    return sr_launch_status();
}

extern "C" int sr_rasterize_forward_f64(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w,
                                        int repeat_v, int repeat_f, int perspective, const double* v,
                                        const int64_t* tri, int64_t* index, double* coeff,
                                        double* zbuf, double eps, const double* tex, int64_t tex_c,
                                        double* attr, int32_t* win, int32_t* big, void* work,
                                        sr_stream_t stream) {
    return forward_impl<double>(b, nv, nf, h, w, repeat_v, repeat_f, perspective, v,
                                reinterpret_cast<const long long*>(tri),
                                reinterpret_cast<long long*>(index), coeff, zbuf, eps, tex, tex_c, attr,
                                win, big, work, sr_stream(stream));
}
extern "C" int sr_rasterize_backward_f32(int64_t b, int64_t n, int64_t h, int64_t w, int repeat_v,
                                         int perspective, const float* v, const int64_t* index,
                                         float* dcoeff, float eps, sr_stream_t stream) {
    (void)repeat_v;
    return backward_impl<float>(b, n, h, w, perspective, v, reinterpret_cast<const long long*>(index),
                                dcoeff, eps, sr_stream(stream));
}
extern "C" int sr_rasterize_backward_f64(int64_t b, int64_t n, int64_t h, int64_t w, int repeat_v,
                                         int perspective, const double* v, const int64_t* index,
                                         double* dcoeff, double eps, sr_stream_t stream) {
    (void)repeat_v;
    return backward_impl<double>(b, n, h, w, perspective, v, reinterpret_cast<const long long*>(index),
                                 dcoeff, eps, sr_stream(stream));
}
extern "C" int sr_rasterize_grad_f32(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w, int repeat_f,
                                     int perspective, const float* v, const float* tex, int64_t tex_c,
                                     const int64_t* tri, const int32_t* win, const int32_t* big,
                                     const float* grad_out,
                                     const int32_t* adj_off, const int32_t* adj, int64_t adj_off_bstride,
                                     int64_t adj_bstride, const int32_t* adj_slot, float* grad_v, float* grad_tex,
                                     float eps, void* work, sr_stream_t stream) {
    return grad_impl<float>(b, nv, nf, h, w, repeat_f, perspective, v, tex, tex_c,
                            reinterpret_cast<const long long*>(tri), win, big, grad_out, adj_off, adj,
                            adj_off_bstride, adj_bstride, adj_slot, grad_v, grad_tex, eps, work, sr_stream(stream));
}
extern "C" int sr_rasterize_grad_f64(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w, int repeat_f,
                                     int perspective, const double* v, const double* tex, int64_t tex_c,
                                     const int64_t* tri, const int32_t* win, const int32_t* big,
                                     const double* grad_out,
                                     const int32_t* adj_off, const int32_t* adj, int64_t adj_off_bstride,
                                     int64_t adj_bstride, const int32_t* adj_slot, double* grad_v, double* grad_tex,
                                     double eps, void* work, sr_stream_t stream) {
    return grad_impl<double>(b, nv, nf, h, w, repeat_f, perspective, v, tex, tex_c,
                             reinterpret_cast<const long long*>(tri), win, big, grad_out, adj_off, adj,
                             adj_off_bstride, adj_bstride, adj_slot, grad_v, grad_tex, eps, work, sr_stream(stream));
}
extern "C" int sr_rasterize_forward_cpu_f32(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w, int repeat_v,
                                            int repeat_f, int perspective, const float* v, const int64_t* tri,
                                            int64_t* index, float* coeff, float* zbuf, float eps) {
    return forward_cpu<float>(b, nv, nf, h, w, repeat_v, repeat_f, perspective, v,
                              reinterpret_cast<const long long*>(tri), reinterpret_cast<long long*>(index), coeff,
                              zbuf, eps);
}
extern "C" int sr_rasterize_forward_cpu_f64(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w, int repeat_v,
                                            int repeat_f, int perspective, const double* v, const int64_t* tri,
                                            int64_t* index, double* coeff, double* zbuf, double eps) {
    return forward_cpu<double>(b, nv, nf, h, w, repeat_v, repeat_f, perspective, v,
                               reinterpret_cast<const long long*>(tri), reinterpret_cast<long long*>(index), coeff,
                               zbuf, eps);
}
extern "C" int sr_rasterize_backward_cpu_f32(int64_t b, int64_t n, int64_t h, int64_t w, int perspective,
                                             const float* v, const int64_t* index, float* dcoeff, float eps) {
    return backward_cpu<float>(b, n, h, w, perspective, v, reinterpret_cast<const long long*>(index), dcoeff, eps);
}
extern "C" int sr_rasterize_backward_cpu_f64(int64_t b, int64_t n, int64_t h, int64_t w, int perspective,
                                             const double* v, const int64_t* index, double* dcoeff, double eps) {
    return backward_cpu<double>(b, n, h, w, perspective, v, reinterpret_cast<const long long*>(index), dcoeff, eps);
}
