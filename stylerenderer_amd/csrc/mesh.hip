// Vertex normals of a posed triangle mesh (C ABI: sr_vertex_normals_f32) — the step in front of
// op.rasterize in the reference's training loop (train.py:250-251, 305-306: random_apply_pose3D ->
// mesh_point_normal -> GeneratorWithMap).
//
// The reference (utils_3d.py:379-404) builds three [nv x nf] sparse incidence matrices with Python-side
// index tensors on every call and runs three sparse.mm scatters plus a normalisation.  Here the
// incidence is a CSR list built once per topology (host side, stylerenderer_amd/utils_3d.py) and one
// lane per (sample, vertex) GATHERS its incident face normals in a fixed order — no atomics, run-to-run
// identical, and the same association as the reference:
//     vn = ((0 + S_0) + S_1) + S_2,   S_k = sum over faces f (ascending) with tri[f][k] == vertex
//     fn(f) = (b - a) x (c - a)       out = vn / max(sqrt((x^2 + y^2) + z^2), eps)      (layers.py:19-22)
// Compiled with -ffp-contract=off like the other bit-comparable kernels.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void k_vertex_normals(float* __restrict__ vn, float* __restrict__ norm_out,
                                                        const float* __restrict__ v, const int64_t* __restrict__ tri,
                                                        const int* __restrict__ adj_off, const int* __restrict__ adj,
                                                        int nv, int nf, float eps) {
    const int vert = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const bool live = vert < nv;
    const float* vb = v + (int64_t)b * nv * 3;
    float ax = 0.f, ay = 0.f, az = 0.f;        // running total over corners
    float sx = 0.f, sy = 0.f, sz = 0.f;        // S_k of the current corner
    int cur_k = 0;
    const int e0 = live ? adj_off[vert] : 0, e1 = live ? adj_off[vert + 1] : 0;
    auto face_normal = [&](int f, float& nx, float& ny, float& nz) {
        const int64_t i0 = tri[(int64_t)f * 3], i1 = tri[(int64_t)f * 3 + 1], i2 = tri[(int64_t)f * 3 + 2];
        const float p0x = vb[i0 * 3], p0y = vb[i0 * 3 + 1], p0z = vb[i0 * 3 + 2];
        const float abx = vb[i1 * 3] - p0x, aby = vb[i1 * 3 + 1] - p0y, abz = vb[i1 * 3 + 2] - p0z;
        const float acx = vb[i2 * 3] - p0x, acy = vb[i2 * 3 + 1] - p0y, acz = vb[i2 * 3 + 2] - p0z;
        nx = aby * acz - abz * acy;
        ny = abz * acx - abx * acz;
        nz = abx * acy - aby * acx;
    };
    // A vertex of high valence (the poles of a UV mesh: 192 faces; any fan) is summed by its WAVE: one lane walking 192
    // dependent gathers held the whole launch (70 us for 24 770 vertices; the rasterizer's k_grad_vert had the same
    // shape).  Per corner k the lanes take the list positions of that corner strided by 64, in order, then a fixed-order
    // tree over the lanes; the association over corners stays ((0 + S_0) + S_1) + S_2.
    const bool wide = e1 - e0 > 24;
    if (!wide) {
        for (int e = e0; e < e1; ++e) {
            const int idx = adj[e];                // corner-major: k * nf + f
            const int k = idx / nf, f = idx - k * nf;
            if (k != cur_k) {
                ax += sx; ay += sy; az += sz;
                sx = sy = sz = 0.f;
                cur_k = k;
            }
            float nx, ny, nz;
            face_normal(f, nx, ny, nz);
            sx += nx;
            sy += ny;
            sz += nz;
        }
    }
    unsigned long long wm = __ballot(wide);
    const int lane = threadIdx.x & 63;
    while (wm) {
        const int src = __ffsll((long long)wm) - 1;
        wm &= wm - 1ull;
        const int we0 = __shfl(e0, src, 64), we1 = __shfl(e1, src, 64);
        float p[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};     // [corner][component]
        for (int e = we0 + lane; e < we1; e += 64) {
            const int idx = adj[e];
            const int k = idx / nf, f = idx - k * nf;
            float nx, ny, nz;
            face_normal(f, nx, ny, nz);
#pragma unroll
            for (int kk = 0; kk < 3; ++kk)
                if (k == kk) { p[kk][0] += nx; p[kk][1] += ny; p[kk][2] += nz; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int kk = 0; kk < 3; ++kk)
#pragma unroll
                for (int j = 0; j < 3; ++j) p[kk][j] += __shfl_down(p[kk][j], o, 64);
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
#pragma unroll
            for (int j = 0; j < 3; ++j) p[kk][j] = __shfl(p[kk][j], 0, 64);
        if (lane == src) {
            ax = (0.f + p[0][0]) + p[1][0]; ay = (0.f + p[0][1]) + p[1][1]; az = (0.f + p[0][2]) + p[1][2];
            sx = p[2][0]; sy = p[2][1]; sz = p[2][2];
        }
    }
    if (!live) return;
    ax += sx; ay += sy; az += sz;
    float n = sqrtf((ax * ax + ay * ay) + az * az);
    n = n < eps ? eps : n;
    float* o = vn + ((int64_t)b * nv + vert) * 3;
    o[0] = ax / n;
    o[1] = ay / n;
    o[2] = az / n;
    if (norm_out) norm_out[(int64_t)b * nv + vert] = n;
}

// ---- posed mesh: out[b, i, :] = v[b | 0, i, :] @ M[b] + t[b] (M [3,3] row-major, t [3] or NULL) ------------------------
// The pose of the latent-inversion loop and of the training loop's random_apply_pose3D is a [nv,3] x [3,3] product:
// as a library GEMM it is a 16x16-tile kernel of ~150 us at nv = 24 770 (a 3-wide output is all padding) and its
// backward two more.  One streaming pass here; the gradient of (M, t) is a fixed-order tree sum (deterministic).
__global__ __launch_bounds__(256) void k_affine3_fwd(float* __restrict__ out, const float* __restrict__ v,
                                                     const float* __restrict__ m, const float* __restrict__ t,
                                                     int64_t nv, int64_t v_bstride) {
    const int b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nv) return;
    const float* mb = m + 9 * b;
    const float* p = v + b * v_bstride + 3 * i;
    const float x = p[0], y = p[1], z = p[2];
    float o0 = x * mb[0], o1 = x * mb[1], o2 = x * mb[2];
    o0 += y * mb[3]; o1 += y * mb[4]; o2 += y * mb[5];
    o0 += z * mb[6]; o1 += z * mb[7]; o2 += z * mb[8];
    if (t) { o0 += t[3 * b]; o1 += t[3 * b + 1]; o2 += t[3 * b + 2]; }
    float* o = out + (b * nv + i) * 3;
    o[0] = o0; o[1] = o1; o[2] = o2;
}

// gm[b][j][k] = sum_i v[i][j] g[i][k], gt[b][k] = sum_i g[i][k]: one workgroup per sample, each lane sums a strided
// subset in index order, then a fixed-order LDS tree — 12 values.
__global__ __launch_bounds__(1024) void k_affine3_bwd(float* __restrict__ gm, float* __restrict__ gt,
                                                      const float* __restrict__ v, const float* __restrict__ g,
                                                      int64_t nv, int64_t v_bstride) {
    __shared__ float s[12][1024];
    const int b = blockIdx.x;
    float a[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) a[k] = 0.f;
#pragma unroll 4
    for (int64_t i = threadIdx.x; i < nv; i += 1024) {          // (loads of four vertices in flight per lane)
        const float* p = v + b * v_bstride + 3 * i;
        const float* q = g + (b * nv + i) * 3;
        const float x = p[0], y = p[1], z = p[2], g0 = q[0], g1 = q[1], g2 = q[2];
        a[0] += x * g0; a[1] += x * g1; a[2] += x * g2;
        a[3] += y * g0; a[4] += y * g1; a[5] += y * g2;
        a[6] += z * g0; a[7] += z * g1; a[8] += z * g2;
        a[9] += g0; a[10] += g1; a[11] += g2;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k][threadIdx.x] = a[k];
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int k = 0; k < 12; ++k) s[k][threadIdx.x] += s[k][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x < 9 && gm) gm[9 * b + threadIdx.x] = s[threadIdx.x][0];
    if (threadIdx.x >= 9 && threadIdx.x < 12 && gt) gt[3 * b + (threadIdx.x - 9)] = s[threadIdx.x][0];
}

// ---- pose of the latent-inversion loop: (yaw, pitch, roll, tx, ty, tz, log-scale) -> rot = Rz(roll) Rx(pitch) Ry(yaw) ("yxz"
// order of utils_3d.euler_mat: later axes multiply from the left) and lin = exp(log-scale) * rot, and the gradient of the
// seven numbers given the gradients of the two matrices.  As tensor algebra this is ~60 launches of one-element kernels
// per step (sin / cos / cat / view / three 3x3 products and their backward); here one lane each way.
__device__ __forceinline__ void mat3_mul(const float* a, const float* b, float* o) {      // o = a @ b, row-major
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}
__device__ __forceinline__ void axis_mats(const float* pose, float* ry, float* rx, float* rz, float* dry, float* drx,
                                          float* drz) {
    const float c0 = cosf(pose[0]), s0 = sinf(pose[0]), c1 = cosf(pose[1]), s1 = sinf(pose[1]);
    const float c2 = cosf(pose[2]), s2 = sinf(pose[2]);
    const float y[9] = {c0, 0.f, s0, 0.f, 1.f, 0.f, -s0, 0.f, c0}, dy[9] = {-s0, 0.f, c0, 0.f, 0.f, 0.f, -c0, 0.f, -s0};
    const float x[9] = {1.f, 0.f, 0.f, 0.f, c1, -s1, 0.f, s1, c1}, dx[9] = {0.f, 0.f, 0.f, 0.f, -s1, -c1, 0.f, c1, -s1};
    const float z[9] = {c2, -s2, 0.f, s2, c2, 0.f, 0.f, 0.f, 1.f}, dz[9] = {-s2, -c2, 0.f, c2, -s2, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 9; ++i) { ry[i] = y[i]; rx[i] = x[i]; rz[i] = z[i]; dry[i] = dy[i]; drx[i] = dx[i]; drz[i] = dz[i]; }
}

__global__ void k_pose_fwd(float* __restrict__ lin, float* __restrict__ rot, const float* __restrict__ pose, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;          // one pose per thread
    if (b >= B) return;
    pose += 7 * b;
    float ry[9], rx[9], rz[9], d0[9], d1[9], d2[9], t[9], r[9];
    axis_mats(pose, ry, rx, rz, d0, d1, d2);
    mat3_mul(rx, ry, t);
    mat3_mul(rz, t, r);
    const float sc = expf(pose[6]);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if (rot) rot[9 * b + i] = r[i];
        lin[9 * b + i] = sc * r[i];
    }
}

__global__ void k_pose_bwd(float* __restrict__ gpose, const float* __restrict__ glin, const float* __restrict__ grot,
                           const float* __restrict__ pose) {
    float ry[9], rx[9], rz[9], dry[9], drx[9], drz[9], t[9], r[9], u[9], d[9];
    axis_mats(pose, ry, rx, rz, dry, drx, drz);
    mat3_mul(rx, ry, t);
    mat3_mul(rz, t, r);
    const float sc = expf(pose[6]);
    float gt[9];                          // dL/dR = grot + sc * glin
    float gs = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float gl = glin ? glin[i] : 0.f;
        gt[i] = (grot ? grot[i] : 0.f) + sc * gl;
        gs += gl * r[i];
    }
    auto dot9 = [&](const float* m) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) a += gt[i] * m[i];
        return a;
    };
    mat3_mul(rx, dry, u); mat3_mul(rz, u, d); gpose[0] = dot9(d);      // d/d yaw:   Rz Rx Ry'
    mat3_mul(drx, ry, u); mat3_mul(rz, u, d); gpose[1] = dot9(d);      // d/d pitch: Rz Rx' Ry
    mat3_mul(drz, t, d);                      gpose[2] = dot9(d);      // d/d roll:  Rz' Rx Ry
    gpose[3] = gpose[4] = gpose[5] = 0.f;                              // (translation: sr_affine3_bwd's gt)
    gpose[6] = sc * gs;
}

}  // namespace

extern "C" int sr_pose_fwd(float* lin, float* rot, const float* pose, sr_stream_t stream) {
    if (!lin || !rot || !pose) return SR_EINVAL;
    hipLaunchKernelGGL(k_pose_fwd, dim3(1), dim3(1), 0, sr_stream(stream), lin, rot, pose, 1);
    return sr_launch_status();
}

extern "C" int sr_pose_batch_fwd(float* lin, float* rot, const float* pose, int64_t B, sr_stream_t stream) {
    if (B < 0) return SR_EINVAL;
    if (B == 0) return SR_OK;
    if (!lin || !pose) return SR_EINVAL;
    if (B > (1 << 24)) return SR_ERANGE;
    hipLaunchKernelGGL(k_pose_fwd, dim3((unsigned)sr_ceil_div(B, 64)), dim3(64), 0, sr_stream(stream), lin, rot, pose, (int)B);
    return sr_launch_status();
}

extern "C" int sr_pose_bwd(float* gpose, const float* glin, const float* grot, const float* pose, sr_stream_t stream) {
    if (!gpose || !pose) return SR_EINVAL;
    hipLaunchKernelGGL(k_pose_bwd, dim3(1), dim3(1), 0, sr_stream(stream), gpose, glin, grot, pose);
    return sr_launch_status();
}

extern "C" int sr_affine3_fwd(float* out, const float* v, const float* m, const float* t, int64_t B, int64_t nv,
                              int64_t v_bstride, sr_stream_t stream) {
    if (B < 0 || nv < 0) return SR_EINVAL;
    if (B == 0 || nv == 0) return SR_OK;
    if (!out || !v || !m) return SR_EINVAL;
    if (B > 65535) return SR_ERANGE;
    hipLaunchKernelGGL(k_affine3_fwd, dim3((unsigned)sr_ceil_div(nv, 256), (unsigned)B), dim3(256), 0, sr_stream(stream),
                       out, v, m, t, nv, v_bstride);
    return sr_launch_status();
}

extern "C" int sr_affine3_bwd(float* gm, float* gt, const float* v, const float* g, int64_t B, int64_t nv,
                              int64_t v_bstride, sr_stream_t stream) {
    if (B < 0 || nv < 0) return SR_EINVAL;
    if (B == 0 || (!gm && !gt)) return SR_OK;
    if (!v || !g) return SR_EINVAL;
    hipLaunchKernelGGL(k_affine3_bwd, dim3((unsigned)B), dim3(1024), 0, sr_stream(stream), gm, gt, v, g, nv, v_bstride);
    return sr_launch_status();
}

extern "C" int sr_vertex_normals_f32(float* vn, float* norm_out, const float* v, const int64_t* tri,
                                     const int32_t* adj_off, const int32_t* adj, int64_t B, int64_t nv, int64_t nf,
                                     float eps, sr_stream_t stream) {
    if (B < 0 || nv < 0 || nf < 0) return SR_EINVAL;
    if (B == 0 || nv == 0) return SR_OK;
    if (!vn || !v || !adj_off || (nf > 0 && (!tri || !adj))) return SR_EINVAL;
    if (B > 65535 || nv >= (1LL << 30) || 3 * nf >= (1LL << 31)) return SR_ERANGE;
    hipLaunchKernelGGL(k_vertex_normals, dim3((unsigned)sr_ceil_div(nv, 256), (unsigned)B), dim3(256), 0,
                       sr_stream(stream), vn, norm_out, v, tri, adj_off, adj, (int)nv, (int)(nf > 0 ? nf : 1), eps);
    return sr_launch_status();
}
