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
    if (vert >= nv) return;
    const float* vb = v + (int64_t)b * nv * 3;
    float ax = 0.f, ay = 0.f, az = 0.f;        // running total over corners
    float sx = 0.f, sy = 0.f, sz = 0.f;        // S_k of the current corner
    int cur_k = 0;
    const int e1 = adj_off[vert + 1];
    for (int e = adj_off[vert]; e < e1; ++e) {
        const int idx = adj[e];                // corner-major: k * nf + f
        const int k = idx / nf, f = idx - k * nf;
        if (k != cur_k) {
            ax += sx; ay += sy; az += sz;
            sx = sy = sz = 0.f;
            cur_k = k;
        }
        const int64_t i0 = tri[(int64_t)f * 3], i1 = tri[(int64_t)f * 3 + 1], i2 = tri[(int64_t)f * 3 + 2];
        const float p0x = vb[i0 * 3], p0y = vb[i0 * 3 + 1], p0z = vb[i0 * 3 + 2];
        const float abx = vb[i1 * 3] - p0x, aby = vb[i1 * 3 + 1] - p0y, abz = vb[i1 * 3 + 2] - p0z;
        const float acx = vb[i2 * 3] - p0x, acy = vb[i2 * 3 + 1] - p0y, acz = vb[i2 * 3 + 2] - p0z;
        sx += aby * acz - abz * acy;
        sy += abz * acx - abx * acz;
        sz += abx * acy - aby * acx;
    }
    ax += sx; ay += sy; az += sz;
    float n = sqrtf((ax * ax + ay * ay) + az * az);
    n = n < eps ? eps : n;
    float* o = vn + ((int64_t)b * nv + vert) * 3;
    o[0] = ax / n;
    o[1] = ay / n;
    o[2] = az / n;
    if (norm_out) norm_out[(int64_t)b * nv + vert] = n;
}

}  // namespace

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
