// Skinny linear algebra of the style path (C ABI: sr_linear_fwd / _bwd_x / _bwd_w, sr_demod_fwd / _bwd).
//
// Every modulated convolution needs s = EqualLinear(style) [B, Ci] (reference layers.py:222-248, 293)
// and the demodulation scale d = rsqrt(s^2 @ Wsq + eps) [B, Co] (layers.py:298-300 rewritten, see
// weight_prep.hip); the mapping network is eight [B,512]x[512,512] EqualLinear + fused leaky-ReLU
// layers (model.py:87-95).  B is the per-GPU batch (<= 64), so these are launch-latency-bound: the
// reference spends 7 ATen launches per layer forward and ~18 backward on them.  Three kernel shapes
// cover all of it, each with the element-wise pre/post operations fused:
//   NT   out[b,n] = post(sum_k A[b,k] * M[n,k])        one wave per n, lanes across k
//   NN   out[b,j] = post(sum_i pre(A[b,i]) * M[i,j])   lanes across j (float4), 4 waves split i
//   TN   out[i,j] = post(sum_b pre(A[b,i]) * C[b,j])   one workgroup per i, lanes across j
// All reductions run in a fixed order (deterministic).
#include "common.h"

namespace {

struct ActP {
    int act;        // 0 none, 1 leaky-ReLU * gain
    float alpha, gain;
};

__device__ __forceinline__ float act_fwd(float t, ActP a) { return a.act ? (t > 0.0f ? t : a.alpha * t) * a.gain : t; }
// derivative factor applied to an upstream gradient, from the saved OUTPUT y (sign(y) == sign(t))
__device__ __forceinline__ float act_bwd(float gy, float y, ActP a) {
    return a.act ? (y > 0.0f ? gy : a.alpha * gy) * a.gain : gy;
}
// d(rsqrt(q))/dq * gd  with d = rsqrt(q)
__device__ __forceinline__ float demod_gq(float gd, float d) { return -0.5f * gd * d * d * d; }

__device__ __forceinline__ float dot4(float4 a, float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// ---- NT.  MODE 0: y = act(wscale * x.W^T + bscale*bias).   MODE 1 (demod backward, style part):
// out[b,ci] = add[b,ci] + 2*s[b,ci] * sum_co gq[b,co]*wsq[ci,co],  gq = demod_gq(gd, d).
template <int MODE>
__global__ __launch_bounds__(256) void k_nt(float* __restrict__ out, const float* __restrict__ A,
                                            const float* __restrict__ A2, const float* __restrict__ M,
                                            const float* __restrict__ vec, const float* __restrict__ add,
                                            int B, int K, int N, int64_t lda, float wscale, float bscale,
                                            ActP ap) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float4* Mr = reinterpret_cast<const float4*>(M + (int64_t)n * K);
    const int k4n = K >> 2;
    // these kernels are latency bound (a few KB per wave): keep the row of M in registers when it fits
    // (K <= 1024) and put the loads of 8 rows of A in flight at once — two dependent memory round trips
    // per 8 rows instead of one per row group
    constexpr int RB = 8;
    float4 mreg[4];
    const bool mfit = k4n <= 256;
    if (mfit) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k4 = lane + 64 * q;
            mreg[q] = k4 < k4n ? Mr[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (int b0 = 0; b0 < B; b0 += RB) {
        float acc[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) acc[u] = 0.0f;
        if (mfit) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k4 = lane + 64 * q;
                if (k4 < k4n) {
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const int b = min(b0 + u, B - 1);
                        float4 a = reinterpret_cast<const float4*>(A + (int64_t)b * lda)[k4];
                        if (MODE == 1) {
                            const float4 d = reinterpret_cast<const float4*>(A2 + (int64_t)b * K)[k4];
                            a.x = demod_gq(a.x, d.x); a.y = demod_gq(a.y, d.y);
                            a.z = demod_gq(a.z, d.z); a.w = demod_gq(a.w, d.w);
                        }
                        acc[u] += dot4(a, mreg[q]);
                    }
                }
            }
        } else {
            for (int k4 = lane; k4 < k4n; k4 += 64) {
                const float4 m = Mr[k4];
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const int b = min(b0 + u, B - 1);
                    float4 a = reinterpret_cast<const float4*>(A + (int64_t)b * lda)[k4];
                    if (MODE == 1) {
                        const float4 d = reinterpret_cast<const float4*>(A2 + (int64_t)b * K)[k4];
                        a.x = demod_gq(a.x, d.x); a.y = demod_gq(a.y, d.y);
                        a.z = demod_gq(a.z, d.z); a.w = demod_gq(a.w, d.w);
                    }
                    acc[u] += dot4(a, m);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const float r = sr_wave_sum(acc[u]);
            const int b = b0 + u;
            if (lane == 0 && b < B) {
                if (MODE == 0) {
                    const float t = wscale * r + (vec ? bscale * vec[n] : 0.0f);
                    out[(int64_t)b * N + n] = act_fwd(t, ap);
                } else {
                    const float s = vec[(int64_t)b * N + n];
                    out[(int64_t)b * N + n] = (add ? add[(int64_t)b * N + n] : 0.0f) + 2.0f * s * r;
                }
            }
        }
    }
}

// ---- NN.  MODE 0 (linear backward, input part): gx[b,k] = wscale * sum_n act_bwd(gy,y)[b,n]*W[n,k].
// MODE 1 (demod forward): d[b,co] = rsqrt(sum_ci s[b,ci]^2 * wsq[ci,co] + eps)   (eps in wscale).
// NW waves split the I loop (16: a [B, 512] x [512, 512] product is 2 x B workgroups — with four waves each lane walked
// 128 rows of M: 14 us for 1 MB)
constexpr int NN_WAVES = 16;
template <int MODE>
__global__ __launch_bounds__(64 * NN_WAVES) void k_nn(float* __restrict__ out, const float* __restrict__ A,
                                                      const float* __restrict__ A2, const float* __restrict__ M, int I,
                                                      int J, float wscale, ActP ap) {
    __shared__ float4 part[NN_WAVES - 1][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int j4 = blockIdx.x * 64 + lane;
    const int j4n = J >> 2;
    const bool ok = j4 < j4n;
    const int per = (I + NN_WAVES - 1) / NN_WAVES;
    const int i_lo = wave * per, i_hi = min(I, i_lo + per);
    const float* Ar = A + (int64_t)b * I;
    const float* A2r = A2 ? A2 + (int64_t)b * I : nullptr;
    const float4* Mc = reinterpret_cast<const float4*>(M) + (ok ? j4 : 0);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 32
    for (int i = i_lo; i < i_hi; ++i) {
        float a = Ar[i];
        if (MODE == 0) a = act_bwd(a, A2r ? A2r[i] : 1.0f, ap);
        else a = a * a;
        const float4 m = Mc[(int64_t)i * j4n];
        acc.x += a * m.x; acc.y += a * m.y; acc.z += a * m.z; acc.w += a * m.w;
    }
    if (wave > 0) part[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0 && ok) {
#pragma unroll
        for (int w = 0; w < NN_WAVES - 1; ++w) {
            const float4 t = part[w][lane];
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        if (MODE == 0) {
            acc.x *= wscale; acc.y *= wscale; acc.z *= wscale; acc.w *= wscale;
        } else {
            acc.x = 1.0f / sqrtf(acc.x + wscale); acc.y = 1.0f / sqrtf(acc.y + wscale);
            acc.z = 1.0f / sqrtf(acc.z + wscale); acc.w = 1.0f / sqrtf(acc.w + wscale);
        }
        reinterpret_cast<float4*>(out + (int64_t)b * J)[j4] = acc;
    }
}

// ---- TN.  MODE 0 (linear backward, weight part): gW[n,k] = wscale * sum_b act_bwd(gy,y)[b,n]*x[b,k],
// gb[n] = bscale * sum_b act_bwd(gy,y)[b,n].   MODE 1 (demod backward, matrix part):
// gwsq[ci,co] = sum_b s[b,ci]^2 * gq[b,co],  gq = demod_gq(gd, d)  (C = gd, C2 = d).
template <int MODE>
__global__ __launch_bounds__(128) void k_tn(float* __restrict__ out, float* __restrict__ colsum,
                                            const float* __restrict__ A, const float* __restrict__ A2,
                                            const float* __restrict__ C, const float* __restrict__ C2, int B,
                                            int I, int J, int64_t ldc, float wscale, float bscale, ActP ap) {
    const int i = blockIdx.x;
    const int j4n = J >> 2;
    float asum = 0.0f;
    for (int j4 = threadIdx.x; j4 < j4n; j4 += 128) {          // J >= 4: thread 0 always runs once
        const bool ok = true;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        asum = 0.0f;
#pragma unroll 4
        for (int b = 0; b < B; ++b) {
            float a = A[(int64_t)b * I + i];
            if (MODE == 0) a = act_bwd(a, A2 ? A2[(int64_t)b * I + i] : 1.0f, ap);
            else a = a * a;
            asum += a;
            if (ok) {
                float4 c = reinterpret_cast<const float4*>(C + (int64_t)b * ldc)[j4];
                if (MODE == 1) {
                    const float4 d = reinterpret_cast<const float4*>(C2 + (int64_t)b * J)[j4];
                    c.x = demod_gq(c.x, d.x); c.y = demod_gq(c.y, d.y);
                    c.z = demod_gq(c.z, d.z); c.w = demod_gq(c.w, d.w);
                }
                acc.x += a * c.x; acc.y += a * c.y; acc.z += a * c.z; acc.w += a * c.w;
            }
        }
        if (ok) {
            if (MODE == 0) { acc.x *= wscale; acc.y *= wscale; acc.z *= wscale; acc.w *= wscale; }
            reinterpret_cast<float4*>(out + (int64_t)i * J)[j4] = acc;
        }
    }
    if (MODE == 0 && colsum && threadIdx.x == 0) colsum[i] = bscale * asum;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool dims_ok(int64_t B, int64_t K, int64_t N) {
    return B > 0 && K > 0 && N > 0 && B <= 65535 && K <= (1 << 20) && N <= (1 << 20);
}

}  // namespace

extern "C" int sr_linear_fwd(float* y, const float* x, const float* w, const float* bias, int64_t B, int64_t K,
                             int64_t N, int64_t ldx, float wscale, float bscale, int act, float alpha,
                             float gain, sr_stream_t stream) {
    if (B == 0 || N == 0) return SR_OK;
    if (!dims_ok(B, K, N) || !y || !x || !w || (K & 3) || ldx < K || (ldx & 3) || !al16(x) || !al16(w))
        return SR_EINVAL;
    const ActP ap{act ? 1 : 0, alpha, gain};
    hipLaunchKernelGGL(k_nt<0>, dim3((unsigned)sr_ceil_div(N, 4)), dim3(256), 0, sr_stream(stream), y, x,
                       (const float*)nullptr, w, bias, (const float*)nullptr, (int)B, (int)K, (int)N, ldx,
                       wscale, bscale, ap);
    return sr_launch_status();
}

extern "C" int sr_linear_bwd_x(float* gx, const float* gy, const float* y, const float* w, int64_t B, int64_t K,
                               int64_t N, float wscale, int act, float alpha, float gain, sr_stream_t stream) {
    if (B == 0 || K == 0) return SR_OK;
    if (!dims_ok(B, K, N) || !gx || !gy || !w || (act && !y) || (K & 3) || !al16(gx) || !al16(w)) return SR_EINVAL;
    const ActP ap{act ? 1 : 0, alpha, gain};
    hipLaunchKernelGGL(k_nn<0>, dim3((unsigned)sr_ceil_div(K >> 2, 64), (unsigned)B), dim3(64 * NN_WAVES), 0,
                       sr_stream(stream), gx, gy, act ? y : (const float*)nullptr, w, (int)N, (int)K, wscale, ap);
    return sr_launch_status();
}

extern "C" int sr_linear_bwd_w(float* gw, float* gbias, const float* gy, const float* y, const float* x,
                               int64_t B, int64_t K, int64_t N, int64_t ldx, float wscale, float bscale, int act,
                               float alpha, float gain, sr_stream_t stream) {
    if (N == 0 || K == 0) return SR_OK;
    if (!dims_ok(B, K, N) || !gw || !gy || !x || (act && !y) || (K & 3) || ldx < K || (ldx & 3) || !al16(gw) ||
        !al16(x))
        return SR_EINVAL;
    const ActP ap{act ? 1 : 0, alpha, gain};
    hipLaunchKernelGGL(k_tn<0>, dim3((unsigned)N), dim3(128), 0, sr_stream(stream), gw, gbias, gy,
                       act ? y : (const float*)nullptr, x, (const float*)nullptr, (int)B, (int)N, (int)K, ldx,
                       wscale, bscale, ap);
    return sr_launch_status();
}

extern "C" int sr_demod_fwd(float* d, const float* s, const float* wsq, int64_t B, int64_t Ci, int64_t Co,
                            float eps, sr_stream_t stream) {
    if (B == 0 || Co == 0) return SR_OK;
    if (!dims_ok(B, Ci, Co) || !d || !s || !wsq || (Co & 3) || !al16(d) || !al16(wsq)) return SR_EINVAL;
    const ActP ap{0, 0.f, 1.f};
    hipLaunchKernelGGL(k_nn<1>, dim3((unsigned)sr_ceil_div(Co >> 2, 64), (unsigned)B), dim3(64 * NN_WAVES), 0,
                       sr_stream(stream), d, s, (const float*)nullptr, wsq, (int)Ci, (int)Co, eps, ap);
    return sr_launch_status();
}

extern "C" int sr_demod_bwd(float* gs, float* gwsq, const float* gd, const float* gs_add, const float* s,
                            const float* d, const float* wsq, int64_t B, int64_t Ci, int64_t Co,
                            sr_stream_t stream) {
    if (B == 0) return SR_OK;
    if (!dims_ok(B, Ci, Co) || !gd || !s || !d || !wsq || (!gs && !gwsq) || (Co & 3) || !al16(gd) || !al16(d) ||
        !al16(wsq) || (gwsq && !al16(gwsq)))
        return SR_EINVAL;
    const ActP ap{0, 0.f, 1.f};
    hipStream_t st = sr_stream(stream);
    if (gs)
        hipLaunchKernelGGL(k_nt<1>, dim3((unsigned)sr_ceil_div(Ci, 4)), dim3(256), 0, st, gs, gd, d, wsq, s, gs_add,
                           (int)B, (int)Co, (int)Ci, Co, 1.0f, 1.0f, ap);
    if (gwsq)
        hipLaunchKernelGGL(k_tn<1>, dim3((unsigned)Ci), dim3(128), 0, st, gwsq, (float*)nullptr, s,
                           (const float*)nullptr, gd, d, (int)B, (int)Ci, (int)Co, Co, 1.0f, 1.0f, ap);
    return sr_launch_status();
}
