// Skinny matrix products of ALL modulated layers of a pass in one launch each (C ABI: sr_bank_nt / _nn / _tn).
//
// A generator pass evaluates, per modulated convolution, s = EqualLinear(latent row) [B, Ci] and — for the demodulated
// ones — q = s^2 @ Wsq [B, Co] (reference layers.py:222-248, 293-300): 20 + 13 products of a few MFLOP at 256^2, pure
// launch latency when they are launched per layer, and the path-length regulariser differentiates them twice.  Round 3
// batched them through stacked weights and rocBLAS strided-batched GEMMs (op/style_bank.py); the stacking itself was
// then ~25 copy launches per forward and the products the only vendor-library kernels of the hot path.  Here a launch
// takes a TABLE of problems (per-problem device pointers and extents by value in the kernel argument, like
// sr_weight_prep_batch), so the weights are read where they lie:
//
//   NT   out_p[b,n] = alpha * sum_k A_p[b,k] * M_p[n,k] + bscale * bias_p[n]        one wave per (p, n)
//   NN   out_o[b,j] = alpha * sum_{t in terms(o)} sum_i A_t[b,i] * M_t[i,j]          lanes across j, 4 waves split i
//   TN   out_p[i,j] = alpha * sum_b A_p[b,i] * C_p[b,j];  col_p[i] = bscale * sum_b A_p[b,i]
//
// The three are each other's derivatives (op/bankmm.py closes them under autograd), rows of A / C / out have their own
// pitches (a latent row of [B, n_latent, K] is read in place; NN adds the terms of one output — the layers that share
// a latent row — in table order), and every sum runs in a fixed order: deterministic, like csrc/style_linear.hip whose
// per-lane arithmetic these kernels repeat (same k / i / b order per output element).
#include "common.h"

namespace {

constexpr int BANK_MAX = SR_BANK_MAX;

struct BankNT {
    int n;
    int first[BANK_MAX + 1];           // first workgroup of problem p (4 output columns per workgroup)
    const float* A[BANK_MAX];
    const float* M[BANK_MAX];
    const float* bias[BANK_MAX];
    float* out[BANK_MAX];
    int lda[BANK_MAX], ldo[BANK_MAX], K[BANK_MAX], N[BANK_MAX];
};

struct BankNN {
    int n_out;
    int first[BANK_MAX + 1];           // first workgroup column of output o (256 output columns per workgroup)
    float* out[BANK_MAX];
    int ldo[BANK_MAX], J[BANK_MAX], term0[BANK_MAX + 1];
    const float* A[BANK_MAX];          // per TERM
    const float* M[BANK_MAX];
    int lda[BANK_MAX], I[BANK_MAX];
};

struct BankTN {
    int n;
    int first[BANK_MAX + 1];           // first workgroup of problem p (one output row per workgroup)
    const float* A[BANK_MAX];
    const float* C[BANK_MAX];
    float* out[BANK_MAX];
    float* col[BANK_MAX];
    int lda[BANK_MAX], ldc[BANK_MAX], I[BANK_MAX], J[BANK_MAX];
};

__device__ __forceinline__ float dot4(float4 a, float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// which problem does workgroup `blk` belong to (tables of <= 48 entries: a linear scan of wave-uniform values)
__device__ __forceinline__ int find_problem(const int* first, int n, int blk) {
    int p = 0;
    while (p + 1 < n && blk >= first[p + 1]) ++p;
    return p;
}

__global__ __launch_bounds__(256) void k_bank_nt(const BankNT t, int B, float alpha, float bscale) {
    const int lane = threadIdx.x & 63;
    const int p = find_problem(t.first, t.n, blockIdx.x);
    const int N = t.N[p], K = t.K[p];
    const int n = (blockIdx.x - t.first[p]) * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* A = t.A[p];
    const int64_t lda = t.lda[p];
    const float4* Mr = reinterpret_cast<const float4*>(t.M[p] + (int64_t)n * K);
    const int k4n = K >> 2;
    constexpr int RB = 8;
    for (int b0 = 0; b0 < B; b0 += RB) {
        float acc[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) acc[u] = 0.0f;
        for (int k4 = lane; k4 < k4n; k4 += 64) {
            const float4 m = Mr[k4];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int b = min(b0 + u, B - 1);
                acc[u] += dot4(reinterpret_cast<const float4*>(A + (int64_t)b * lda)[k4], m);
            }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const float r = sr_wave_sum(acc[u]);
            const int b = b0 + u;
            if (lane == 0 && b < B)
                t.out[p][(int64_t)b * t.ldo[p] + n] = alpha * r + (t.bias[p] ? bscale * t.bias[p][n] : 0.0f);
        }
    }
}

__global__ __launch_bounds__(256) void k_bank_nn(const BankNN t, float alpha) {
    __shared__ float4 part[3][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int o = find_problem(t.first, t.n_out, blockIdx.x);
    const int b = blockIdx.y;
    const int J = t.J[o], j4n = J >> 2;
    const int j4 = (blockIdx.x - t.first[o]) * 64 + lane;
    const bool ok = j4 < j4n;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int term = t.term0[o]; term < t.term0[o + 1]; ++term) {
        const int I = t.I[term];
        const int per = (I + 3) / 4;
        const int i_lo = wave * per, i_hi = min(I, i_lo + per);
        const float* Ar = t.A[term] + (int64_t)b * t.lda[term];
        const float4* Mc = reinterpret_cast<const float4*>(t.M[term]) + (ok ? j4 : 0);
#pragma unroll 16
        for (int i = i_lo; i < i_hi; ++i) {
            const float a = Ar[i];
            const float4 m = Mc[(int64_t)i * j4n];
            acc.x += a * m.x; acc.y += a * m.y; acc.z += a * m.z; acc.w += a * m.w;
        }
    }
    if (wave > 0) part[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0 && ok) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const float4 q = part[w][lane];
            acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
        }
        acc.x *= alpha; acc.y *= alpha; acc.z *= alpha; acc.w *= alpha;
        reinterpret_cast<float4*>(t.out[o] + (int64_t)b * t.ldo[o])[j4] = acc;
    }
}

__global__ __launch_bounds__(128) void k_bank_tn(const BankTN t, int B, float alpha, float bscale) {
    const int p = find_problem(t.first, t.n, blockIdx.x);
    const int i = blockIdx.x - t.first[p];
    const int J = t.J[p], j4n = J >> 2;
    const float* A = t.A[p];
    const float* C = t.C[p];
    const int64_t lda = t.lda[p], ldc = t.ldc[p];
    float asum = 0.0f;
    for (int j4 = threadIdx.x; j4 < j4n; j4 += 128) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        asum = 0.0f;
#pragma unroll 4
        for (int b = 0; b < B; ++b) {
            const float a = A[(int64_t)b * lda + i];
            asum += a;
            const float4 c = reinterpret_cast<const float4*>(C + (int64_t)b * ldc)[j4];
            acc.x += a * c.x; acc.y += a * c.y; acc.z += a * c.z; acc.w += a * c.w;
        }
        acc.x *= alpha; acc.y *= alpha; acc.z *= alpha; acc.w *= alpha;
        reinterpret_cast<float4*>(t.out[p] + (int64_t)i * J)[j4] = acc;
    }
    // (J >= 4 is checked by the host: thread 0 always ran the loop above and holds the column sum)
    if (t.col[p] && threadIdx.x == 0) t.col[p][i] = bscale * asum;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int sr_bank_nt(int n, float* const* out, const float* const* A, const float* const* M,
                          const float* const* bias, const int64_t* lda, const int64_t* ldo, const int64_t* K,
                          const int64_t* N, int64_t B, float alpha, float bscale, sr_stream_t stream) {
    if (n < 0 || B < 0 || (n > 0 && (!out || !A || !M || !lda || !ldo || !K || !N))) return SR_EINVAL;
    if (B == 0) return SR_OK;
    if (B > 65535) return SR_ERANGE;
    hipStream_t st = sr_stream(stream);
    for (int base = 0; base < n; base += BANK_MAX) {
        BankNT t;
        t.n = n - base < BANK_MAX ? n - base : BANK_MAX;
        int64_t total = 0;
        for (int i = 0; i < t.n; ++i) {
            const int g = base + i;
            if (!out[g] || !A[g] || !M[g] || K[g] <= 0 || N[g] <= 0 || (K[g] & 3) || (lda[g] & 3) || lda[g] < K[g] ||
                ldo[g] < N[g] || !al16(A[g]) || !al16(M[g]))
                return SR_EINVAL;
            if (K[g] > (1 << 20) || N[g] > (1 << 20) || lda[g] > 0x7FFFFFFF || ldo[g] > 0x7FFFFFFF) return SR_ERANGE;
            t.out[i] = out[g]; t.A[i] = A[g]; t.M[i] = M[g]; t.bias[i] = bias ? bias[g] : nullptr;
            t.lda[i] = (int)lda[g]; t.ldo[i] = (int)ldo[g]; t.K[i] = (int)K[g]; t.N[i] = (int)N[g];
            t.first[i] = (int)total;
            total += sr_ceil_div(N[g], 4);
            if (total > 0x7FFFFFFFLL) return SR_ERANGE;
        }
        t.first[t.n] = (int)total;
        if (total > 0)
            hipLaunchKernelGGL(k_bank_nt, dim3((unsigned)total), dim3(256), 0, st, t, (int)B, alpha, bscale);
    }
    return sr_launch_status();
}

extern "C" int sr_bank_nn(int n_out, float* const* out, const int64_t* ldo, const int64_t* J, const int* n_terms,
                          const float* const* A, const float* const* M, const int64_t* lda, const int64_t* I,
                          int64_t B, float alpha, sr_stream_t stream) {
    if (n_out < 0 || B < 0 || (n_out > 0 && (!out || !ldo || !J || !n_terms || !A || !M || !lda || !I))) return SR_EINVAL;
    if (B == 0) return SR_OK;
    if (B > 65535) return SR_ERANGE;
    hipStream_t st = sr_stream(stream);
    int term = 0, o = 0;
    while (o < n_out) {
        BankNN t;
        t.n_out = 0;
        int64_t total = 0;
        int nt = 0;
        while (o < n_out && t.n_out < BANK_MAX) {
            const int k = n_terms[o];
            if (k <= 0 || k > BANK_MAX) return SR_EINVAL;
            if (nt + k > BANK_MAX) break;
            if (!out[o] || J[o] <= 0 || (J[o] & 3) || ldo[o] < J[o] || (ldo[o] & 3) || !al16(out[o])) return SR_EINVAL;
            if (J[o] > (1 << 20) || ldo[o] > 0x7FFFFFFF) return SR_ERANGE;
            const int i = t.n_out;
            t.out[i] = out[o]; t.ldo[i] = (int)ldo[o]; t.J[i] = (int)J[o]; t.term0[i] = nt;
            for (int q = 0; q < k; ++q, ++term, ++nt) {
                if (!A[term] || !M[term] || I[term] <= 0 || lda[term] < I[term] || !al16(M[term])) return SR_EINVAL;
                if (I[term] > (1 << 20) || lda[term] > 0x7FFFFFFF) return SR_ERANGE;
                t.A[nt] = A[term]; t.M[nt] = M[term]; t.lda[nt] = (int)lda[term]; t.I[nt] = (int)I[term];
            }
            t.first[i] = (int)total;
            total += sr_ceil_div(J[o] >> 2, 64);
            ++t.n_out;
            ++o;
        }
        t.term0[t.n_out] = nt;
        t.first[t.n_out] = (int)total;
        if (total > 0 && total <= 0x7FFFFFFFLL)
            hipLaunchKernelGGL(k_bank_nn, dim3((unsigned)total, (unsigned)B), dim3(256), 0, st, t, alpha);
    }
    return sr_launch_status();
}

extern "C" int sr_bank_tn(int n, float* const* out, float* const* col, const float* const* A, const float* const* C,
                          const int64_t* lda, const int64_t* ldc, const int64_t* I, const int64_t* J, int64_t B,
                          float alpha, float bscale, sr_stream_t stream) {
    if (n < 0 || B < 0 || (n > 0 && (!out || !A || !C || !lda || !ldc || !I || !J))) return SR_EINVAL;
    if (B > 65535) return SR_ERANGE;
    hipStream_t st = sr_stream(stream);
    for (int base = 0; base < n; base += BANK_MAX) {
        BankTN t;
        t.n = n - base < BANK_MAX ? n - base : BANK_MAX;
        int64_t total = 0;
        for (int i = 0; i < t.n; ++i) {
            const int g = base + i;
            if (!out[g] || !A[g] || !C[g] || I[g] <= 0 || J[g] < 4 || (J[g] & 3) || (ldc[g] & 3) || ldc[g] < J[g] ||
                lda[g] < I[g] || !al16(C[g]) || !al16(out[g]))
                return SR_EINVAL;
            if (I[g] > (1 << 20) || J[g] > (1 << 20) || lda[g] > 0x7FFFFFFF || ldc[g] > 0x7FFFFFFF) return SR_ERANGE;
            t.out[i] = out[g]; t.col[i] = col ? col[g] : nullptr; t.A[i] = A[g]; t.C[i] = C[g];
            t.lda[i] = (int)lda[g]; t.ldc[i] = (int)ldc[g]; t.I[i] = (int)I[g]; t.J[i] = (int)J[g];
            t.first[i] = (int)total;
            total += I[g];
            if (total > 0x7FFFFFFFLL) return SR_ERANGE;
        }
        t.first[t.n] = (int)total;
        if (total > 0)
            hipLaunchKernelGGL(k_bank_tn, dim3((unsigned)total), dim3(128), 0, st, t, (int)B, alpha, bscale);
    }
    return sr_launch_status();
}
