/*
 * stylerenderer_amd — C ABI of the MI355X-native (gfx950) StyleRenderer generator hot path.
 *
 * One shared library, libstylerenderer_hip.so, built from the .hip sources under stylerenderer_amd/csrc.
 * Every entry point is extern "C", takes plain device pointers + sizes + an explicit HIP
 * stream, allocates nothing, keeps no state and never synchronises the host (all calls are
 * hipGraph-capturable).  Each function states which reference interface it replaces
 * (paths relative to the reference repository WestlyPark/StyleRenderer).
 *
 * Ownership: the caller allocates every buffer (inputs contiguous, fp32 unless noted).
 * Errors: 0 on success; a negative SR_E* code for a rejected argument; a positive value is the
 * hipError_t of a failed launch.  sr_error_string() maps either to text.  (The reference's
 * native functions return `bool` and never check anything, op/upfirdn2d_kernel.cu:205-257.)
 *
 * Threading: any host thread; the kernels are enqueued on `stream` (pass the framework's
 * current stream — the reference uses c10::cuda::getCurrentCUDAStream for two ops,
 * op/fused_bias_act_kernel.cu:78-80, and the legacy default stream for the rasterizer,
 * op/rasterize.cu:89-92, which is deliberately not reproduced).
 */
#ifndef STYLERENDERER_AMD_H_
#define STYLERENDERER_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sr_stream_t; /* hipStream_t */

#define SR_OK 0
#define SR_EINVAL (-1)   /* bad size / null pointer / unsupported combination */
#define SR_ERANGE (-2)   /* size exceeds what the kernel indexes */

const char* sr_error_string(int code);
/* ABI version of this header; bumped on any signature change. */
int sr_abi_version(void);

/* ---------------------------------------------------------------------------------------
 * fused bias + activation
 * Replaces  bool fused_bias_act_op(float* y, const float* x, const float* b, const float* ref,
 *             int act, int grad, float alpha, float scale, int size_x, int step_b, int size_b,
 *             int use_bias, int use_ref, int use_cuda)   reference op/fused_bias_act_kernel.cu:71-76
 * (arithmetic: op/fused_bias_act_kernel.cu:15-42).  out[i] = f(x[i] + b[(i / step_b) % size_b]) * scale,
 * f selected by act*10+grad: 30 lrelu(x), 31 lrelu'(ref)*x, 32/12 zero, else identity.
 * Sizes are 64-bit here (the reference's `int` caps tensors at 2^31 elements). */
int sr_fused_bias_act(float* y, const float* x, const float* b, const float* ref, int act, int grad,
                      float alpha, float scale, int64_t size_x, int64_t step_b, int64_t size_b,
                      int use_bias, int use_ref, sr_stream_t stream);

/* Backward of the fused activation with the bias gradient reduced in the same pass.
 * Replaces the pair {fused_bias_act(gy, empty, out, 3, 1, ..) ; grad_input.sum(dims != 1)}
 * of reference op/fused_act.py:29-38.  x layout [n, c, inner]:
 *   gx[i]    = (out[i] > 0 ? gy[i] : alpha * gy[i]) * scale
 *   gb[ch]   = sum over n, inner of gx                      (deterministic two-stage tree)
 * `partial` is caller scratch of sr_fused_act_bwd_scratch_floats(n, c, inner) floats. */
int64_t sr_fused_act_bwd_scratch_floats(int64_t n, int64_t c, int64_t inner);
int sr_fused_act_bwd(float* gx, float* gb, const float* gy, const float* out, float alpha,
                     float scale, int64_t n, int64_t c, int64_t inner, float* partial,
                     sr_stream_t stream);

/* Fused StyledConv tail (reference model.py:26-32: NoiseInjection layers.py:328-332 followed by
 * FusedLeakyReLU op/fused_act.py:52-62) in one pass.  x, y [n, c, inner]; noise [n or 1, 1, inner]
 * with batch stride noise_bstride (0 = shared); noise_w a 1-element DEVICE tensor; bias [c].
 *   t = x + noise_w[0]*noise[b, i] + bias[ch] ;  y = (cond > 0 ? t : alpha*t) * scale
 * cond = t, or ref[...] when ref != NULL (the double-backward form).  noise / bias may be NULL.
 * Requires inner % 4 == 0 and 16-byte aligned pointers (SR_EINVAL otherwise: the caller then uses
 * sr_fused_bias_act). */
int sr_noise_bias_act(float* y, const float* x, const float* noise, const float* noise_w,
                      const float* bias, const float* ref, float alpha, float scale, int64_t n,
                      int64_t c, int64_t inner, int64_t noise_bstride, sr_stream_t stream);
/* Its backward in one pass: gx = (out > 0 ? gy : alpha*gy)*scale, gbias[ch] = sum gx,
 * gnoise_w[0] = sum gx*noise (deterministic two-stage reductions; gbias / gnoise_w may be NULL). */
int64_t sr_noise_bias_act_bwd_scratch_floats(int64_t n, int64_t c, int64_t inner);
int sr_noise_bias_act_bwd(float* gx, float* gbias, float* gnoise_w, const float* gy, const float* out,
                          const float* noise, float alpha, float scale, int64_t n, int64_t c,
                          int64_t inner, int64_t noise_bstride, float* scratch, sr_stream_t stream);
/* The same with one more output: rowdot[b*c + ch] = sum_i gx * y0, where y0 is the forward pass's input
 * rebuilt from its output (out / scale, or out / (alpha*scale) where negative, minus noise_w*noise and bias) —
 * the demodulation gradient of the modulated convolution in front of the activation, without a separate pass
 * over two tensors.  Needs alpha != 0 and scale != 0. */
int64_t sr_noise_bias_act_bwd_dot_scratch_floats(int64_t n, int64_t c, int64_t inner);
int sr_noise_bias_act_bwd_dot(float* gx, float* gbias, float* gnoise_w, float* rowdot, const float* gy,
                              const float* out, const float* noise, const float* noise_w, const float* bias,
                              float alpha, float scale, int64_t n, int64_t c, int64_t inner,
                              int64_t noise_bstride, float* scratch, sr_stream_t stream);
/* StyledMapConv tail (reference model.py:49-54: `out * stylemap[:, :1] + stylemap[:, 1:2]`, then
 * NoiseInjection and FusedLeakyReLU) as one pass and its backward as one pass (+ the fixed-order finish):
 *   y = lrelu((x * a[b,p] + s[b,p]) + noise_w * noise[b,p] + bias[c]) * scale
 *   gx = g * a ; gamap[b,p] = sum_c g * x ; gsmap[b,p] = sum_c g ; gbias[c] = sum_{b,p} g ; gnoise_w = sum g * noise
 * with g = lrelu'(y) * gy * scale.  x / y / gx [n, c, inner]; amap / smap point at [n] planes of `inner` floats
 * `map_bstride` floats apart (two channels of one rasterised-map tensor); gamap / gsmap [n, inner] contiguous.
 * inner % 4 == 0 and 16-byte aligned pointers. */
int sr_noise_bias_act_affine(float* y, const float* x, const float* amap, const float* smap, int64_t map_bstride,
                             const float* noise, const float* noise_w, const float* bias, float alpha,
                             float scale, int64_t n, int64_t c, int64_t inner, int64_t noise_bstride,
                             sr_stream_t stream);
int64_t sr_noise_bias_act_affine_bwd_scratch_floats(int64_t n, int64_t c, int64_t inner);
int sr_noise_bias_act_affine_bwd(float* gx, float* gamap, float* gsmap, float* gbias, float* gnoise_w,
                                 const float* gy, const float* out, const float* x, const float* amap,
                                 int64_t map_bstride, const float* noise, float alpha, float scale, int64_t n,
                                 int64_t c, int64_t inner, int64_t noise_bstride, float* scratch,
                                 sr_stream_t stream);
/* Second-order pass of the same tail (path-length regulariser: the backward above is itself differentiated, reference
 * train.py:118-134).  Cotangents of the first-order outputs: Gx [n, c, inner] | NULL, Gmap = (Ga, Gs) planes at
 * Gmap + b * gmap_bstride (+ inner) | NULL, Gb [c] | NULL, Gnw [1] | NULL.  Writes d_gy, d_x [n, c, inner] and the
 * gradient of the scale plane d_amap[b * d_amap_bstride + p] = sum_c (gy * m) * Gx (nothing flows to the shift
 * plane, the bias or the noise weight: the activation mask is piecewise constant).  scratch:
 * sr_noise_bias_act_affine_bwd2_scratch_floats(n, c, inner) floats. */
int64_t sr_noise_bias_act_affine_bwd2_scratch_floats(int64_t n, int64_t c, int64_t inner);
int sr_noise_bias_act_affine_bwd2(float* d_gy, float* d_x, float* d_amap, int64_t d_amap_bstride, const float* Gx,
                                  const float* Gmap, int64_t gmap_bstride, const float* Gb, const float* Gnw,
                                  const float* gy, const float* out, const float* x, const float* amap,
                                  int64_t map_bstride, const float* noise, float alpha, float scale, int64_t n,
                                  int64_t c, int64_t inner, int64_t noise_bstride, float* scratch, sr_stream_t stream);

/* Adam step over ONE flat fp32 parameter buffer (the optimiser of reference train.py:529-536: torch.optim.Adam with
 * the lazy-regularisation corrected lr / betas, no weight decay) in a single pass over p, g, m, v (28 B/parameter):
 *   m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;  p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * `step` points at the device scalar t (a float the caller increments before the call), so the launch can be
 * captured in a hipGraph.  All four buffers 16-byte aligned. */
int sr_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                 float eps, const float* step, sr_stream_t stream);
/* Guarded form for data-parallel training: `guard_offs` (host array, <= 16 entries, read at launch time) are positions
 * of `g` — the first element of every all-reduce bucket.  If any of them is NaN the kernel leaves p / m / v untouched
 * stores 1 into `*skipped_host` (pinned host memory, may be NULL) and takes the caller's increment of `*step` back (a
 * refused step does not advance the bias correction).  The guard samples only these positions: NaN elsewhere in `g` is
 * not detected, and a gradient that is legitimately NaN at a guard position refuses the step as well.  Together with sr_signal_wait_poison this
 * turns a lost bucket signal on ONE rank into a refused optimiser step on EVERY rank (the SUM carries the NaN),
 * where torch DDP's reducer (reference distributed.py:98-105) would raise on the rank that lost it. */
int sr_adam_flat_guarded(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                         float eps, float* step, const int64_t* guard_offs, int n_guards,
                         int32_t* skipped_host, sr_stream_t stream);

/* Row-wise dot products of two [rows, inner] tensors, optionally with a scaled copy in the same
 * sweep: dots[r] = sum_i a[r,i]*b[r,i] ; out_scaled[r,i] = b[r,i]*scale[r] (out_scaled may be NULL).
 * These are the style / demodulation gradients of the modulated convolution (sum_p x*dx', sum_p g*y)
 * that autograd would otherwise compute as a multiply pass plus a reduction pass.  Rows of any length
 * (16-byte vector accesses when inner % 4 == 0 and the pointers are aligned, dword accesses otherwise). */
int64_t sr_rowdot_scratch_floats(int64_t rows, int64_t inner);
int sr_rowdot(float* dots, float* out_scaled, const float* a, const float* b, const float* scale,
              int64_t rows, int64_t inner, float* scratch, sr_stream_t stream);
/* dots[r] = (sum_p a[r,p] * b[r,p]) / rdiv[r] — the demodulation gradient sum_p g * y0 / d of a modulated convolution
 * (reference layers.py:298-300 differentiated) without a separate division launch; same sums, same order as sr_rowdot. */
int sr_rowdot_div(float* dots, const float* a, const float* b, const float* rdiv, int64_t rows, int64_t inner,
                  float* scratch, sr_stream_t stream);
/* Backward of sr_rowdot in one pass (second-order sweep of the path-length regulariser): ga = gd[r] * b,
 * gb = gd[r] * a + go * scale[r], gs[r] = sum_p go * b.  ga / gb / gs / gd / go / scale may be NULL (gs needs go and
 * `scratch` of sr_rowdot_scratch_floats(rows, inner) floats). */
int sr_rowdot_bwd(float* ga, float* gb, float* gs, const float* a, const float* b, const float* gd, const float* go,
                  const float* scale, int64_t rows, int64_t inner, float* scratch, sr_stream_t stream);

/* 1x1 modulated convolution with N <= 4 output channels (ToRGB: reference model.py:56-69 via
 * layers.py:293-323 with kernel_size 1, demodulate off), as streaming passes instead of MFMA tiles.
 * ws [B, N, C] = W[j,c] * style[b,c]; x [B, C, hw]; out / g [B, N, hw]; hw % 4 == 0.
 *   fwd: out[b,j,p] = sum_c ws[b,j,c]*x[b,c,p] + bias[j]      dx: dx[b,c,p] = sum_j ws[b,j,c]*g[b,j,p]
 *   dw:  dws[b,j,c] = sum_p g[b,j,p]*x[b,c,p]  (deterministic two-stage reduction) */
int sr_smallconv_fwd(float* out, const float* x, const float* ws, const float* bias, int64_t B, int64_t C,
                     int64_t N, int64_t hw, sr_stream_t stream);
int sr_smallconv_dx(float* dx, const float* g, const float* ws, int64_t B, int64_t C, int64_t N, int64_t hw,
                    sr_stream_t stream);
/* dx = addend + W^T g: the feature map that feeds ToRGB also feeds the next layer (reference model.py:206-219); its two
 * gradients are added in this pass (addend [B, C, hw] or NULL, may alias dx) */
int sr_smallconv_dx_add(float* dx, const float* g, const float* ws, const float* addend, int64_t B, int64_t C, int64_t N,
                        int64_t hw, sr_stream_t stream);
int64_t sr_smallconv_dw_scratch_floats(int64_t B, int64_t C, int64_t N, int64_t hw);
int sr_smallconv_dw(float* dws, const float* g, const float* x, int64_t B, int64_t C, int64_t N, int64_t hw,
                    float* scratch, sr_stream_t stream);
/* the same with the bias gradient gb[j] = sum_{b,p} g[b,j,p] of ToRGB (reference model.py:57-69) summed by the same
 * launch (gb may be NULL); same scratch */
int sr_smallconv_dw_bias(float* dws, float* gb, const float* g, const float* x, int64_t B, int64_t C, int64_t N, int64_t hw,
                         float* scratch, sr_stream_t stream);
/* Modulated weight rows of that convolution and their pull-back (reference layers.py:293-297, demodulate=False):
 *   sr_modrows_fwd: ws[b,j,c] = (scale * w[j,c]) * s[b,c]                       w [N, C], s [B, C], ws [B, N, C]
 *   sr_modrows_bwd: gs[b,c] = sum_j dws[b,j,c] * (scale * w[j,c]);  gw[j,c] = scale * sum_b dws[b,j,c] * s[b,c]
 *                   (gs or gw may be NULL).  Fixed summation order. */
int sr_modrows_fwd(float* ws, const float* w, const float* s, float scale, int64_t B, int64_t N, int64_t C,
                   sr_stream_t stream);
int sr_modrows_bwd(float* gs, float* gw, const float* dws, const float* w, const float* s, float scale, int64_t B,
                   int64_t N, int64_t C, sr_stream_t stream);

/* Vertex normals of a posed mesh (replaces reference utils_3d.py:379-404 mesh_point_normal: three
 * sparse.mm scatters + layers.py:13-34 Normalize).  v [B, nv, 3]; tri [nf, 3] int64 with ids in
 * [0, nv); (adj_off [nv + 1], adj [3 nf]) = CSR list of the corner-major incidences k*nf + f of every
 * vertex, ascending (the caller builds it once per topology).  vn [B, nv, 3] = sum of incident face
 * normals (b-a)x(c-a), divided by max(|.|, eps); norm_out [B, nv] (may be NULL) = the clamped length.
 * Gather in a fixed order: deterministic, no atomics. */
int sr_vertex_normals_f32(float* vn, float* norm_out, const float* v, const int64_t* tri, const int32_t* adj_off,
                          const int32_t* adj, int64_t B, int64_t nv, int64_t nf, float eps, sr_stream_t stream);

/* Posed mesh of the inversion / training loops (reference utils_3d.py random_apply_pose3D: vertices @ R * s + t):
 *   out[b, i, :] = v[b, i, :] @ M[b] + t[b]     v [B or 1, nv, 3] (v_bstride = nv*3, or 0 to share one mesh), M [B,3,3]
 * row-major, t [B,3] or NULL.  sr_affine3_bwd: gm[b] = v[b]^T @ g[b] ([3,3]), gt[b] = sum_i g[b, i, :] — fixed-order
 * sums (run-to-run identical); either may be NULL.  The gradient w.r.t. v is g @ M^T = sr_affine3_fwd(g, M^T). */
int sr_affine3_fwd(float* out, const float* v, const float* m, const float* t, int64_t B, int64_t nv,
                   int64_t v_bstride, sr_stream_t stream);
int sr_affine3_bwd(float* gm, float* gt, const float* v, const float* g, int64_t B, int64_t nv, int64_t v_bstride,
                   sr_stream_t stream);

/* Pose parameters of the inversion loop, pose = (yaw, pitch, roll, tx, ty, tz, log-scale): rot [3,3] = Rz(roll) Rx(pitch)
 * Ry(yaw) (utils_3d.euler_mat(angles, "yxz"), row-major), lin = exp(log-scale) * rot; sr_pose_bwd: gradient of the seven
 * numbers from the gradients of the two matrices (either may be NULL; entries 3..5 are written as 0: the translation's
 * gradient is sr_affine3_bwd's gt).  One lane each: replaces ~60 one-element tensor-algebra launches per step. */
int sr_pose_fwd(float* lin, float* rot, const float* pose, sr_stream_t stream);
int sr_pose_bwd(float* gpose, const float* glin, const float* grot, const float* pose, sr_stream_t stream);
/* B poses at once, forward only: lin [B, 3, 3], rot [B, 3, 3] or NULL, pose [B, 7].  The random poses of the training
 * loop (reference utils_3d.py:360-376 random_apply_pose3D: euler_mat + scale for a batch of sampled poses). */
int sr_pose_batch_fwd(float* lin, float* rot, const float* pose, int64_t B, sr_stream_t stream);

/* Skinny linear algebra of the style path, B = per-GPU batch rows (csrc/style_linear.hip).
 * EqualLinear (reference layers.py:222-248), optionally with the fused leaky-ReLU of the mapping
 * network (act != 0: op/fused_act.py:86-97 semantics, bias inside the activation):
 *   sr_linear_fwd:   y[b,n] = act(wscale * sum_k x[b,k]*w[n,k] + bscale*bias[n])      (bias may be NULL)
 *   sr_linear_bwd_x: gx[b,k] = wscale * sum_n g'[b,n]*w[n,k],  g' = act'(y) * gy  (y = saved output)
 *   sr_linear_bwd_w: gw[n,k] = wscale * sum_b g'[b,n]*x[b,k];  gbias[n] = bscale * sum_b g'[b,n] (or NULL)
 * Demodulation scale of ModulatedConv2d (layers.py:298-300 as rsqrt(s^2 @ wsq + eps), wsq from
 * sr_weight_prep):
 *   sr_demod_fwd: d[b,co] = rsqrt(sum_ci s[b,ci]^2 * wsq[ci,co] + eps)
 *   sr_demod_bwd: gq = -gd*d^3/2;  gs[b,ci] = gs_add[b,ci] + 2*s[b,ci]*sum_co gq[b,co]*wsq[ci,co]
 *                 (gs_add may be NULL);  gwsq[ci,co] = sum_b s[b,ci]^2 * gq[b,co]   (gs or gwsq may be NULL)
 * x rows have pitch ldx floats (a [B, n_latent, K] latent slice is read in place); everything else is
 * dense.  K, ldx (resp. Co) must be multiples of 4 and the matrices 16-byte aligned (SR_EINVAL otherwise). */
int sr_linear_fwd(float* y, const float* x, const float* w, const float* bias, int64_t B, int64_t K, int64_t N,
                  int64_t ldx, float wscale, float bscale, int act, float alpha, float gain, sr_stream_t stream);
int sr_linear_bwd_x(float* gx, const float* gy, const float* y, const float* w, int64_t B, int64_t K, int64_t N,
                    float wscale, int act, float alpha, float gain, sr_stream_t stream);
int sr_linear_bwd_w(float* gw, float* gbias, const float* gy, const float* y, const float* x, int64_t B,
                    int64_t K, int64_t N, int64_t ldx, float wscale, float bscale, int act, float alpha,
                    float gain, sr_stream_t stream);
int sr_demod_fwd(float* d, const float* s, const float* wsq, int64_t B, int64_t Ci, int64_t Co, float eps,
                 sr_stream_t stream);
int sr_demod_bwd(float* gs, float* gwsq, const float* gd, const float* gs_add, const float* s, const float* d,
                 const float* wsq, int64_t B, int64_t Ci, int64_t Co, sr_stream_t stream);

/* Weight preparation for sr_conv2d_mfma (replaces the per-layer ATen passes of reference
 * layers.py:293-300 / 213-216: scale * weight, pow, sum, views).  w [Co, Ci, k, k], k in {1, 3}.
 *   sr_weight_prep:     wt [k*k, Ci, ld] = scale*w  (ld % 4 == 0, ld >= Co, pad columns zeroed);
 *                       wsq [Ci, Co] = sum_taps (scale*w)^2, or NULL.
 *   sr_weight_prep_bwd: gw [Co,Ci,k,k] = scale*gwt^T + 2*scale^2*w*gwsq  (gwt [k*k,Ci,ldg] or NULL,
 *                       gwsq [Ci,Co] or NULL, not both NULL).
 *   sr_weight_adjoint:  out [taps, N, ldc] = in [taps', C, ldn] transposed in (C, N), taps reversed
 *                       when flip != 0 — the weights of the data-gradient convolution. */
int sr_weight_prep(float* wt, float* wsq, const float* w, float scale, int64_t Co, int64_t Ci, int ksize,
                   int64_t ld, sr_stream_t stream);
int sr_weight_prep_bwd(float* gw, const float* gwt, const float* gwsq, const float* w, float scale,
                       int64_t Co, int64_t Ci, int ksize, int64_t ldg, sr_stream_t stream);
int sr_weight_adjoint(float* out, const float* in, int64_t taps, int64_t C, int64_t N, int64_t ldn,
                      int64_t ldc, int flip, sr_stream_t stream);

/* Batched forms of the two preparations a forward pass needs: n layers per call, per-layer arguments as host arrays
 * (pointers to device tensors, shapes); one launch per 48 layers instead of one per layer — a training iteration at 4
 * images per GPU prepares ~230 weights in its forward passes, each a 5 us launch (stylerenderer_amd/op/weight_bank.py).
 * Same arithmetic per element as the single-layer entry points: results are bit-identical.  wsq[i] may be NULL. */
int sr_weight_prep_batch(int n, float* const* wt, float* const* wsq, const float* const* w, const float* scale,
                         const int64_t* Co, const int64_t* Ci, const int* ksize, const int64_t* ld, sr_stream_t stream);
int sr_weight_adjoint_batch(int n, float* const* out, const float* const* in, const int64_t* taps, const int64_t* C,
                            const int64_t* N, const int64_t* ldn, const int64_t* ldc, const int* flip,
                            sr_stream_t stream);

/* Skinny products of ALL modulated layers of a pass, one launch per call (csrc/bank_mm.hip, op/bankmm.py): the
 * modulation s_l = EqualLinear_l(latent row) of every ModulatedConv2d and the demodulation sums q_l = s_l^2 @ Wsq_l
 * (reference layers.py:222-248, 293-300) and their gradients of any order.  A call takes a TABLE of problems: host
 * arrays of device pointers and extents (<= SR_BANK_MAX per launch; longer tables are cut into several launches).
 *   sr_bank_nt: out_p[b,n] = alpha * sum_k A_p[b,k] * M_p[n,k] + bscale * bias_p[n]      A_p rows of pitch lda[p]
 *               (a latent row of [B, n_latent, K] is read in place), M_p [N_p, K_p] dense, out_p rows of pitch ldo[p];
 *               bias (the array or single entries) may be NULL.
 *   sr_bank_nn: out_o[b,j] = alpha * sum over the n_terms[o] terms t of output o, in table order, of
 *               sum_i A_t[b,i] * M_t[i,j]       (M_t [I_t, J_o] dense; the term arrays A / M / lda / I are the
 *               concatenation of all outputs' terms).  Several layers reading one latent row add their gradients here.
 *   sr_bank_tn: out_p[i,j] = alpha * sum_b A_p[b,i] * C_p[b,j] (dense [I_p, J_p]);  col_p[i] = bscale * sum_b A_p[b,i]
 *               (col, or single entries, may be NULL).
 * K, J, the pitches of A (nt) / C / out (nn) multiples of 4 and those arrays 16-byte aligned (SR_EINVAL otherwise);
 * B <= 65535.  Fixed summation order: deterministic. */
#define SR_BANK_MAX 48
int sr_bank_nt(int n, float* const* out, const float* const* A, const float* const* M, const float* const* bias,
               const int64_t* lda, const int64_t* ldo, const int64_t* K, const int64_t* N, int64_t B, float alpha,
               float bscale, sr_stream_t stream);
int sr_bank_nn(int n_out, float* const* out, const int64_t* ldo, const int64_t* J, const int* n_terms,
               const float* const* A, const float* const* M, const int64_t* lda, const int64_t* I, int64_t B,
               float alpha, sr_stream_t stream);
int sr_bank_tn(int n, float* const* out, float* const* col, const float* const* A, const float* const* C,
               const int64_t* lda, const int64_t* ldc, const int64_t* I, const int64_t* J, int64_t B, float alpha,
               float bscale, sr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * upfirdn2d: zero-insert upsample -> pad/crop -> 2-D FIR (correlation with the flipped kernel)
 * -> decimate.  Replaces  bool upfirdn2d_op(float* out, const float* x, const float* k,
 *             UpFirDn2DKernelParams& p, int mode, int use_cuda)
 * reference op/upfirdn2d_kernel.cu:205-206 with the struct of op/upfirdn2d.cpp:2-22 flattened into
 * scalars (minor_dim is always 1 in the reference's callers, op/upfirdn2d.py:99,122, and is not
 * carried).  x is [major, in_h, in_w], out is [major, out_h, out_w] with
 * out = (in*up + pad0 + pad1 - k) / down + 1  (op/upfirdn2d.py:103-104); the caller passes the
 * out sizes it allocated and they are checked.  `k` is the [kh, kw] kernel in DEVICE memory. */
int sr_upfirdn2d(float* out, const float* x, const float* k, int64_t major, int in_h, int in_w,
                 int out_h, int out_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                 int pad_x0, int pad_x1, int pad_y0, int pad_y1, sr_stream_t stream);

/* ToRGB skip connection (reference model.py:66-68: `out + self.upsample(skip)`, Upsample = upfirdn2d with up = 2 and the
 * 4x4 kernel, layers.py:170-181) in one pass: out = upfirdn2d(x, k, up = 2, pad = (pad0, pad1)) + addend.
 * x [major, in_h, in_w]; out / addend [major, out_h, out_w], out_h = 2 in_h + pad0 + pad1 - 3.  Same arithmetic
 * as sr_upfirdn2d followed by an addition (IEEE addition commutes), bit for bit. */
int sr_upsample2_add(float* out, const float* x, const float* k, const float* addend, int64_t major, int in_h,
                     int in_w, int out_h, int out_w, int pad0, int pad1, sr_stream_t stream);

/* Blur (4x4 FIR, up = down = 1, pad (pad0, pad1) on both axes: reference layers.py:194-203 after the
 * stride-2 transposed convolution) with the StyledConv tail fused into its store (reference
 * model.py:26-32): y = lrelu((fir(x) + noise_w[0]*noise[b, p]) + bias[ch], alpha) * gain.
 * x [n, c, in_h, in_w] -> y [n, c, out_h, out_w], out = in + pad0 + pad1 - 3; noise [n or 1, 1, out_h, out_w]
 * with batch stride noise_bstride (0 = shared) or NULL; bias [c] or NULL.  One pass instead of two. */
int sr_blur_noise_bias_act(float* y, const float* x, const float* k, const float* noise, const float* noise_w,
                           const float* bias, float alpha, float gain, int64_t n, int64_t c, int in_h, int in_w,
                           int out_h, int out_w, int pad0, int pad1, int64_t noise_bstride, sr_stream_t stream);

/* First-order backward of sr_blur_noise_bias_act in ONE pass (the gradient of reference layers.py:194-203 + model.py:26-32
 * taken together): gx = blur^T(lrelu'(y) * gy * gain) [n, c, out_h, out_w], gbias [c] = sum lrelu'(y) gy gain, gnoise_w [1]
 * = the same sum weighted with the noise (both NULL when bias / noise strength are frozen), rowdot [n * c] = the sum
 * weighted with the pre-activation value rebuilt from y (the demodulation gradient of the convolution in front, see
 * sr_noise_bias_act_bwd_dot).  gy, y [n, c, in_h, in_w] = the forward's OUTPUT extent, out = in + 3 - 2 pad0 its input
 * extent (pad0 = the forward's symmetric padding); k_flipped = the forward taps flipped in both axes (what sr_upfirdn2d
 * takes for the gradient of a blur).  gx is bit-identical to sr_noise_bias_act_bwd followed by sr_upfirdn2d; the three
 * sums are deterministic (per-workgroup partials added in a fixed order) but associate differently from
 * sr_noise_bias_act_bwd_dot.  scratch: sr_blur_nba_bwd_scratch_floats floats. */
int64_t sr_blur_nba_bwd_scratch_floats(int64_t n, int64_t c, int out_h, int out_w);
int sr_blur_nba_bwd(float* gx, float* gbias, float* gnoise_w, float* rowdot, const float* gy, const float* y,
                    const float* k_flipped, const float* noise, const float* noise_w, const float* bias, float alpha,
                    float gain, int64_t n, int64_t c, int in_h, int in_w, int out_h, int out_w, int pad0,
                    int64_t noise_bstride, float* scratch, sr_stream_t stream);


/* ---------------------------------------------------------------------------------------
 * 3DMM triangle rasterizer (z-buffered, deterministic; results equal the reference's
 * SEQUENTIAL CPU loops bit for bit: op/rasterize.cpp:21-67).
 * Replaces  bool rasterize_gpu<scalar,index>(b, nv, nf, h, w, repeat_v, repeat_f, perspective,
 *             const scalar* v, const index* f, index* i, scalar* c, scalar* zB, scalar eps)
 * reference op/rasterize.cu:84-88.
 *   v     [b, nv, 3] (or [nv, 3] when repeat_v)         tri  int64 [nf, 3] (or [b, nf, 3])
 *   index int64 [b, h, w, 3]   coeff [b, h, w, 3]       zbuf [b, h, w] or NULL
 * Unlike the reference the outputs need NOT be pre-initialised: every pixel is written
 * (uncovered: index 0, coeff 0, zbuf -MAX).  `work` is caller scratch of
 * sr_rasterize_scratch_bytes(b, nf, h, w, is_double) bytes.
 * Optional fused attribute interpolation (reference op/rasterize.py:29-37): when `tex` != NULL,
 * attr[b,h,w,c] = sum_k tex[index_k, :] * coeff_k  with tex [b*nv (or nv), c].
 * Optional winner map `win` int32 [b,h,w]: id of the triangle that owns the pixel, -1 where uncovered
 * (what sr_rasterize_grad_* walks; index / coeff may then be NULL: the fused autograd path writes
 * 16 B per pixel instead of 48).  Triangles whose bounding box exceeds 64 pixels are walked by the whole
 * workgroup (LDS queue) instead of one lane; optional `big` int32 [2 + 2*b*nf] is the gradient state of the call for
 * sr_rasterize_grad_*: their count, b*nf slots for the flat ids sample * nf + triangle (any order), b*nf slots
 * for the leader table (smallest row-major pixel each triangle won) and ONE state word: the LDS-tiled forward path
 * builds the table while it resolves its tiles and stores 1, the global-key path fills it with INT_MAX and stores 0 —
 * the gradient pass reads that word on the device and completes the table in place when it is 0 (it never re-derives
 * which path the forward took).
 * `perspective` is a flags word: bit 0 = perspective projection (the reference's bool), bit 1 = SR_RASTER_CHW: the
 * interpolated attributes are written channel-major, attr[b,c,h,w] — what the generator's map heads convolve — instead of
 * the reference's [b,h,w,c] (model.py:262 permutes and every consumer re-lays it out); sr_rasterize_grad_* with the same
 * bit reads grad_out as [b,c,h,w].  Same values either way.  Bit 2 = SR_RASTER_GRAD_ACC (sr_rasterize_grad_* only):
 * grad_v / grad_tex are ADDED to what the buffers hold — one mesh rasterised at several resolutions (GeneratorWithMap,
 * reference model.py:255-262) sums its gradients inside the gather, in call order, instead of in a tensor addition per
 * resolution. */
#define SR_RASTER_CHW 2
#define SR_RASTER_GRAD_ACC 4
int64_t sr_rasterize_scratch_bytes(int64_t b, int64_t nf, int64_t h, int64_t w, int is_double);
int sr_rasterize_forward_f32(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w, int repeat_v,
                             int repeat_f, int perspective, const float* v, const int64_t* tri,
                             int64_t* index, float* coeff, float* zbuf, float eps,
                             const float* tex, int64_t tex_c, float* attr, int32_t* win, int32_t* big,
                             void* work, sr_stream_t stream);
int sr_rasterize_forward_f64(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w, int repeat_v,
                             int repeat_f, int perspective, const double* v, const int64_t* tri,
                             int64_t* index, double* coeff, double* zbuf, double eps,
                             const double* tex, int64_t tex_c, double* attr, int32_t* win, int32_t* big,
                             void* work, sr_stream_t stream);

/* The same mesh rasterised at n resolutions (GeneratorWithMap draws one normal map per synthesis resolution, reference
 * model.py:255-262) in THREE launches instead of three per resolution: interpolated attributes only (attr[l]:
 * [b,h_l,w_l,c] or, with SR_RASTER_CHW, [b,c,h_l,w_l]) plus the gradient state win[l] / big[l] of each level (the arrays, or
 * single entries, may be NULL: no gradient state); work[l] >= sr_rasterize_scratch_bytes(b, nf, h_l, w_l, 0) bytes each.
 * Same kernels' bodies as sr_rasterize_forward_f32 on its global-key path: same bits.  Levels the per-call dispatcher would
 * give to the LDS-tiled path are not taken (sr_rasterize_levels_supported returns 0, the call SR_EINVAL): the caller then
 * goes level by level. */
#define SR_RASTER_MAX_LEVELS 12
int sr_rasterize_levels_supported(int n, int64_t b, int64_t nf, const int64_t* h, const int64_t* w);
int sr_rasterize_forward_levels_f32(int n, int64_t b, int64_t nv, int64_t nf, const int64_t* h, const int64_t* w,
                                    int repeat_v, int repeat_f, int perspective, const float* v, const int64_t* tri,
                                    float eps, const float* tex, int64_t tex_c, float* const* attr, int32_t* const* win,
                                    int32_t* const* big, void* const* work, sr_stream_t stream);

/* Gradient of those n levels in FOUR launches: grad_v / grad_tex = the sum over the levels (added in table order — the
 * value n sr_rasterize_grad_f32 calls with SR_RASTER_GRAD_ACC from the second on produce, bit for bit) of the gradient of
 * level l's attribute map w.r.t. the vertices / attribute rows, given grad_out[l] ([b,h_l,w_l,c], or [b,c,h_l,w_l] with
 * SR_RASTER_CHW) and the level's gradient state win[l] / big[l].  work[l] >= sr_rasterize_grad_scratch_bytes(b, nf, c, 0)
 * each.  fp32, c <= 4, nf > 0 and the cached inverse incidence table adj_slot are required (SR_EINVAL otherwise: the
 * caller accumulates level by level). */
int sr_rasterize_grad_levels_f32(int n, int64_t b, int64_t nv, int64_t nf, const int64_t* h, const int64_t* w, int repeat_f,
                                 int perspective, const float* v, const float* tex, int64_t tex_c, const int64_t* tri,
                                 const int32_t* const* win, int32_t* const* big, const float* const* grad_out,
                                 const int32_t* adj_off, const int32_t* adj, int64_t off_bstride, int64_t adj_bstride,
                                 const int32_t* adj_slot, float* grad_v, float* grad_tex, float eps, void* const* work,
                                 sr_stream_t stream);

/* d(coeff)/d(vertex) per pixel.  Replaces  bool rasterize_gpu_backward<scalar,index>(b, n, h, w,
 *   repeat_v, perspective, const scalar* v, const index* i, scalar* dcoeff, scalar eps)
 * reference op/rasterize.cu:124-127 (arithmetic op/rasterize.h:169-228).  dcoeff [b,h,w,3,9];
 * every pixel is written (zeros where the reference leaves its torch::zeros untouched). */
int sr_rasterize_backward_f32(int64_t b, int64_t n, int64_t h, int64_t w, int repeat_v,
                              int perspective, const float* v, const int64_t* index, float* dcoeff,
                              float eps, sr_stream_t stream);
int sr_rasterize_backward_f64(int64_t b, int64_t n, int64_t h, int64_t w, int repeat_v,
                              int perspective, const double* v, const int64_t* index,
                              double* dcoeff, double eps, sr_stream_t stream);

/* Fused backward of `rasterize` (reference op/rasterize.py:39-80: dcoeff, [1x3]@[3x9], and two
 * sparse scatter matmuls) without materialising dcoeff or a COO matrix, and WITHOUT atomics — a
 * deterministic two-phase gather (run-to-run identical results):
 *   phase 1  per (sample, triangle): sums over the pixels the triangle won (`win` of the forward call), in
 *            pixel order, of (grad_out . tex[vertex_i]) * dcoeff[i, :] and grad_out * coeff_k — pixel-parallel
 *            with the triangle's first pixel as its leader; workgroup-cooperative for the triangles in `big`;
 *   phase 2  per (sample, vertex): sum over its incident triangles in the order of the incidence list.
 *   Between the phases the corner values live in VERTEX-MAJOR slots: position e of the (vertex-sorted) incidence list is
 *   the slot of that corner, so phase 1 scatters each corner to adj_slot[k * nf + f] and phase 2 reads ONE contiguous
 *   span per vertex (consecutive vertices: consecutive spans) instead of gathering scattered per-triangle records.
 *   grad_v  [b, nv, 3], grad_tex [b, nv, c]: every row is written (no pre-zeroing); either may be NULL.
 * v [b, nv, 3]; tex [b, nv, c]; tri int64 [nf, 3] (or [b, nf, 3] when !repeat_f); grad_out [b, h, w, c].
 * Incidence list (CSR, built once per topology by the caller): adj_off int32 [nv + 1], adj int32 [3 nf] holding
 * corner-major entries k * nf + f with tri[f][k] == vertex, ascending; per-sample topologies pass batch strides
 * (elements) adj_off_bstride / adj_bstride, shared ones pass 0.  adj_slot int32 [3 nf] (batch stride adj_bstride) is
 * the inverse permutation, adj_slot[adj[e]] = e, built once per topology next to the list; NULL = rebuilt by every
 * call inside `work` (one extra launch).
 * `work`: sr_rasterize_grad_scratch_bytes(b, nf, c, is_double) bytes. */
int64_t sr_rasterize_grad_scratch_bytes(int64_t b, int64_t nf, int64_t tex_c, int is_double);
int sr_rasterize_grad_f32(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w, int repeat_f,
                          int perspective, const float* v, const float* tex, int64_t tex_c,
                          const int64_t* tri, const int32_t* win, const int32_t* big, const float* grad_out,
                          const int32_t* adj_off, const int32_t* adj, int64_t adj_off_bstride,
                          int64_t adj_bstride, const int32_t* adj_slot, float* grad_v, float* grad_tex, float eps,
                          void* work, sr_stream_t stream);
int sr_rasterize_grad_f64(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w, int repeat_f,
                          int perspective, const double* v, const double* tex, int64_t tex_c,
                          const int64_t* tri, const int32_t* win, const int32_t* big, const double* grad_out,
                          const int32_t* adj_off, const int32_t* adj, int64_t adj_off_bstride,
                          int64_t adj_bstride, const int32_t* adj_slot, double* grad_v, double* grad_tex, double eps,
                          void* work, sr_stream_t stream);

/* Host (CPU) path for HOST pointers: the reference's extension also serves CPU tensors
 * (rasterize_cpu / rasterize_cpu_backward, reference op/rasterize.cpp:21-95, dispatched at :126-150).
 * Sequential loops on the same arithmetic as the kernels (this file's functions are __host__ __device__);
 * outputs are initialised by the callee; zbuf may be NULL.  No stream: runs on the calling thread. */
int sr_rasterize_forward_cpu_f32(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w, int repeat_v,
                                 int repeat_f, int perspective, const float* v, const int64_t* tri,
                                 int64_t* index, float* coeff, float* zbuf, float eps);
int sr_rasterize_forward_cpu_f64(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w, int repeat_v,
                                 int repeat_f, int perspective, const double* v, const int64_t* tri,
                                 int64_t* index, double* coeff, double* zbuf, double eps);
int sr_rasterize_backward_cpu_f32(int64_t b, int64_t n, int64_t h, int64_t w, int perspective,
                                  const float* v, const int64_t* index, float* dcoeff, float eps);
int sr_rasterize_backward_cpu_f64(int64_t b, int64_t n, int64_t h, int64_t w, int perspective,
                                  const double* v, const int64_t* index, double* dcoeff, double eps);

/* ---------------------------------------------------------------------------------------
 * Dense contraction of the modulated convolution on the matrix cores (exact fp32 MFMA).
 * The reference has no native code here: it calls torch's F.conv2d / F.conv_transpose2d with
 * groups = batch on per-sample weight copies (reference layers.py:293-323).  These entry points
 * are what a host binding calls instead.
 *
 *   out[b,n,oy,ox] = oscale[b,n] * sum_{ky,kx,c} wt[ky*k+kx][c][n] * iscale[b,c] * in[b,c,iy,ix] + obias[n]
 *
 *   in  [B, C, IH, IW]   out [B, N, OH, OW]   wt [k*k, C, wt_ld] (row pitch wt_ld >= N, a multiple
 *   of 4 floats, base 16-byte aligned; columns N..wt_ld-1 are padding and may hold anything)
 *   iscale [B, C] | NULL   oscale [B, N] | NULL   obias [N] | NULL
 *   transposed = 0: correlation, iy = oy*stride + ky - pad   (k in {1,3}, stride in {1,2})
 *   transposed = 1: k = 3, stride = 2, pad = 0: out[2y+ky, 2x+kx] += in[y,x] * wt[ky*3+kx]
 *                   (OH = 2*IH + 1), the stride-2 transposed conv of the upsampling layers.
 *   scratch: sr_conv2d_scratch_floats(...) floats (0 for large maps) for the split-K partial sums
 *   of small feature maps, reduced in fixed order (deterministic); NULL disables the split. */
int64_t sr_conv2d_scratch_floats(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW, int64_t OH,
                                 int64_t OW, int ksize, int stride, int pad, int transposed);
int sr_conv2d_mfma(float* out, const float* in, const float* wt, const float* iscale,
                   const float* oscale, const float* obias, int64_t B, int64_t C, int64_t N,
                   int64_t wt_ld, int64_t IH, int64_t IW, int64_t OH, int64_t OW, int ksize, int stride,
                   int pad, int transposed, float* scratch, sr_stream_t stream);
/* The modulated 3x3 stride-1 pad-1 convolution with the StyledConv tail (reference model.py:26-32) fused into
 * its store: out = lrelu((oscale*conv(iscale*in, wt) + noise_w[0]*noise[b, p]) + abias[n], alpha) * gain.
 * Served by the Winograd kernel only: SR_EINVAL when the shape is not eligible (H % 8, W % 32, C % 8, N % 64,
 * C <= 512) — the caller then uses sr_conv2d_mfma + sr_noise_bias_act.  scratch: sr_conv2d_scratch_floats. */
int sr_conv2d_nba(float* out, const float* in, const float* wt, const float* iscale, const float* oscale,
                  const float* noise, const float* noise_w, const float* abias, float alpha, float gain, int64_t B,
                  int64_t C, int64_t N, int64_t wt_ld, int64_t H, int64_t W, int64_t noise_bstride, float* scratch,
                  sr_stream_t stream);

/* The same two entry points with a `flags` word.  SR_CONV_U_READY: `scratch` is the buffer an EARLIER call with the same
 * `wt`, C, N (and the same call geometry) used and nobody has written since — its leading block still holds the
 * Winograd-domain weights, so their preparation (k_wino_weights, one launch per call) is skipped.  For frozen networks
 * (latent inversion, sampling): the caller keeps one scratch per (weight, geometry).  Ignored on the direct path. */
#define SR_CONV_U_READY 1
/* 1 when a stride-1 3x3 call with these sizes and THESE buffers (16-byte alignment of in / out is part of the rule) is
 * served by the Winograd kernel, 0 when by the direct one — what a caller that keeps a scratch for SR_CONV_U_READY
 * must ask before it marks the scratch's weight block as written. */
int sr_conv2d_uses_winograd(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW, const float* in, const float* out);

/* Generic-geometry convolution (csrc/conv_generic.hip): any kernel extent kh x kw, stride (sy, sx) and zero padding
 * (py, px), dilation 1, one group, weights in the reference's layout [N, C, kh, kw] — what the reference's EqualConv2d
 * (layers.py:204-221: F.conv2d with any kernel_size / stride / padding) and ModulatedConv2d with kernel_size other than
 * 1 or 3 (layers.py:259-323, F.conv2d / F.conv_transpose2d) reach outside the matrix-core geometries above; replaces the
 * MIOpen call a device tensor would otherwise make.  Direct fp32 kernels, fixed summation order.
 *   sr_conv2d_generic        y [B,N,OH,OW] = conv(x [B,C,IH,IW], w) (+ bias[N] or NULL); OH = (IH + 2 py - kh) / sy + 1
 *   sr_conv2d_generic_dgrad  dx [B,C,IH,IW] = its gradient w.r.t. x for g [B,N,OH,OW]; ALSO F.conv_transpose2d(g, w,
 *                            stride, padding) as a forward operator (w then reads [in = N, out = C, kh, kw])
 *   sr_conv2d_generic_wgrad  dw [N,C,kh,kw] = its gradient w.r.t. w
 * SR_EINVAL when (OH, OW) is not the convolution's extent for (IH, IW); SR_ERANGE past 65535 (sample, channel) planes. */
int sr_conv2d_generic(float* y, const float* x, const float* w, const float* bias, int64_t B, int64_t C, int64_t N,
                      int64_t IH, int64_t IW, int64_t OH, int64_t OW, int kh, int kw, int sy, int sx, int py, int px,
                      sr_stream_t stream);
int sr_conv2d_generic_dgrad(float* dx, const float* g, const float* w, int64_t B, int64_t C, int64_t N, int64_t IH,
                            int64_t IW, int64_t OH, int64_t OW, int kh, int kw, int sy, int sx, int py, int px,
                            sr_stream_t stream);
int sr_conv2d_generic_wgrad(float* dw, const float* x, const float* g, int64_t B, int64_t C, int64_t N, int64_t IH,
                            int64_t IW, int64_t OH, int64_t OW, int kh, int kw, int sy, int sx, int py, int px,
                            sr_stream_t stream);
int sr_conv2d_mfma_ex(float* out, const float* in, const float* wt, const float* iscale,
                      const float* oscale, const float* obias, int64_t B, int64_t C, int64_t N,
                      int64_t wt_ld, int64_t IH, int64_t IW, int64_t OH, int64_t OW, int ksize,
                      int stride, int pad, int transposed, int flags, float* scratch, sr_stream_t stream);
/* out = oscale[b,n] * conv1x1(in, wt)[b,n,p] + addend[b,n,p]  (stride 1, no window; P = pixels per plane): the skip branch of the
 * discriminator's ResBlock with the sum of the two branches in its store (reference model.py: ResBlock.forward
 * `(out + skip) / sqrt(2)`, the factor folded into oscale and the other branch's gain).  _supported: 1 when the GEMM-shaped
 * kernel takes the call (C % 16, N % 128, P % 128, 16-byte aligned operands, enough tiles), else the caller adds separately. */
int sr_conv1x1_add_supported(int64_t B, int64_t C, int64_t N, int64_t wt_ld, int64_t P, const float* in, const float* wt,
                             const float* out, const float* addend);
int sr_conv1x1_add(float* out, const float* in, const float* wt, const float* oscale, const float* addend, int64_t B,
                   int64_t C, int64_t N, int64_t wt_ld, int64_t P, sr_stream_t stream);
int sr_conv2d_nba_ex(float* out, const float* in, const float* wt, const float* iscale, const float* oscale,
                     const float* noise, const float* noise_w, const float* abias, float alpha, float gain,
                     int64_t B, int64_t C, int64_t N, int64_t wt_ld, int64_t H, int64_t W, int64_t noise_bstride,
                     int flags, float* scratch, sr_stream_t stream);

/* Weight gradient of sr_conv2d_mfma (same geometry arguments):
 *   dwt[ky*k+kx][c][n] = sum_{b, pixels} (xscale[b,c] * x[b,c,window]) * (gscale[b,n] * gy[b,n,pixel])
 * x [B,C,IH,IW], gy [B,N,OH,OW], dwt [k*k, C, N]; xscale / gscale may be NULL.  The batch is folded
 * into the reduction (the reference's grouped convolution would give per-sample gradients).
 * `scratch`: sr_conv2d_wgrad_scratch_floats(...) floats of split-K partial sums, reduced in a
 * fixed order (deterministic). */
int64_t sr_conv2d_wgrad_scratch_floats(int64_t B, int64_t C, int64_t N, int64_t IH, int64_t IW,
                                       int64_t OH, int64_t OW, int ksize, int stride, int pad,
                                       int transposed);
int sr_conv2d_wgrad_mfma(float* dwt, const float* x, const float* gy, const float* xscale,
                         const float* gscale, int64_t B, int64_t C, int64_t N, int64_t IH,
                         int64_t IW, int64_t OH, int64_t OW, int ksize, int stride, int pad,
                         int transposed, float* scratch, sr_stream_t stream);

/* ---- signalling out of a replayed hipGraph (overlapped gradient all-reduce) -----------------------------------
 * The reference overlaps the gradient all-reduce with the backward through torch DDP's bucket hooks
 * (reference distributed.py:98-105, used for the path-length step at train.py:335-352).  This build replays each
 * training phase as ONE hipGraph; a bucket of the flat gradient buffer that is complete in the middle of that graph
 * is announced by a one-lane kernel node.  sr_signal_set stores `*epoch` (a device word the host writes, in stream
 * order, in front of every replay it launches) into `*word` once all work enqueued before it on `stream` has finished;
 * sr_signal_bump is the incrementing form (`*counter += 1`).  sr_signal_wait enqueues a one-lane kernel on another
 * stream that returns when `*counter - at_least >= 0` (signed 32-bit distance): with at_least = the epoch of the replay
 * just launched, everything enqueued behind it starts at that point of the replay.  The producer must already be
 * enqueued when the wait is (the wait spins on the device).  sr_signal_wait_timeout gives up after `timeout_us`
 * (0 = never): it stores `code` into `*status_host` (pinned host memory the caller polls; may be NULL) and returns, so
 * a lost signal becomes a host-side error instead of a silent device-side spin. */
int sr_signal_bump(uint32_t* counter, sr_stream_t stream);
int sr_signal_set(uint32_t* word, const uint32_t* epoch, sr_stream_t stream);
int sr_signal_wait(const uint32_t* counter, uint32_t at_least, sr_stream_t stream);
int sr_signal_wait_timeout(const uint32_t* counter, uint32_t at_least, uint64_t timeout_us, int32_t* status_host,
                           int32_t code, sr_stream_t stream);
/* As sr_signal_wait_timeout; on expiry `*poison` (a float of the data the wait guards, may be NULL) is overwritten
 * with NaN before the kernel returns: what is queued behind the wait then carries a marker that
 * sr_adam_flat_guarded refuses. */
int sr_signal_wait_poison(const uint32_t* counter, uint32_t at_least, uint64_t timeout_us, int32_t* status_host,
                          int32_t code, float* poison, sr_stream_t stream);
/* sr_signal_set for a word in pinned HOST memory (system-scope store): the host-released mode of the bucketed reducer
 * polls it from the CPU and queues the collective itself — no kernel ever spins on the communication stream, so the
 * mode does not depend on the two streams owning separate hardware queues (GPU_MAX_HW_QUEUES). */
int sr_signal_set_host(uint32_t* word_host, const uint32_t* epoch, sr_stream_t stream);

/* Repairs a captured, not yet instantiated hipGraph_t for the HIP 7.0 runtime PyTorch-ROCm 2.10 bundles: memset nodes
 * replay a corrupted value from the second launch on (torch's multi-block reductions zero their semaphores that way),
 * so each memset node is replaced by a fill-kernel node with the same edges.  `replaced` receives the count.  The
 * reference has no counterpart: it never captures its step (train.py enqueues ~4 000 launches per iteration). */
int sr_graph_replace_memset_nodes(void* hip_graph, int* replaced);
/* Number of kernel nodes / of all nodes of a captured hipGraph_t (launch census of a replayed phase: bench.py reports
 * launches per step beside the device time). */
int sr_graph_node_count(void* graph, int* kernel_nodes, int* all_nodes);

/* One layer of the LPIPS distance (reference lpips/networks_basic.py:62-85 + lpips/__init__.py:42-44: normalize_tensor,
 * squared difference, 1x1 `lin` convolution, spatial average) fused — used by the latent-inversion loop:
 *   d[b] = mean_hw sum_c lin[c] * (f0[b,c,p] / (sqrt(sum_c f0^2) + eps) - t[b,c,p])^2
 * f0 [b, c, hw] raw features, t [b | 1, c, hw] NORMALISED target features (t_bstride = c*hw or 0), lin [c].
 * sr_lpips_layer_bwd: gf = d(sum_b gd[b] d[b]) / d f0.  scratch: sr_lpips_layer_scratch_floats(b, hw) floats. */
int64_t sr_lpips_layer_scratch_floats(int64_t b, int64_t hw);
int sr_lpips_layer_fwd(float* d, const float* f0, const float* t, const float* lin, int64_t b, int64_t c, int64_t hw,
                       int64_t t_bstride, float eps, float* scratch, sr_stream_t stream);
int sr_lpips_layer_bwd(float* gf, const float* gd, const float* f0, const float* t, const float* lin, int64_t b,
                       int64_t c, int64_t hw, int64_t t_bstride, float eps, sr_stream_t stream);
/* Pixel term of the inversion loss (BASELINE config[4]): out[0] = mean((a - b)^2) over n elements (one workgroup, fixed
 * order), and ga = gout[0] * 2 / n * (a - b).  a, b 16-byte aligned for the forward. */
int sr_mse_fwd(float* out, const float* a, const float* b, int64_t n, sr_stream_t stream);
int sr_mse_bwd(float* ga, const float* gout, const float* a, const float* b, int64_t n, sr_stream_t stream);
/* 2 x 2 / stride 2 max pooling of the LPIPS trunk (reference lpips/pretrained_networks.py:97-135, torchvision VGG16
 * features 4 / 9 / 16 / 23) over `planes` maps of ih x iw (both even), and its gradient: gx gets gy at the arg-max of
 * every window (first maximum in row-major order, a NaN wins — torch.nn.functional.max_pool2d's rule) and zero elsewhere;
 * every input pixel is written. */
int sr_maxpool2_fwd(float* out, const float* x, int64_t planes, int64_t ih, int64_t iw, sr_stream_t stream);
int sr_maxpool2_bwd(float* gx, const float* gy, const float* x, int64_t planes, int64_t ih, int64_t iw, sr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* STYLERENDERER_AMD_H_ */
