/*
 * nerf_hip.h -- C ABI of libnerf_hip.so, the MI355X (gfx950) implementation of the
 * ray-march -> encode -> MLP -> composite path of josedelrey/nerf-pytorch.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless said otherwise;
 *     inputs are borrowed, outputs are caller-allocated; the library allocates nothing
 *     and keeps no pointer after a call returns;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only
 *     enqueue work, they never synchronise;
 *   - return value 0 = enqueued, <0 = error (NERF_E_*); nerf_last_error() gives the text
 *     of the last error raised on the calling thread;
 *   - all arithmetic is fp32 (dense layers on v_mfma_f32_32x32x2_f32).
 *
 * Each entry point names the reference code it replaces (path:line in the reference
 * repository).
 */
#ifndef NERF_HIP_H
#define NERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_ABI_VERSION 10

#define NERF_OK 0
#define NERF_E_ARG (-1)      /* bad shape / size / unsupported configuration */
#define NERF_E_NULL (-2)     /* required pointer is NULL */
#define NERF_E_LAUNCH (-3)   /* HIP reported a launch or runtime error */

#define NERF_MODEL_NERF 0    /* nerf/models.py:9-75   (ReLU MLP, positional encoding) */
#define NERF_MODEL_SIREN 1   /* nerf/models.py:130-203 (sine MLP) */

int nerf_abi_version(void);
const char* nerf_last_error(void);

/* ---- sampling: nerf/rendering.py:6-27 (stratified_sampling) and :54-57 (deltas) -------
 * t[s] = lo[s] + (hi[s]-lo[s])*u[s] over the S+1 edges of linspace(near, far, S+1), bit-exact
 * with ATen's CPU fp32 linspace (two-sided, one fused rounding per edge);
 * deltas[s] = t[s+1]-t[s], deltas[S-1] = 1e10.  u, t, deltas: [S].  deltas may be NULL. */
int nerf_stratified_t(double near, double far, int S, const float* u, float* t, float* deltas, void* stream);

/* ---- sample positions: nerf/rendering.py:59-62 (+ :82 when normalize != 0) -------------
 * pos[r,s,:] = o[r,:] + t[s]*d[r,:]; with normalize: 2*(pos-near)/(far-near)-1 in the
 * reference's operation order (bit-exact with the CPU reference).  pos: [R,S,3]. */
int nerf_sample_positions(const float* rays_o, const float* rays_d, const float* t, int64_t R, int S,
                          int normalize, double near, double far, float* pos, void* stream);

/* ---- pinhole rays: nerf/data.py:65-114 (compute_rays) -------------------------------------
 * c2w [N,4,4] row-major camera-to-world matrices -> rays_o, rays_d [N, H*W, 3]: direction of
 * pixel (u,v) = normalise(c2w[:3,:3] @ (u - W/2, -(v - H/2), -focal)), origin = c2w[:3,3]. */
int nerf_compute_rays(const float* c2w, int64_t N, int H, int W, double focal, float* rays_o, float* rays_d,
                      void* stream);

/* ---- positional encoding: nerf/encoding.py:4-20 ----------------------------------------
 * out[m,:] = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], x: [M,D],
 * out: [M, D*(1+2L)], each sin/cos block D wide.  1 <= D <= 4, 0 <= L <= 16. */
int nerf_posenc_fwd(const float* x, int64_t M, int D, int L, float* out, void* stream);

/* What autograd derives for nerf/encoding.py:15-20 wrt x: grad_out [M, D*(1+2L)] -> dx [M,D] =
 * g_x + sum_j 2^j (g_sin_j cos(2^j x) - g_cos_j sin(2^j x)). */
int nerf_posenc_bwd(const float* x, const float* grad_out, int64_t M, int D, int L, float* dx, void* stream);

/* ---- compositing: nerf/rendering.py:110-153 (compute_accumulated_transmittance,
 * composite_volume) ------------------------------------------------------------------------
 * colors [R,S,3], densities [R,S], deltas [S] -> rgb [R,3]; optional outputs (NULL to skip):
 * weights [R,S] (= T_excl(1-alpha)*alpha), depth [R] (= sum w*t, needs t [S]; no reference
 * counterpart), acc [R] (= sum w).  One ray per wavefront; 1 <= S <= 1024.
 * per_ray != 0: deltas and t are [R,S] (every ray has its own sample depths, as after
 * hierarchical re-sampling) instead of one shared [S] vector.
 * colors == NULL together with rgb == NULL: weights only (weights must then be given). */
int nerf_composite_fwd(const float* colors, const float* densities, const float* deltas, const float* t,
                       int64_t R, int S, int per_ray, int white_background, float* rgb, float* weights,
                       float* depth, float* acc, void* stream);

/* compute_accumulated_transmittance on its own (nerf/rendering.py:110-122): out[r,s] =
 * prod_{j<s} betas[r,j], out[r,0] = 1.  betas, out: [R,S]. */
int nerf_exclusive_cumprod(const float* betas, int64_t R, int S, float* out, void* stream);

/* What autograd derives for compute_accumulated_transmittance (nerf/rendering.py:120-122: ATen's
 * cumprod_backward through the shift): grad_out [R,S] = dL/dT -> dbetas [R,S] = dL/dbetas, exact when
 * betas contain zeros (suffix scan, no division).  1 <= S <= 1024. */
int nerf_exclusive_cumprod_bwd(const float* betas, const float* grad_out, int64_t R, int S, float* dbetas,
                               void* stream);

/* Gradient of nerf_composite_fwd's rgb wrt colors and densities (what autograd derives for
 * nerf/rendering.py:143-151, including ATen's zero-safe cumprod backward).
 * grad_rgb [R,3] -> dcolors [R,S,3], ddensities [R,S]. */
int nerf_composite_bwd(const float* colors, const float* densities, const float* deltas, int64_t R, int S,
                       int per_ray, int white_background, const float* grad_rgb, float* dcolors,
                       float* ddensities, void* stream);

/* ---- MLP: nerf/models.py NeRF.forward :62-75, Siren.forward :185-203 ---------------------
 * Only the architecture the reference's callers build is supported: hidden 256, 10/4
 * encoding frequencies, 8 Siren layers, w0 30 / 1, sigma_mul 10, rgb_mul 1 (train.py:101-103).
 *
 * Parameters are exchanged in the order of the reference module's state_dict
 * (24 tensors, [out,in] row-major weights): `params` is a HOST array of 24 device pointers.
 */
#define NERF_NUM_PARAM_TENSORS 24

size_t nerf_mlp_param_floats(int model);                       /* 661636 (NeRF) / 564356 (Siren) */
size_t nerf_mlp_packed_floats(int model);                      /* size of the packed weight image */
size_t nerf_mlp_fwd_workspace_floats(int model, int64_t M);    /* activations kept for backward */
size_t nerf_mlp_bwd_workspace_floats(int model, int64_t M);    /* layer gradients + wgrad partials */

/* Re-order the 24 parameter tensors into the MFMA-fragment-ordered image the fused kernels
 * stream (forward image [k/8][n][8] and its transpose for the backward data gradient).
 * Must be re-run whenever a parameter changes. */
int nerf_mlp_pack(int model, const float* const* params, float* packed, void* stream);

/* Fused encode + MLP forward over M samples.
 * Inputs, one of:
 *   points != NULL: points [M,3] (already normalised), dirs [M,3]            (models.py forward)
 *   points == NULL: rays_o, rays_d [R,3], t [S] (t_per_ray == 0) or t [R,S] (t_per_ray != 0),
 *                   M = R*S; sample m belongs to ray m/S and the position is generated and
 *                   normalised in registers as nerf/rendering.py:59-62,82 would (never
 *                   materialised in HBM).
 * Outputs: colors [M,3], density [M].  colors == NULL (workspace == NULL only): the
 * pass stops after the density row and writes densities only -- the coarse pass of a hierarchical
 * render without gradients needs nothing else (no counterpart in the reference, whose renderer is
 * single pass).
 * workspace: NULL for inference; else nerf_mlp_fwd_workspace_floats(model, M) floats that
 * nerf_mlp_bwd reads. */
int nerf_mlp_fwd(int model, const float* packed, int64_t M, const float* points, const float* dirs,
                 const float* rays_o, const float* rays_d, const float* t, int S, int t_per_ray, double near,
                 double far, float* colors, float* density, float* workspace, void* stream);

/* ---- opt-in bf16 inference mode (NERF_MODEL_NERF only; no counterpart in the reference, which is
 * fp32 throughout) ------------------------------------------------------------------------------
 * Same computation as nerf_mlp_fwd without a workspace, with every weight and every layer input
 * (encodings and activations) rounded to bf16 and fp32 accumulation (v_mfma_f32_32x32x16_bf16);
 * biases, the sigmoid and the sample positions stay fp32.  Not a parity path: expect about 1e-2
 * absolute deviation of colors from the fp32 kernels.  `packed` is an image of
 * nerf_mlp_packed_bf16_bytes(model) BYTES built by nerf_mlp_pack_bf16 (re-run after every
 * parameter change); it is unrelated to the fp32 image of nerf_mlp_pack.  colors == NULL: the pass stops after the
 * layer that holds the density row and writes densities only (as nerf_mlp_fwd does). */
size_t nerf_mlp_packed_bf16_bytes(int model);                  /* 0 for models without a bf16 kernel */
int nerf_mlp_pack_bf16(int model, const float* const* params, void* packed, void* stream);
int nerf_mlp_fwd_bf16(int model, const void* packed, int64_t M, const float* points, const float* dirs,
                      const float* rays_o, const float* rays_d, const float* t, int S, int t_per_ray, double near,
                      double far, float* colors, float* density, void* stream);

/* ---- opt-in three-term inference mode "bf16x3" (both models; no counterpart in the reference) -----
 * Same computation as nerf_mlp_fwd without a workspace and same arguments as nerf_mlp_fwd_bf16, at fp32 accuracy: every weight, encoding value and
 * activation is the exact sum of three bf16 terms (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)) and every
 * product is formed from the six term products w0x0 + w0x1 + w0x2 + w1x0 + w1x1 + w2x0, each exact in the fp32
 * accumulator of v_mfma_f32_16x16x32_bf16 (16-sample tiles, csrc/mlp_x3i.hip); what is dropped is below 2^-24 of the
 * product.  The results differ from nerf_mlp_fwd's by fp32 accumulation rounding only and hold the same 1e-5 bar
 * against the reference; six bf16 MFMAs take 0.375 of the matrix-pipe time of the fp32 MFMAs they replace.  `packed` is an image of
 * nerf_mlp_packed_bf16x3_bytes(model) BYTES built by nerf_mlp_pack_bf16x3 (re-run after every parameter change). */
size_t nerf_mlp_packed_bf16x3_bytes(int model);                /* 0 for an unknown model */
int nerf_mlp_pack_bf16x3(int model, const float* const* params, void* packed, void* stream);
int nerf_mlp_fwd_bf16x3(int model, const void* packed, int64_t M, const float* points, const float* dirs,
                        const float* rays_o, const float* rays_d, const float* t, int S, int t_per_ray, double near,
                        double far, float* colors, float* density, void* stream);

/* Backward of nerf_mlp_fwd wrt the 24 parameter tensors.
 * dcolors [M,3], ddensity [M]: upstream gradients; colors: the forward's output.
 * grads: flat buffer of nerf_mlp_param_floats(model) floats, tensors laid out back to back
 * in state_dict order; accumulate == 0: overwritten, != 0: added to (as autograd does when several
 * ray chunks contribute to one optimizer step).
 * bwd_workspace: nerf_mlp_bwd_workspace_floats(model, M) floats of scratch. */
int nerf_mlp_bwd(int model, const float* packed, int64_t M, const float* dcolors, const float* ddensity,
                 const float* colors, const float* fwd_workspace, float* bwd_workspace, float* grads,
                 int accumulate, void* stream);

/* The two stages of nerf_mlp_bwd, callable separately (nerf_mlp_bwd = data then weights):
 *   _data:    fused per-tile data-gradient chain; fills bwd_workspace with every layer's dY;
 *   _weights: per-layer dW = dY^T X over all samples (MFMA, two-stage deterministic reduce)
 *             and bias gradients, written into `grads`. */
int nerf_mlp_bwd_data(int model, const float* packed, int64_t M, const float* dcolors, const float* ddensity,
                      const float* colors, const float* fwd_workspace, float* bwd_workspace, void* stream);
int nerf_mlp_bwd_weights(int model, int64_t M, const float* fwd_workspace, float* bwd_workspace, float* grads,
                         int accumulate, void* stream);

/* ---- opt-in three-term TRAINING mode "bf16x3" (both models; no counterpart in the reference, whose
 * train step, train.py:166-183, is fp32 throughout) ------------------------------------------------------
 * nerf_mlp_fwd with a workspace, and nerf_mlp_bwd_data, on the bf16 matrix pipe with the arithmetic of the bf16x3
 * inference mode above (three bf16 terms per fp32 operand, six exact term products per product, fp32
 * accumulation): same arguments, same workspaces, results within fp32 accumulation rounding of the fp32 kernels'
 * (same 1e-5 bar, gradients included).  `packed` is an image of nerf_mlp_packed_x3_floats(model) FLOATS built by
 * nerf_mlp_pack_x3: the fp32 image of nerf_mlp_pack (so the same pointer also serves nerf_mlp_fwd without a
 * workspace) followed by the three-term images the kernels of this mode stream (NeRF: the transposed planes of the
 * data-gradient chain, then the image of nerf_mlp_pack_bf16x3, which the forward -- csrc/mlp_x3i.hip, the inference
 * kernel with workspace stores -- reads; Siren: forward and transposed planes).  A workspace written by
 * nerf_mlp_fwd_x3 must be read by nerf_mlp_bwd_data_x3 (the ReLU decisions have this mode's own layout);
 * nerf_mlp_bwd_weights, or its three-term variant below, then runs on either workspace. */
size_t nerf_mlp_packed_x3_floats(int model);                   /* 0 for a model without this mode */
int nerf_mlp_pack_x3(int model, const float* const* params, float* packed, void* stream);
int nerf_mlp_fwd_x3(int model, const float* packed, int64_t M, const float* points, const float* dirs,
                    const float* rays_o, const float* rays_d, const float* t, int S, int t_per_ray, double near,
                    double far, float* colors, float* density, float* workspace, void* stream);
int nerf_mlp_bwd_data_x3(int model, const float* packed, int64_t M, const float* dcolors, const float* ddensity,
                         const float* colors, const float* fwd_workspace, float* bwd_workspace, void* stream);
/* nerf_mlp_bwd_weights with its wide products (the nine 256 x 256 layers and the 128 x 256 one: 95 % of the flops) on
 * the bf16 matrix pipe, operands split into three bf16 terms while they are staged; the narrow products, the partial
 * slabs and the ordered reduction are those of nerf_mlp_bwd_weights.  Reads either kind of workspace. */
int nerf_mlp_bwd_weights_x3(int model, int64_t M, const float* fwd_workspace, float* bwd_workspace, float* grads,
                            int accumulate, void* stream);

/* ---- layer-wise dense path for the other architectures the reference's constructors accept
 * (nerf/models.py:18-60 NeRF(pos_encoding_dim, dir_encoding_dim, hidden_dim), :148-183 Siren(...)):
 * one nn.Linear + activation per call, any M, K, N.
 *   forward:  y[M,N] = post * f(pre * (x[M,K] W[N,K]^T + b[N])),  f = NERF_ACT_*; x and y may be column
 *             slices of wider matrices (row pitch ldx / ldy); aux (required for NERF_ACT_SIN, row pitch
 *             ldy) receives post*pre*cos(pre z) for the backward.
 *   backward: dy[M,N] (pitch lddy), the forward's y / aux (pitch ldy) and x -> dx[M,K] (pitch lddx, NULL to
 *             skip), dW[N,K], db[N] (NULL to skip); workspace: nerf_linear_bwd_workspace_floats(M,K,N)
 *             floats; M <= 65535*64 per call.  Reductions are ordered (no atomics). */
#define NERF_ACT_NONE 0
#define NERF_ACT_RELU 1
#define NERF_ACT_SIN 2
#define NERF_ACT_SIGMOID 3
size_t nerf_linear_bwd_workspace_floats(int64_t M, int K, int N);
int nerf_linear_fwd(const float* x, int64_t ldx, const float* W, const float* b, int64_t M, int K, int N, int act,
                    double pre, double post, float* y, int64_t ldy, float* aux, void* stream);
int nerf_linear_bwd(const float* dy, int64_t lddy, const float* y, const float* aux, int64_t ldy, int act, double pre,
                    double post, const float* x, int64_t ldx, const float* W, int64_t M, int K, int N, float* workspace,
                    float* dx, int64_t lddx, float* dW, float* db, void* stream);

/* Allocation-free variant of nerf_linear_bwd.  NO counterpart in the reference: it exists for callers that
 * cannot allocate device memory around the call -- code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  Same arguments and results as nerf_linear_bwd minus `workspace`: the scratch
 * (activation gradients, weight-gradient partial slabs, bias partials) is a buffer of
 * NERF_NOALLOC_SCRATCH_FLOATS floats (32 MiB) built into the library, one copy per GPU, held until the process
 * ends.  Shapes with nerf_linear_bwd_workspace_floats(M, K, N) > NERF_NOALLOC_SCRATCH_FLOATS are rejected with
 * NERF_E_ARG.  The call allocates nothing and frees nothing; its first use on a device resolves that device's
 * buffer address (hipGetSymbolAddress; make that first call outside a graph capture).  All calls on one GPU share the buffer:
 * calls that overlap on the same GPU (different streams or threads) race; calls on one stream are safe.
 * The device current at the call must be the one the pointers live on. */
#define NERF_NOALLOC_SCRATCH_FLOATS (8u << 20)
int nerf_linear_bwd_noalloc(const float* dy, int64_t lddy, const float* y, const float* aux, int64_t ldy, int act,
                            double pre, double post, const float* x, int64_t ldx, const float* W, int64_t M, int K, int N,
                            float* dx, int64_t lddx, float* dW, float* db, void* stream);

/* ---- optimizer: torch.optim.Adam as train.py:123,182 uses it (no weight decay, no amsgrad) --
 * One launch over a flat parameter buffer: p, g, m, v [n].  step is 1-based. */
int nerf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step, double lr, double beta1,
                   double beta2, double eps, double grad_scale, void* stream);

/* ---- hierarchical sampling (no counterpart in the reference; canonical NeRF sample_pdf) ----
 * nerf_weights_to_cdf: weights [R,B] -> cdf [R,B+1] = [0, cumsum((w+1e-5)/sum(w+1e-5))].
 * nerf_sample_pdf: bins [R,B+1], cdf [R,B+1], u [R,N] -> samples [R,N] (linear interpolation
 * inside the bin) and idx [R,N] (int32: searchsorted(cdf, u, right=True); a pure comparison
 * result, bit-exact for identical cdf and u).
 * nerf_merge_samples: per ray, the sorted union of the shared coarse depths t_coarse [Sc] and
 * that ray's fine depths t_fine [R,Sf] -> t_all [R,Sc+Sf] (ascending) and deltas [R,Sc+Sf]
 * (successive differences, last = 1e10).  Sc+Sf <= 1024. */
int nerf_weights_to_cdf(const float* weights, int64_t R, int B, float* cdf, void* stream);
int nerf_merge_samples(const float* t_coarse, const float* t_fine, int64_t R, int Sc, int Sf, float* t_all,
                       float* deltas, void* stream);
int nerf_sample_pdf(const float* bins, const float* cdf, const float* u, int64_t R, int B, int N,
                    float* samples, int32_t* idx, void* stream);

/* The three steps above as they are chained between a coarse and a fine pass, in one launch (one ray per
 * wavefront; CDF, bin edges and fine depths stay in LDS): weights [R,Sc] = the coarse pass's compositing weights,
 * of which the interior ones weights[:,1:Sc-1] form the PDF over the Sc-1 mid-points of t_coarse [Sc]; u [R,Sf] ->
 * t_all, deltas [R,Sc+Sf]; optional outputs (NULL to skip) t_fine [R,Sf], idx [R,Sf].  Bitwise the results of
 * nerf_weights_to_cdf -> nerf_sample_pdf -> nerf_merge_samples.  Sc >= 3, Sc+Sf <= 1024. */
int nerf_resample(const float* weights, const float* t_coarse, const float* u, int64_t R, int Sc, int Sf,
                  float* t_all, float* deltas, float* t_fine, int32_t* idx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_HIP_H */
