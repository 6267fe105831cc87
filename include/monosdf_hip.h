/* C ABI of libmonosdf_hip.so -- the gfx950 (MI355X) implementation of MonoSDF's SDF
 * volume-rendering hot path.  Plain C: raw DEVICE pointers, sizes, an explicit
 * hipStream_t passed as void*; every function returns 0 on success (see MSDF_* codes).
 * Callers own every buffer; nothing here allocates, frees or synchronises.
 *
 * What each entry point replaces in the reference (paths relative to /root/reference/code):
 *
 *   msdf_hash_encode_forward / _backward / _second_backward
 *       the pybind11 module `_hash_encoder` (hashencoder/src/bindings.cpp:5-8,
 *       hashencoder/src/hashencoder.h:13-15): same argument order and meaning, tensors
 *       replaced by device pointers, plus the stream the reference never passed.
 *   msdf_pack_weights, msdf_sdf_forward, msdf_sdf_fwd_grad, msdf_sdf_backward
 *       ImplicitNetwork / ImplicitNetworkGrid .forward / .get_sdf_vals / .get_outputs /
 *       .gradient_sdf and their autograd double backward (model/network.py:79-137, 247-309).
 *   msdf_color_forward, msdf_color_backward
 *       RenderingNetwork.forward and its backward (model/network.py:389-470).
 *   msdf_wgrad, msdf_reduce
 *       the dW / db `mm` + `sum` nodes autograd emits for those networks
 *       (implied by loss.backward(), training/monosdf_train.py:431).
 *   msdf_composite_forward, msdf_composite_backward
 *       LaplaceDensity (model/density.py:21-30), MonoSDFNetwork.volume_rendering and the
 *       composites (model/network.py:552-562, 603-605, 626-640) and their backward.
 *   msdf_sampler_* (see below)
 *       ErrorBoundSampler.get_z_vals / UniformSampler.get_z_vals (model/ray_sampler.py:48-83, 110-272).
 *   msdf_weightnorm_forward, msdf_weightnorm_backward
 *       the nn.utils.weight_norm hooks of every Linear (model/network.py:72-73, 239-240, 381-382).
 *   msdf_camera_rays
 *       rend_util.get_camera_params + lift, called twice per chunk (utils/rend_util.py:63-91, 105-118;
 *       model/network.py:505-516).
 *   msdf_pixel_rays
 *       the dataset's per-pixel ray tables and the pixel branch of its __getitem__ (datasets/scene_dataset.py:269-307,
 *       374-401).
 *   msdf_monosdf_loss
 *       MonoSDFLoss.forward and its backward (model/loss.py:180-311 with 29-87, 156-171).
 *   msdf_probe_loss
 *       no reference counterpart: the fixed scalar bench.py differentiates.
 *
 * Entry points that take an msdf_plan_t run on the matrix core named by plan->precision (monosdf_plan.h).
 */
#ifndef MONOSDF_HIP_H
#define MONOSDF_HIP_H

#include <stdint.h>
#include "monosdf_plan.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MSDF_OK 0
#define MSDF_ERR_ARG 1
#define MSDF_ERR_LAUNCH 2
#define MSDF_ERR_UNSUPPORTED 3

#define MSDF_ABI_VERSION 8
int msdf_abi_version(void);

/* ---- hash grid (reference: hashencoder/src/hashencoder.h:13-15) ----
 * calc_grad_inputs: 0 / 1 as in the reference (dy_dx laid out [B, L, 3 C]); 2 = the same with dy_dx level-major
 * [L, B, 3 C], which every kernel here reads and writes as contiguous rows -- a caller that owns dy_dx end to end
 * passes 2 to all three calls. */
int msdf_hash_encode_forward(const float* inputs, const float* embeddings, const int* offsets, float* outputs,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             int calc_grad_inputs, float* dy_dx, void* stream);
int msdf_hash_encode_backward(const float* grad, const float* inputs, const float* embeddings, const int* offsets,
                              float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                              uint32_t H, int calc_grad_inputs, const float* dy_dx, float* grad_inputs,
                              void* stream);
int msdf_hash_encode_second_backward(const float* grad, const float* inputs, const float* embeddings,
                                     const int* offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                     uint32_t H, int calc_grad_inputs, const float* dy_dx,
                                     const float* grad_grad_inputs, float* grad_grad, float* grad2_embeddings,
                                     void* stream);

/* The same two gradients with the embedding scatter summed per table slice in LDS instead of one memory-side float
 * atomic per corner (csrc/hashgrid.hip "Binned scatter"): the argument lists above, followed by the number of rows of
 * the embedding table and a caller-owned DEVICE workspace of at least msdf_hash_scatter_workspace_bytes() bytes
 * (16-byte aligned; contents are scratch, nothing persists between calls).  Results equal the plain entry points up
 * to the order of the fp32 sums. */
int64_t msdf_hash_scatter_workspace_bytes(uint32_t B, uint32_t C, uint32_t L, uint64_t n_entries);
int msdf_hash_encode_backward_ws(const float* grad, const float* inputs, const float* embeddings, const int* offsets,
                                 float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                 uint32_t H, int calc_grad_inputs, const float* dy_dx, float* grad_inputs,
                                 uint64_t n_entries, void* workspace, uint64_t workspace_bytes, void* stream);
int msdf_hash_encode_second_backward_ws(const float* grad, const float* inputs, const float* embeddings,
                                        const int* offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                        uint32_t H, int calc_grad_inputs, const float* dy_dx,
                                        const float* grad_grad_inputs, float* grad_grad, float* grad2_embeddings,
                                        uint64_t n_entries, void* workspace, uint64_t workspace_bytes, void* stream);
/* Both embedding gradients of a training step in one scatter (they visit the same corners): grad_embeddings +=
 * what msdf_hash_encode_backward adds for `grad_first` plus what msdf_hash_encode_second_backward adds for
 * (`grad_second`, grad_grad_inputs).  grad_first / grad_second: [L,B,C]. */
int msdf_hash_encode_backward_fused(const float* grad_first, const float* grad_second, const float* inputs,
                                    const int* offsets, float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                    uint32_t L, float S, uint32_t H, const float* grad_grad_inputs,
                                    uint64_t n_entries, void* workspace, uint64_t workspace_bytes, void* stream);
/* The same with "=" instead of "+=": grad_embeddings is an output and need not be initialised by the caller (the
 * reference's callers zero-fill it first, hashgrid.py:75-76,93-94: here neither that fill nor a read of the table
 * is needed). */
int msdf_hash_encode_backward_fused_out(const float* grad_first, const float* grad_second, const float* inputs,
                                        const int* offsets, float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                        uint32_t L, float S, uint32_t H, const float* grad_grad_inputs,
                                        uint64_t n_entries, void* workspace, uint64_t workspace_bytes, void* stream);

/* "Node" forms for the fused grid node (no counterpart in the reference: it runs these steps as tensor expressions
 * around its three kernels).  Points in world coordinates: x01 = (x / divide_factor + 1) / 2 is formed inside, rounded
 * as the tensor expression rounds it, and returned in x01_out (may be NULL).  Features and the gradients with respect
 * to them: pitch == 0 level-major [L, B, C] (the hash kernels' own, coalesced layout); pitch > 0 point-major rows of
 * `pitch` floats ([B, pitch], level l channel c at column l*C + c, columns >= L*C written as zeros) -- what the fused
 * SDF kernels read / write.  dy_dx is level-major [L, B, 3C] (NULL: none).
 * msdf_hash_transpose: [L, B, C] <-> [B, pitch] through LDS tiles (both sides coalesced), one or two tensors per launch.
 * msdf_hash_node_input_gradient: inout[b,:] += scale * sum_{l,c} g(b,l,c) * dy_dx[l,b,:,c].
 * msdf_hash_node_second_grad: gg_out[b,:] = scale * (b < n_split ? g_a[b,:] : g_b[b - n_split,:]) (NULL part = 0),
 * grad_grad[b, l*C+c] = sum_d gg_out[b,d] * dy_dx[l,b,d,c].
 * msdf_hash_node_scatter: msdf_hash_encode_backward_fused_out with point-major grad_first / grad_second. */
int msdf_hash_node_forward(const float* x, double divide_factor, float* x01_out, const float* embeddings,
                           const int* offsets, float* feat, uint32_t pitch, uint32_t B, uint32_t C, uint32_t L,
                           float S, uint32_t H, float* dy_dx, void* stream);
int msdf_hash_node_input_gradient(const float* g, uint32_t pitch, const float* dy_dx, uint32_t B, uint32_t C,
                                  uint32_t L, float scale, float* inout, void* stream);
int msdf_hash_node_second_grad(const float* g_a, const float* g_b, uint32_t n_split, float scale, float* gg_out,
                               const float* dy_dx, float* grad_grad, uint32_t pitch, uint32_t B, uint32_t C,
                               uint32_t L, void* stream);
int msdf_hash_transpose(const float* src, float* dst, const float* src2, float* dst2, uint32_t L, uint32_t B,
                        uint32_t C, uint32_t pitch, int to_point_major, void* stream);
int msdf_hash_node_scatter(const float* grad_first, const float* grad_second, uint32_t pitch, const float* inputs,
                           const int* offsets, float* grad_embeddings, uint32_t B, uint32_t C, uint32_t L, float S,
                           uint32_t H, const float* grad_grad_inputs, uint64_t n_entries, void* workspace,
                           uint64_t workspace_bytes, void* stream);

/* ---- fused MLPs ----
 * Every entry point that takes a plan runs on the matrix core named by plan->precision (monosdf_plan.h):
 * MSDF_PRECISION_F32 = fp32 MFMA; MSDF_PRECISION_BF16X3 = each product as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi
 * on the bf16 matrix cores with fp32 accumulation (~1e-5 relative error, measured in tests/); MSDF_PRECISION_BF16X6 =
 * three bf16 planes per operand and the six cross terms down to 2^-16 of the product (fp32-grade: its parity tests
 * are held to the fp32 core's tolerances; msdf_wgrad runs the fp32 kernel for it).  The cores use
 * different weight packs: `wpack` is opaque, sized by the host (monosdf_amd/plan.py) and produced by
 * msdf_pack_weights for the same plan. */
int msdf_pack_weights(const msdf_plan_t* plan, const msdf_packrule_t* rules_dev, const int* maps_dev,
                      const float* flat_w, const float* flat_b, void* wpack, float* bpack, void* stream);

/* weight normalisation of every layer of a network in one launch (reference: nn.utils.weight_norm hooks,
 * model/network.py:72-73, 239-240, 381-382).  row_layer_dev[r] = layer index of global row r. */
typedef struct {
  const float* v;      /* [rows, cols] weight_v (or the plain weight when has_g == 0) */
  const float* g;      /* [rows] weight_g */
  const float* b;      /* [rows] bias */
  int32_t rows, cols;
  int32_t w_off;       /* float offset of this layer in the flat weight buffer */
  int32_t b_off;       /* float offset of this layer in the flat bias buffer */
  int32_t row_off;     /* first global row of this layer */
  int32_t has_g;
} msdf_wn_layer_t;
int msdf_weightnorm_forward(const msdf_wn_layer_t* layers_dev, const int* row_layer_dev, int total_rows,
                            float* flat_w, float* flat_b, float* norms, void* stream);
int msdf_weightnorm_backward(const msdf_wn_layer_t* layers_dev, const int* row_layer_dev, int total_rows,
                             const float* g_flat_w, const float* norms, float* dv_flat, float* dg_rows, void* stream);

int msdf_sdf_forward(const msdf_plan_t* plan, const void* wpack, const float* bpack, const float* x,
                     const float* aux, int P, float clamp_radius, float sphere_scale, float* sdf, void* stream);
/* the same, skipped on the device when *run_flag == 0 (run_flag may be NULL): the SDF evaluation of a sampler round
 * that the previous round did not ask for (msdf_sampler_args_t.flags) costs a kernel start, not a network pass */
int msdf_sdf_forward_if(const msdf_plan_t* plan, const void* wpack, const float* bpack, const float* x,
                        const float* aux, int P, float clamp_radius, float sphere_scale, float* sdf,
                        const uint32_t* run_flag, void* stream);
/* the same with the extra input features given as the hash encoder's level-major tensor [aux_LC / aux_C][P][aux_C]
 * (msdf_fg_args_t.aux_C); aux_C == 0 is msdf_sdf_forward_if */
int msdf_sdf_forward_lm(const msdf_plan_t* plan, const void* wpack, const float* bpack, const float* x,
                        const float* aux, int aux_C, int aux_LC, int P, float clamp_radius, float sphere_scale,
                        float* sdf, const uint32_t* run_flag, void* stream);

typedef struct {
  const void* wpack;
  const float* bpack;
  const float* x;          /* [P,3] */
  const float* aux;        /* [P, 16*aux_tiles] or NULL */
  int32_t P, P_pad;        /* P_pad = P rounded up to 64 (workspace rows) */
  int32_t n_clamp;         /* points [0,n_clamp) get the bounding-sphere clamp */
  int32_t n_feat;          /* points [0,n_feat) need the feature vector (multiple of 16 or == P) */
  float clamp_radius, sphere_scale;
  float* sdf;              /* [P] */
  float* feat;             /* [n_feat, 16*feat_tiles] */
  float* nrm;              /* [P,3] */
  float* r_aux;            /* [P, 16*aux_tiles] or NULL */
  unsigned char* clamped;  /* [P] */
  float* H;                /* [hsum * P_pad] */
  float* PM;               /* [hsum * P_pad] */
  float* IN0;              /* [P_pad, 16*(e_tiles+aux_tiles)] */
  int32_t save;
  int32_t aux_C;           /* layout of aux / r_aux: 0 = rows [P, 16*aux_tiles]; C = 1, 2, 4, 8: the hash encoder's own
                              level-major tensors [aux_LC / C][P][C] (column l C + c of a point = level l, channel c) */
  int32_t aux_LC;          /* valid columns (levels x channels) of the level-major form */
  float aux_dx_scale;      /* with dy_dx: the chain-rule factor of x -> x01 (0.5 / divide_factor) */
  const float* dy_dx;      /* NULL, or (aux_C == 2 only) the hash encoder's Jacobian [L][P][3][2] as
                              msdf_hash_node_forward writes it: nrm then also holds the grid part of d sdf / d x,
                              nrm += aux_dx_scale * sum_{l,c} r_aux[l,c] dy_dx[l,.,c] (what msdf_hash_node_input_gradient
                              adds as a launch of its own; reference kernel_input_backward, hashencoder.cu:346-372) */
} msdf_fg_args_t;
int msdf_sdf_fwd_grad(const msdf_plan_t* plan, const msdf_fg_args_t* args, void* stream);

typedef struct {
  const void* wpack;
  const float* bpack;
  const float* x;
  int32_t P, P_pad;
  int32_t n_feat;
  int32_t n_split;         /* points [0, n_split) take g_sdf / g_nrm, points [n_split, P) take g_sdf_b / g_nrm_b:
                              the two groups (ray samples | eikonal points) get their gradients from different
                              consumers, and this spares the caller the zero-fill + copy + add of joining them */
  const float* g_sdf;      /* [n_split] or NULL */
  const float* g_feat;     /* [n_feat, 16*feat_tiles] or NULL */
  const float* g_nrm;      /* [n_split,3] or NULL */
  const float* g_raux;     /* [P, 16*aux_tiles] or NULL */
  const unsigned char* clamped;
  const float* H;
  const float* PM;
  float* QB;               /* [qsum * P_pad] */
  float* T;                /* unused (may be NULL): the second-order term 100 (1 - s) p p-bar is formed again in the
                              sweep down from PM and the stored q-bar rows instead of being written and re-read */
  float* AB;               /* [absum * P_pad] */
  float* GSDF;             /* [P_pad] */
  float* QLAST;            /* [P_pad, 16*kt_last] */
  float* g_aux;            /* [P, 16*aux_tiles] or NULL */
  const float* g_sdf_b;    /* [P - n_split] or NULL */
  const float* g_nrm_b;    /* [P - n_split, 3] or NULL */
  int32_t aux_C, aux_LC;   /* layout of g_raux / g_aux, as in msdf_fg_args_t */
  const float* dy_dx;      /* NULL, or (aux_C == 2 only) the encoder's Jacobian [L][P][3][2]: the gradient arriving at
                              d sdf / d features from the grid part of d sdf / d x is then formed in the kernel from
                              g_nrm, g_raux[l,c] = sum_d (aux_dx_scale g_nrm[d]) dy_dx[l,d,c] (g_raux itself is ignored) --
                              what msdf_hash_node_second_grad computes as a launch of its own (reference
                              kernel_grid_second_backward_grad, hashencoder.cu:375-428), bit-identical to it */
  float* gg_out;           /* with dy_dx: [P,3] <- aux_dx_scale * g_nrm (the scatter's grad_grad_inputs), or NULL */
  float aux_dx_scale;
  int32_t pad_;
} msdf_bw_args_t;
int msdf_sdf_backward(const msdf_plan_t* plan, const msdf_bw_args_t* args, void* stream);

typedef struct {
  const void* wpack;
  const float* bpack;
  const float* x;          /* [P,3] */
  const float* dirs;       /* [P/spr,3] */
  const float* nrm;        /* [P,3] */
  const float* feat;       /* [P, 16*feat_tiles] */
  const float* code;       /* [P/spr, 16*aux_tiles] or NULL */
  int32_t P, P_pad;
  int32_t spr;             /* samples per ray */
  int32_t save;
  float* rgb;              /* [P,3] */
  float* H;                /* [n_hidden][P_pad][16*hid_tiles] */
  float* MISC;             /* [P_pad, 16*misc_tiles] */
} msdf_color_fwd_args_t;
int msdf_color_forward(const msdf_plan_t* plan, const msdf_color_fwd_args_t* args, void* stream);

typedef struct {
  const void* wpack;
  const float* bpack;
  const float* rgb;
  const float* g_rgb;
  int32_t P, P_pad;
  const float* H;
  float* AB;               /* [absum * P_pad] */
  float* g_feat;           /* [P, 16*feat_tiles] */
  float* g_misc;           /* [P, 16*misc_tiles] */
  float* g_nrm;            /* [P,3] or NULL: the normal columns of g_misc again, dense (mode idr) */
} msdf_color_bwd_args_t;
int msdf_color_backward(const msdf_plan_t* plan, const msdf_color_bwd_args_t* args, void* stream);

/* wg_map_dev: int32 pairs (item, split) for each of the n_wgs workgroups (balanced by the host) */
int msdf_wgrad(const msdf_wgrad_item_t* items_dev, const int32_t* wg_map_dev, int n_wgs, float* partials,
               int P_pad, int precision /* MSDF_PRECISION_* */, const float* base0, const float* base1 /* the items'
               operand buffers; base1 may be NULL when no item names it */, void* stream);
int msdf_reduce(const msdf_reduce_rule_t* rules_dev, int n_rules, const int* maps_dev, const float* partials,
                float* dst, void* stream);

/* ---- compositor ---- */
typedef struct {
  const float* z;            /* [N,S] */
  const float* sdf;          /* [N,S] */
  const float* rgb;          /* [N,S,3] */
  const float* nrm;          /* [N,S,3] */
  const float* beta;         /* [1] = |beta| + beta_min */
  const float* depth_scale;  /* [N] with a pitch of depth_scale_stride floats (the z column of a [N,3] direction table) */
  int32_t N, S;
  int32_t white_bkgd;
  float bg0, bg1, bg2;
  float* weights;            /* [N,S] */
  float* rgb_values;         /* [N,3] */
  float* depth_values;       /* [N] */
  float* normal_map;         /* [N,3]: R^T sum_i w_i n_i/(|n_i|+1e-6) (camera frame) when pose != NULL, else world frame */
  float* wsum;               /* [N] */
  const float* pose;         /* [N or 1, 4, 4] camera-to-world matrices or NULL (reference network.py:608-616) */
  int32_t pose_stride;       /* 16 per-ray poses, 0 one pose for all rays */
  int32_t depth_scale_stride; /* floats between the depth scales of consecutive rays (0 is read as 1) */
  float* depth_vals;         /* [N,S] = z * depth_scale (reference network.py:572), or NULL */
} msdf_composite_args_t;
int msdf_composite_forward(const msdf_composite_args_t* args, void* stream);

typedef struct {
  const float* z;
  const float* sdf;
  const float* rgb;
  const float* nrm;
  const float* beta;
  const float* depth_scale;
  const float* weights;
  const float* wsum;
  const float* depth_values;
  const float* g_rgb_values;  /* [N,3] or NULL */
  const float* g_depth;       /* [N] or NULL */
  const float* g_normal;      /* [N,3] or NULL */
  const float* g_weights;     /* [N,S] or NULL */
  int32_t N, S;
  int32_t white_bkgd;
  float bg0, bg1, bg2;
  float* g_sdf;               /* [N,S] */
  float* g_rgb;               /* [N,S,3] */
  float* g_nrm;               /* [N,S,3] */
  float* g_beta_part;         /* [N] */
  const float* pose;          /* as in the forward */
  int32_t pose_stride;
  int32_t depth_scale_stride; /* as in the forward */
} msdf_composite_bwd_args_t;
int msdf_composite_backward(const msdf_composite_bwd_args_t* args, void* stream);

/* LaplaceDensity.get_beta (reference model/density.py:28-30) and its adjoint, one launch each:
 * out[0] = |beta_raw[0]| + beta_min;   g_raw[0] = sign(beta_raw[0]) * sum_i g_part[i]  (fixed summation order). */
int msdf_beta_eff(const float* beta_raw, float beta_min, float* out, void* stream);
int msdf_beta_grad(const float* beta_raw, const float* g_part, int n, float* g_raw, void* stream);

/* ---- fused benchmark loss: the value and all gradients in one launch of one workgroup ---- */
typedef struct {
  const float* rgb;      /* [N,3] rgb_values */
  const float* nrm;      /* [N,3] normal_map */
  const float* depth;    /* [N] depth_values */
  const float* g1;       /* [M,3] grad_theta */
  const float* g2;       /* [M,3] grad_theta_nei */
  int32_t N, M;
  float w_normal, w_depth, w_eik, w_smooth;
  float* g_rgb; float* g_nrm; float* g_depth; float* g_g1; float* g_g2;   /* d loss / d input */
  float* partial;        /* [1] the loss value */
} msdf_probe_loss_args_t;
int msdf_probe_loss(const msdf_probe_loss_args_t* args, void* stream);

/* ---- MonoSDFLoss (reference: model/loss.py:180-311 with loss.py:29-87,156-171), pixel-batch mode,
 * L1 colour loss.  One launch returns the seven scalars of the reference's output dict and the gradients of
 * `loss` with respect to every model output.  The decay factor (loss.py:291-296) is folded into w_depth /
 * w_nl1 / w_ncos by the caller. */
typedef struct {
  const float* rgb;          /* [N,3] rgb_values */
  const float* depth;        /* [N]   depth_values */
  const float* normal;       /* [N,3] normal_map */
  const float* sdf;          /* [N,S] sdf of the ray samples (foreground mask: sign change along the ray) */
  const float* grad_theta;   /* [E,3] or NULL (no eikonal / smoothness terms) */
  const float* grad_nei;     /* [E,3] or NULL */
  const float* rgb_gt;       /* [N,3] */
  const float* depth_gt;     /* [N]   monocular depth cue */
  const float* normal_gt;    /* [N,3] monocular normal cue */
  const float* mask_gt;      /* [N]   > 0.5 = valid pixel */
  int32_t N, S, E;
  int32_t gamma;             /* if_gamma_loss */
  int32_t scale_invariant;   /* if_scale_invariant_depth */
  float w_eik, w_smooth, w_depth, w_nl1, w_ncos;
  float* mask;               /* [N] scratch / output: the combined mask as 0 / 1 */
  float* out;                /* [8]: loss, rgb_loss, eikonal_loss, smooth_loss, depth_loss, normal_l1, normal_cos, sum(mask) */
  float* g_rgb; float* g_depth; float* g_normal; float* g_theta; float* g_nei;   /* d loss / d input */
} msdf_monosdf_loss_args_t;
int msdf_monosdf_loss(const msdf_monosdf_loss_args_t* args, void* stream);

/* ---- error-bounded sampler (reference: model/ray_sampler.py:48-83, 110-272) ---- */
typedef struct {
  const float* ray_o;        /* [N,3] */
  const float* ray_d;        /* [N,3] */
  int32_t N;
  int32_t M;                 /* current number of sorted samples per ray */
  int32_t m_max;             /* row pitch of z / sdf (>= n_eval * max_rounds) */
  int32_t n_eval, n_final, n_extra;
  int32_t round_idx, max_rounds;
  int32_t training;          /* 1: final u from u_final, 0: linspace */
  int32_t beta_iters;
  float near, far, bound;    /* sampler near / far clip, cube half-size (scene_bounding_sphere) */
  float eps, add_tiny;
  float lemma;               /* 1 / (4 log(1 + eps)) */
  const float* beta0;        /* [1] device: |beta| + beta_min */
  float* z;                  /* [N, m_max] sorted sample distances */
  float* sdf;                /* [N, m_max] sdf at those samples */
  float* new_z;              /* [N, n_eval] samples added by the last round */
  const float* new_sdf;      /* [N, n_eval] their sdf (from msdf_sdf_forward) */
  int32_t* new_pos;          /* [N, n_eval] their positions in z */
  float* pts;                /* [N * n_eval, 3] their 3-D points */
  float* beta;               /* [N] */
  uint32_t* flags;           /* [2 * max_rounds], zeroed by msdf_sampler_init: flags[2r] = bits of the
                                batch's max beta after round r, flags[2r+1] = 1 when round r asks for another one.
                                Rounds r > 0 return at once unless flags[2(r-1)+1] is set, so enqueueing more rounds
                                than needed is harmless (reference: ray_sampler.py:125,179). */
  const float* jitter;       /* [N, n_eval] or NULL (eval) */
  const float* u_final;      /* [N, n_final] or NULL */
  float* final_z;            /* [N, n_final] */
  const int64_t* extra_idx;  /* [max_rounds, n_extra]: row k-1 = the columns of the dense set to add to the final set
                                when k rounds ran (the kernel reads k from `flags`; reference: ray_sampler.py:242-247) */
  const int64_t* eik_idx;    /* [N] or NULL */
  float* z_out;              /* [N, n_final + n_extra + 2] */
  float* z_eik;              /* [N] or NULL */
  float* pts_out;            /* [N * S (+ 4 N), 3] or NULL: ray samples, then (training) the eikonal points */
  const float* eik_uniform;  /* [N,3] uniform points in the bounding cube, or NULL (no eikonal block) */
  const float* nei_rand;     /* [2N,3] U[0,1) jitter of the neighbour points (reference network.py:583-594) */
  float* far_out;            /* [N] or NULL: msdf_sampler_init also returns the far bound of the uniform samples */
  int32_t* rounds_out;       /* [1] or NULL: msdf_sampler_finish writes the number of rounds that ran */
  float* dbg_dstar;          /* [N, m_max] or NULL: msdf_sampler_beta writes d* of the M-1 intervals (tests) */
  float* dbg_err0;           /* [N] or NULL: msdf_sampler_beta writes the error bound at beta0 (tests) */
  float* dbg_cdf;            /* [N, m_max] or NULL: msdf_sampler_resample writes the cdf it inverts (tests) */
  const float* eik_u;        /* [N] U[0,1) or NULL: when eik_idx is NULL the eikonal sample of a ray is column
                                floor(u * S) of its final set (reference: torch.randint, ray_sampler.py:254) */
  int32_t eik_unit;          /* 1: eik_uniform holds U[0,1) values, mapped to (2u - 1) * bound here (reference:
                                uniform_(-R, R), network.py:587); 0: it holds the points themselves */
  int32_t pad_;
} msdf_sampler_args_t;
int msdf_sampler_init(const msdf_sampler_args_t* args, void* stream);
int msdf_sampler_beta(const msdf_sampler_args_t* args, void* stream);
int msdf_sampler_resample(const msdf_sampler_args_t* args, void* stream);
int msdf_sampler_finish(const msdf_sampler_args_t* args, void* stream);

/* ErrorBoundSampler.get_error_bound (reference: model/ray_sampler.py:264-272): z [N,M] sorted, sdf [N,M],
 * dstar [N,M-1], beta [N] (beta_stride 1) or [1] (beta_stride 0) -> out [N] = max opacity error bound per ray. */
int msdf_sampler_error_bound(const float* z, const float* sdf, const float* dstar, const float* beta,
                             int beta_stride, int N, int M, float* out, void* stream);

/* LaplaceDensity.density_func (reference: model/density.py:21-26) on n values laid out in rows of `cols`;
 * beta [1] (beta_stride 0) or one per row (beta_stride 1).  The backward returns d/d sdf and the per-element
 * d/d beta terms (the caller sums them per beta). */
int msdf_laplace_density(const float* sdf, const float* beta, int beta_stride, int64_t n, int cols, float* out,
                         void* stream);
int msdf_laplace_density_backward(const float* sdf, const float* beta, int beta_stride, int64_t n, int cols,
                                  const float* g, float* g_sdf, float* g_beta_elem, void* stream);

/* ---- ray generation (reference: utils/rend_util.py:63-91,105-118 as called at
 * model/network.py:505-516).  uv [n,2] pixel coordinates, pose [4,4] camera-to-world, intrinsics [4,4];
 * outputs: ray_dirs [n,3] world-space unit directions, ray_dirs_cam [n,3] the same ray in the camera frame
 * (the reference's `ray_dirs_tmp`; its z is the depth scale), cam_loc [n,3] the camera centre per ray. */
int msdf_camera_rays(const float* uv, const float* pose, const float* intrinsics, int n, float* ray_dirs,
                     float* ray_dirs_cam, float* cam_loc, void* stream);

/* One pixel-mode training batch assembled on the device (second half; reference: the per-pixel tables of
 * SceneDatasetDN.convert_to_pixels, datasets/scene_dataset.py:269-307, and the pixel branch of __getitem__ 374-401 with
 * the DataLoader's collate + host-to-device copy).  ray_idx [n] int64 in [0, n_frames * hw): ray r is pixel
 * (ray_idx[r] mod hw) of the frame at position ray_idx[r] / hw of frame_list [n_frames] (int32 indices into pose_all /
 * intrinsics_all [N,4,4]; NULL = identity).  The pixel's (u, v) is (column, row) of a row-major image of `width` columns
 * (scene_dataset.py:258-260).  Outputs: ray_dirs, ray_dirs_cam (the reference's ray_dirs_tmp), cam_loc [n,3],
 * ray_pose [n,16], frame_pos [n] int32 (= ray_idx / hw, the reference's ray_frame_idx).  Ground truth: up to 4 image
 * stacks gt_src[k] [n_frames * hw, gt_channels[k]] (rows in frame-list order, device pointers; the arrays of pointers
 * themselves are HOST arrays) gathered into gt_dst[k] [n, gt_channels[k]]; a NULL gt_src[k] is skipped.  An index outside
 * [0, n_frames * hw) touches no memory: its outputs are zeros and its frame_pos is -1. */
int msdf_pixel_rays(const int64_t* ray_idx, int n, const int32_t* frame_list, int n_frames, const float* pose_all,
                    const float* intrinsics_all, int width, int hw, float* ray_dirs, float* ray_dirs_cam,
                    float* cam_loc, float* ray_pose, int32_t* frame_pos, const float* const* gt_src,
                    float* const* gt_dst, const int32_t* gt_channels, int n_gt, void* stream);

#ifdef __cplusplus
}
#endif
#endif
