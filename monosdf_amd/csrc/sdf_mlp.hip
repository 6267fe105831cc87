// SDF-network kernels for gfx950: weight packer, no-grad forward (sampler),
// forward + d sdf/dx (get_outputs / gradient_sdf), and the double-backward sweep.
//
// Replaces, for the hot path, the PyTorch graph built by the reference's
// ImplicitNetwork / ImplicitNetworkGrid (code/model/network.py:79-137, 247-309)
// and its autograd double backward (create_graph=True at network.py:125,285,301).
// Math: DESIGN.md "SDF network kernels".
#include "sdf_kernels.h"

// ---------------------------------------------------------------------------
// weight packer: flat effective weights -> fragment-ordered packs (both orientations)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) msdf_pack_kernel(const msdf_plan_t plan,
                                                        const msdf_packrule_t* __restrict__ rules,
                                                        const int* __restrict__ maps,
                                                        const float* __restrict__ flat_w,
                                                        const float* __restrict__ flat_b,
                                                        v4f* __restrict__ wpack, float* __restrict__ bpack) {
  const int l = blockIdx.y;
  const int which = blockIdx.z;  // 0 forward pack, 1 transposed pack, 2 bias (+ sdf row on the last layer)
  const msdf_layer_t L = plan.layer[l];
  const msdf_packrule_t R = rules[l];
  const int* rowmap = maps + R.rowmap_off;
  const int* colmap = maps + R.colmap_off;
  const float* W = flat_w + R.w_off;
  const int stride = gridDim.x * blockDim.x;
  const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
  if (which == 0) {
    const int ot_even = (L.ot + 1) & ~1;
    const int total = ot_even * L.ktp * 64;
    for (int i = t0; i < total; i += stride) {
      const int lane = i & 63;
      const int blk = i >> 6;
      const int kt = blk % L.ktp, ot = blk / L.ktp;
      const int rs = 16 * ot + (lane & 15);
      const int row = (ot < L.ot) ? rowmap[rs] : -1;
      v4f v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cs = 16 * kt + 4 * (lane >> 4) + r;
        const int col = (kt < L.kt) ? colmap[cs] : -1;
        v[r] = (row >= 0 && col >= 0) ? R.scale * W[(size_t)row * R.cols + col] : 0.f;
      }
      wpack[L.wf_off + i] = v;
    }
  } else if (which == 1) {
    const int kt_even = (L.kt + 1) & ~1;
    const int total = kt_even * L.otp * 64;
    for (int i = t0; i < total; i += stride) {
      const int lane = i & 63;
      const int blk = i >> 6;
      const int ko = blk % L.otp, it = blk / L.otp;
      const int cs = 16 * it + (lane & 15);
      const int col = (it < L.kt) ? colmap[cs] : -1;
      v4f v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rs = 16 * ko + 4 * (lane >> 4) + r;
        const int row = (ko < L.ot) ? rowmap[rs] : -1;
        v[r] = (row >= 0 && col >= 0) ? R.scale * W[(size_t)row * R.cols + col] : 0.f;
      }
      wpack[L.wb_off + i] = v;
    }
  } else {
    for (int i = t0; i < 16 * L.ot; i += stride) {
      const int row = rowmap[i];
      bpack[L.bias_off + i] = (row >= 0) ? flat_b[R.b_off + row] : 0.f;
    }
    if (l == plan.n_layers - 1 && plan.wsdf_off >= 0) {
      for (int rr = 0; rr < plan.out_rows; ++rr) {
        const int row = rowmap[plan.sdf_slot + rr];
        for (int i = t0; i < 16 * L.kt; i += stride) {
          const int col = colmap[i];
          bpack[plan.wsdf_off + rr * 16 * L.kt + i] = (col >= 0 && row >= 0) ? R.scale * W[(size_t)row * R.cols + col] : 0.f;
        }
      }
    }
  }
}

__global__ void __launch_bounds__(MLP_THREADS, MLP_WGS_PER_CU)
msdf_sdf_forward_k(const msdf_plan_t plan, const v4f* __restrict__ wpack, const float* __restrict__ bpack,
                   const float* __restrict__ x, const float* __restrict__ aux, const AuxView av, const int P,
                   const float clamp_radius, const float sphere_scale, float* __restrict__ sdf_out,
                   const uint32_t* __restrict__ run_flag) {
  extern __shared__ v4f lds[];
  if (run_flag != nullptr && *run_flag == 0u) return;     // a sampler round nobody asked for (uniform over the grid)
  sdf_forward_body<CoreF32>(plan, wpack, bpack, x, aux, av, P, clamp_radius, sphere_scale, sdf_out, lds);
}

__global__ void __launch_bounds__(MLP_THREADS, MLP_WGS_PER_CU)
msdf_sdf_fwd_grad_k(const msdf_plan_t plan, const FgArgs a) {
  extern __shared__ v4f lds[];
  sdf_fwd_grad_body<CoreF32>(plan, a, lds);
}

__global__ void __launch_bounds__(MLP_THREADS, MLP_WGS_PER_CU)
msdf_sdf_backward_k(const msdf_plan_t plan, const BwArgs a) {
  extern __shared__ v4f lds[];
  sdf_backward_body<CoreF32>(plan, a, lds);
}

// bf16x3 launchers (sdf_mlp_b16.hip)
int msdf_b16_pack_weights(const msdf_plan_t*, const msdf_packrule_t*, const int*, const float*, const float*, void*,
                          float*, hipStream_t);
int msdf_b16_sdf_forward(const msdf_plan_t*, const void*, const float*, const float*, const float*, int, int, int, float,
                         float, float*, const uint32_t*, hipStream_t);
int msdf_b16_sdf_fwd_grad(const msdf_plan_t*, const msdf_fg_args_t*, hipStream_t);
int msdf_b16_sdf_backward(const msdf_plan_t*, const msdf_bw_args_t*, hipStream_t);

extern "C" int msdf_abi_version(void) { return MSDF_ABI_VERSION; }

extern "C" int msdf_pack_weights(const msdf_plan_t* plan, const msdf_packrule_t* rules_dev, const int* maps_dev,
                                 const float* flat_w, const float* flat_b, void* wpack, float* bpack,
                                 void* stream) {
  if (plan == nullptr || plan->n_layers < 1 || plan->n_layers > MSDF_MAX_LAYERS) return MSDF_ERR_ARG;
  if (plan->precision == MSDF_PRECISION_BF16X3 || plan->precision == MSDF_PRECISION_BF16X6)
    return msdf_b16_pack_weights(plan, rules_dev, maps_dev, flat_w, flat_b, wpack, bpack, (hipStream_t)stream);
  if (plan->precision != MSDF_PRECISION_F32) return MSDF_ERR_ARG;
  const dim3 grid(32, plan->n_layers, 3);
  msdf_pack_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(*plan, rules_dev, maps_dev, flat_w, flat_b, (v4f*)wpack,
                                                          bpack);
  return msdf_check_launch();
}

static bool aux_layout_ok(const msdf_plan_t* plan, const int aux_C, const int aux_LC) {
  if (aux_C == 0) return true;
  return aux_C == 2 && aux_LC > 0 && (aux_LC % aux_C) == 0 &&
         aux_LC <= 16 * plan->aux_tiles;
}

extern "C" int msdf_sdf_forward_lm(const msdf_plan_t* plan, const void* wpack, const float* bpack, const float* x,
                                   const float* aux, int aux_C, int aux_LC, int P, float clamp_radius,
                                   float sphere_scale, float* sdf, const uint32_t* run_flag, void* stream) {
  if (plan == nullptr || P < 0) return MSDF_ERR_ARG;
  if (P == 0) return MSDF_OK;
  if (plan->aux_tiles > 0 && aux == nullptr) return MSDF_ERR_ARG;
  if (!aux_layout_ok(plan, aux_C, aux_LC)) return MSDF_ERR_ARG;
  if (plan->precision == MSDF_PRECISION_BF16X3 || plan->precision == MSDF_PRECISION_BF16X6)
    return msdf_b16_sdf_forward(plan, wpack, bpack, x, aux, aux_C, aux_LC, P, clamp_radius, sphere_scale, sdf, run_flag,
                                (hipStream_t)stream);
  if (mlp_prepare((const void*)msdf_sdf_forward_k)) return MSDF_ERR_LAUNCH;
  const int grid = (P + MLP_PTS_PER_WG - 1) / MLP_PTS_PER_WG;
  const AuxView av = {aux_C, aux_LC, P};
  msdf_sdf_forward_k<<<grid, MLP_THREADS, MLP_LDS_BYTES, (hipStream_t)stream>>>(
      *plan, (const v4f*)wpack, bpack, x, aux, av, P, clamp_radius, sphere_scale, sdf, run_flag);
  return msdf_check_launch();
}

extern "C" int msdf_sdf_forward_if(const msdf_plan_t* plan, const void* wpack, const float* bpack, const float* x,
                                   const float* aux, int P, float clamp_radius, float sphere_scale, float* sdf,
                                   const uint32_t* run_flag, void* stream) {
  return msdf_sdf_forward_lm(plan, wpack, bpack, x, aux, 0, 0, P, clamp_radius, sphere_scale, sdf, run_flag, stream);
}

extern "C" int msdf_sdf_forward(const msdf_plan_t* plan, const void* wpack, const float* bpack, const float* x,
                                const float* aux, int P, float clamp_radius, float sphere_scale, float* sdf,
                                void* stream) {
  return msdf_sdf_forward_if(plan, wpack, bpack, x, aux, P, clamp_radius, sphere_scale, sdf, nullptr, stream);
}

extern "C" int msdf_sdf_fwd_grad(const msdf_plan_t* plan, const msdf_fg_args_t* a, void* stream) {
  if (plan == nullptr || a == nullptr || a->P < 0) return MSDF_ERR_ARG;
  if (a->P == 0) return MSDF_OK;
  if (a->P_pad < a->P || (a->P_pad % MLP_PTS_PER_WG) != 0) return MSDF_ERR_ARG;
  if (plan->aux_tiles > 0 && a->aux == nullptr) return MSDF_ERR_ARG;
  if (!aux_layout_ok(plan, a->aux_C, a->aux_LC)) return MSDF_ERR_ARG;
  if (a->dy_dx != nullptr && (a->aux_C != 2 || a->r_aux == nullptr)) return MSDF_ERR_ARG;
  if (plan->precision == MSDF_PRECISION_BF16X3 || plan->precision == MSDF_PRECISION_BF16X6) return msdf_b16_sdf_fwd_grad(plan, a, (hipStream_t)stream);
  if (mlp_prepare((const void*)msdf_sdf_fwd_grad_k)) return MSDF_ERR_LAUNCH;
  msdf_sdf_fwd_grad_k<<<a->P_pad / MLP_PTS_PER_WG, MLP_THREADS, MLP_LDS_BYTES, (hipStream_t)stream>>>(*plan, *a);
  return msdf_check_launch();
}

extern "C" int msdf_sdf_backward(const msdf_plan_t* plan, const msdf_bw_args_t* a, void* stream) {
  if (plan == nullptr || a == nullptr || a->P < 0) return MSDF_ERR_ARG;
  if (a->P == 0) return MSDF_OK;
  if (a->P_pad < a->P || (a->P_pad % MLP_PTS_PER_WG) != 0) return MSDF_ERR_ARG;
  if (!aux_layout_ok(plan, a->aux_C, a->aux_LC)) return MSDF_ERR_ARG;
  if (a->dy_dx != nullptr && a->aux_C != 2) return MSDF_ERR_ARG;
  if (plan->precision == MSDF_PRECISION_BF16X3 || plan->precision == MSDF_PRECISION_BF16X6) return msdf_b16_sdf_backward(plan, a, (hipStream_t)stream);
  if (mlp_prepare((const void*)msdf_sdf_backward_k)) return MSDF_ERR_LAUNCH;
  msdf_sdf_backward_k<<<a->P_pad / MLP_PTS_PER_WG, MLP_THREADS, MLP_LDS_BYTES, (hipStream_t)stream>>>(*plan, *a);
  return msdf_check_launch();
}
