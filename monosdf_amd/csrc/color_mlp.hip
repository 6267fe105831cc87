// Colour-network kernels for gfx950 (RenderingNetwork, reference: code/model/network.py:389-470,
// mode 'idr' / 'nerf', ReLU hidden layers, sigmoid or ReLU output, no `spec` branch).
//
// Same register-resident scheme as the SDF network (mlp_core.h).  The first layer's
// 289-wide input is split into two GEMM passes over the same accumulators: the
// feature tiles (256 slots, read from the SDF kernel's output) and a small "misc"
// block [x | PE(view) | normal | (per-image code)] built in registers.
//
// The plan's "layers" are pack units: unit 0 = first layer, feature columns;
// unit 1 = first layer, misc columns; units 2.. = the remaining layers.
#include "color_kernels.h"

__global__ void __launch_bounds__(MLP_THREADS, MLP_WGS_PER_CU)
msdf_color_forward_k(const msdf_plan_t plan, const ColorFwdArgs a) {
  extern __shared__ v4f lds[];
  color_forward_body<CoreF32>(plan, a, lds);
}

__global__ void __launch_bounds__(MLP_THREADS, MLP_WGS_PER_CU)
msdf_color_backward_k(const msdf_plan_t plan, const ColorBwdArgs a) {
  extern __shared__ v4f lds[];
  color_backward_body<CoreF32>(plan, a, lds);
}

// bf16x3 launchers (color_mlp_b16.hip)
int msdf_b16_color_forward(const msdf_plan_t*, const msdf_color_fwd_args_t*, hipStream_t);
int msdf_b16_color_backward(const msdf_plan_t*, const msdf_color_bwd_args_t*, hipStream_t);

static int color_prepare(const void* fn) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_BYTES) == hipSuccess
             ? MSDF_OK : MSDF_ERR_LAUNCH;
}

extern "C" int msdf_color_forward(const msdf_plan_t* plan, const msdf_color_fwd_args_t* a, void* stream) {
  if (plan == nullptr || a == nullptr || a->P < 0 || a->spr < 1) return MSDF_ERR_ARG;
  if (a->P == 0) return MSDF_OK;
  if (a->P_pad < a->P || (a->P_pad % MLP_PTS_PER_WG) != 0) return MSDF_ERR_ARG;
  if (plan->precision == MSDF_PRECISION_BF16X3 || plan->precision == MSDF_PRECISION_BF16X6) return msdf_b16_color_forward(plan, a, (hipStream_t)stream);
  if (color_prepare((const void*)msdf_color_forward_k)) return MSDF_ERR_LAUNCH;
  msdf_color_forward_k<<<a->P_pad / MLP_PTS_PER_WG, MLP_THREADS, MLP_LDS_BYTES, (hipStream_t)stream>>>(*plan, *a);
  return msdf_check_launch();
}

extern "C" int msdf_color_backward(const msdf_plan_t* plan, const msdf_color_bwd_args_t* a, void* stream) {
  if (plan == nullptr || a == nullptr || a->P < 0) return MSDF_ERR_ARG;
  if (a->P == 0) return MSDF_OK;
  if (a->P_pad < a->P || (a->P_pad % MLP_PTS_PER_WG) != 0) return MSDF_ERR_ARG;
  if (plan->precision == MSDF_PRECISION_BF16X3 || plan->precision == MSDF_PRECISION_BF16X6) return msdf_b16_color_backward(plan, a, (hipStream_t)stream);
  if (color_prepare((const void*)msdf_color_backward_k)) return MSDF_ERR_LAUNCH;
  msdf_color_backward_k<<<a->P_pad / MLP_PTS_PER_WG, MLP_THREADS, MLP_LDS_BYTES, (hipStream_t)stream>>>(*plan, *a);
  return msdf_check_launch();
}
