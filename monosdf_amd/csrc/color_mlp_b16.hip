// bf16x3 instantiation of the colour-network kernels (color_kernels.h on CoreB16, mlp_core_b16.h).
#include "color_kernels.h"
#include "mlp_core_b16.h"

__global__ void __launch_bounds__(B16_THREADS, 2)
msdf_color_forward_b16_k(const msdf_plan_t plan, const ColorFwdArgs a) {
  extern __shared__ v8bf lds16[];
  color_forward_body<CoreB16>(plan, a, lds16);
}

__global__ void __launch_bounds__(B16_THREADS, 2)
msdf_color_backward_b16_k(const msdf_plan_t plan, const ColorBwdArgs a) {
  extern __shared__ v8bf lds16[];
  color_backward_body<CoreB16>(plan, a, lds16);
}

static int b16_prepare(const void* fn) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, B16_LDS_BYTES) != hipSuccess;
}

int msdf_b16_color_forward(const msdf_plan_t* plan, const msdf_color_fwd_args_t* a, hipStream_t stream) {
  if (b16_prepare((const void*)msdf_color_forward_b16_k)) return MSDF_ERR_LAUNCH;
  msdf_color_forward_b16_k<<<a->P_pad / B16_PTS_PER_WG, B16_THREADS, B16_LDS_BYTES, stream>>>(*plan, *a);
  return msdf_check_launch();
}

int msdf_b16_color_backward(const msdf_plan_t* plan, const msdf_color_bwd_args_t* a, hipStream_t stream) {
  if (b16_prepare((const void*)msdf_color_backward_b16_k)) return MSDF_ERR_LAUNCH;
  msdf_color_backward_b16_k<<<a->P_pad / B16_PTS_PER_WG, B16_THREADS, B16_LDS_BYTES, stream>>>(*plan, *a);
  return msdf_check_launch();
}
