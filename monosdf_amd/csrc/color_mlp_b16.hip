// bf16x3 / bf16x6 instantiations of the colour-network kernels (color_kernels.h on CoreB16N, mlp_core_b16.h).
#include "color_kernels.h"
#include "mlp_core_b16.h"

template <int NS>
__global__ void __launch_bounds__(B16_THREADS, 2)
msdf_color_forward_b16_k(const msdf_plan_t plan, const ColorFwdArgs a) {
  extern __shared__ v8bf lds16[];
  color_forward_body<CoreB16N<NS>>(plan, a, lds16);
}

template <int NS>
__global__ void __launch_bounds__(B16_THREADS, 2)
msdf_color_backward_b16_k(const msdf_plan_t plan, const ColorBwdArgs a) {
  extern __shared__ v8bf lds16[];
  color_backward_body<CoreB16N<NS>>(plan, a, lds16);
}

template <int NS>
static int b16_prepare(const void* fn) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, B16Cfg<NS>::LDS_BYTES) != hipSuccess;
}

#define B16_PLANES(plan, ...)                                                                       \
  if ((plan)->precision == MSDF_PRECISION_BF16X6) { constexpr int NS = 3; __VA_ARGS__; } else { constexpr int NS = 2; __VA_ARGS__; }

int msdf_b16_color_forward(const msdf_plan_t* plan, const msdf_color_fwd_args_t* a, hipStream_t stream) {
  B16_PLANES(plan, {
    if (b16_prepare<NS>((const void*)msdf_color_forward_b16_k<NS>)) return MSDF_ERR_LAUNCH;
    msdf_color_forward_b16_k<NS><<<a->P_pad / B16_PTS_PER_WG, B16_THREADS, B16Cfg<NS>::LDS_BYTES, stream>>>(*plan, *a);
  });
  return msdf_check_launch();
}

int msdf_b16_color_backward(const msdf_plan_t* plan, const msdf_color_bwd_args_t* a, hipStream_t stream) {
  B16_PLANES(plan, {
    if (b16_prepare<NS>((const void*)msdf_color_backward_b16_k<NS>)) return MSDF_ERR_LAUNCH;
    msdf_color_backward_b16_k<NS><<<a->P_pad / B16_PTS_PER_WG, B16_THREADS, B16Cfg<NS>::LDS_BYTES, stream>>>(*plan, *a);
  });
  return msdf_check_launch();
}
