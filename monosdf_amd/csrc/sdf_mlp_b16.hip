// bf16x3 / bf16x6 variants of the SDF-network kernels (mlp_core_b16.h).  Same plans / slot maps as the fp32
// kernels of sdf_mlp.hip; the plan handed to these kernels carries K-BLOCK counts (32 slots) in
// ktp / otp and 16-byte offsets of the bf16 plane packs (2 or 3 planes) in wf_off / wb_off.
#include "sdf_kernels.h"
#include "mlp_core_b16.h"

// flat effective weights -> bf16 plane packs in fragment order, both orientations, + fp32 bias / sdf row
template <int NS>
__global__ void __launch_bounds__(256) msdf_pack_b16_kernel(const msdf_plan_t plan,
                                                            const msdf_packrule_t* __restrict__ rules,
                                                            const int* __restrict__ maps,
                                                            const float* __restrict__ flat_w,
                                                            const float* __restrict__ flat_b,
                                                            v8bf* __restrict__ wpack, float* __restrict__ bpack) {
  const int l = blockIdx.y;
  const int which = blockIdx.z;
  const msdf_layer_t L = plan.layer[l];
  const msdf_packrule_t R = rules[l];
  const int* rowmap = maps + R.rowmap_off;
  const int* colmap = maps + R.colmap_off;
  const float* W = flat_w + R.w_off;
  const int stride = gridDim.x * blockDim.x;
  const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
  if (which < 2) {
    // which 0: rows = out slots, k = in slots (forward);  which 1: rows = in slots, k = out slots (transposed)
    const int n_rt = which == 0 ? L.ot : L.kt;          // row tiles
    const int n_kt = which == 0 ? L.kt : L.ot;          // k tiles (true count)
    const int kbp = which == 0 ? L.ktp : L.otp;         // k blocks in the pack
    const int off = which == 0 ? L.wf_off : L.wb_off;
    const int rt_even = (n_rt + 1) / 2 * 2;
    const int total = rt_even * kbp * 64;
    for (int i = t0; i < total; i += stride) {
      const int lane = i & 63;
      const int blk = i >> 6;
      const int kb = blk % kbp, rt = blk / kbp;
      const int rslot = 16 * rt + (lane & 15);
      v8bf pl[NS];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ktile = 2 * kb + (j >> 2);
        const int kslot = 16 * ktile + 4 * (lane >> 4) + (j & 3);
        float w = 0.f;
        if (rt < n_rt && ktile < n_kt) {
          const int row = which == 0 ? rowmap[rslot] : rowmap[kslot];
          const int col = which == 0 ? colmap[kslot] : colmap[rslot];
          if (row >= 0 && col >= 0) w = R.scale * W[(size_t)row * R.cols + col];
        }
#pragma unroll
        for (int n = 0; n < NS; ++n) {
          const __bf16 h = (__bf16)w;
          pl[n][j] = h;
          w -= (float)h;
        }
      }
      v8bf* dst = wpack + off + ((size_t)(rt * kbp + kb) * NS) * 64 + lane;
#pragma unroll
      for (int n = 0; n < NS; ++n) dst[n * 64] = pl[n];
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

template <int NS>
__global__ void __launch_bounds__(B16_THREADS, 2)
msdf_sdf_forward_b16_k(const msdf_plan_t plan, const v8bf* __restrict__ wpack, const float* __restrict__ bpack,
                       const float* __restrict__ x, const float* __restrict__ aux, const AuxView av, const int P,
                       const float clamp_radius, const float sphere_scale, float* __restrict__ sdf_out,
                       const uint32_t* __restrict__ run_flag) {
  extern __shared__ v8bf lds16[];
  if (run_flag != nullptr && *run_flag == 0u) return;
  sdf_forward_body<CoreB16N<NS>>(plan, wpack, bpack, x, aux, av, P, clamp_radius, sphere_scale, sdf_out, lds16);
}

template <int NS>
__global__ void __launch_bounds__(B16_THREADS, 2)
msdf_sdf_fwd_grad_b16_k(const msdf_plan_t plan, const FgArgs a) {
  extern __shared__ v8bf lds16[];
  sdf_fwd_grad_body<CoreB16N<NS>>(plan, a, lds16);
}

template <int NS>
__global__ void __launch_bounds__(B16_THREADS, 2)
msdf_sdf_backward_b16_k(const msdf_plan_t plan, const BwArgs a) {
  extern __shared__ v8bf lds16[];
  sdf_backward_body<CoreB16N<NS>>(plan, a, lds16);
}

template <int NS>
static int b16_prepare(const void* fn) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, B16Cfg<NS>::LDS_BYTES) != hipSuccess;
}

// launchers called by the precision dispatch of the C entry points (sdf_mlp.hip); plan->precision picks the planes
#define B16_PLANES(plan, ...)                                                                       \
  if ((plan)->precision == MSDF_PRECISION_BF16X6) { constexpr int NS = 3; __VA_ARGS__; } else { constexpr int NS = 2; __VA_ARGS__; }

int msdf_b16_pack_weights(const msdf_plan_t* plan, const msdf_packrule_t* rules_dev, const int* maps_dev,
                          const float* flat_w, const float* flat_b, void* wpack, float* bpack, hipStream_t stream) {
  const dim3 grid(32, plan->n_layers, 3);
  B16_PLANES(plan, (msdf_pack_b16_kernel<NS><<<grid, 256, 0, stream>>>(*plan, rules_dev, maps_dev, flat_w, flat_b,
                                                                       (v8bf*)wpack, bpack)));
  return msdf_check_launch();
}

int msdf_b16_sdf_forward(const msdf_plan_t* plan, const void* wpack, const float* bpack, const float* x,
                         const float* aux, int aux_C, int aux_LC, int P, float clamp_radius, float sphere_scale,
                         float* sdf, const uint32_t* run_flag, hipStream_t stream) {
  if (aux_C != 0) return MSDF_ERR_UNSUPPORTED;
  const int grid = (P + B16_PTS_PER_WG - 1) / B16_PTS_PER_WG;
  const AuxView av = {aux_C, aux_LC, P};
  B16_PLANES(plan, {
    if (b16_prepare<NS>((const void*)msdf_sdf_forward_b16_k<NS>)) return MSDF_ERR_LAUNCH;
    msdf_sdf_forward_b16_k<NS><<<grid, B16_THREADS, B16Cfg<NS>::LDS_BYTES, stream>>>(
        *plan, (const v8bf*)wpack, bpack, x, aux, av, P, clamp_radius, sphere_scale, sdf, run_flag);
  });
  return msdf_check_launch();
}

int msdf_b16_sdf_fwd_grad(const msdf_plan_t* plan, const msdf_fg_args_t* a, hipStream_t stream) {
  if (a->aux_C != 0 || a->dy_dx != nullptr) return MSDF_ERR_UNSUPPORTED;     // the bf16 cores take rows
  B16_PLANES(plan, {
    if (b16_prepare<NS>((const void*)msdf_sdf_fwd_grad_b16_k<NS>)) return MSDF_ERR_LAUNCH;
    msdf_sdf_fwd_grad_b16_k<NS><<<a->P_pad / B16_PTS_PER_WG, B16_THREADS, B16Cfg<NS>::LDS_BYTES, stream>>>(*plan, *a);
  });
  return msdf_check_launch();
}

int msdf_b16_sdf_backward(const msdf_plan_t* plan, const msdf_bw_args_t* a, hipStream_t stream) {
  if (a->aux_C != 0 || a->dy_dx != nullptr) return MSDF_ERR_UNSUPPORTED;
  B16_PLANES(plan, {
    if (b16_prepare<NS>((const void*)msdf_sdf_backward_b16_k<NS>)) return MSDF_ERR_LAUNCH;
    msdf_sdf_backward_b16_k<NS><<<a->P_pad / B16_PTS_PER_WG, B16_THREADS, B16Cfg<NS>::LDS_BYTES, stream>>>(*plan, *a);
  });
  return msdf_check_launch();
}
