// Shared device helpers for the MonoSDF gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/monosdf_hip.h"

typedef float v4f __attribute__((ext_vector_type(4)));

#define MSDF_WAVE 64

static inline int msdf_check_launch() {
  return hipGetLastError() == hipSuccess ? MSDF_OK : MSDF_ERR_LAUNCH;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// exp / log on the hardware transcendental unit (v_exp_f32 / v_log_f32, ~1 ulp)
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fast_log(float x) { return __logf(x); }

// nn.Softplus(beta=100, threshold=20): value h, derivative s = sigmoid(100 a)
// (reference: code/model/network.py:77; torch softplus forward/backward semantics).
// Branch-free on purpose (two raw transcendental ops + selects): this runs 64 times per lane per layer
// between matrix products, and divergent control flow there costs more than the arithmetic.
//   log1p(exp(t)) = max(t, 0) + log1p(exp(-|t|)),  e = exp(-|t|) in (0, 1] never overflows.
__device__ __forceinline__ void softplus100(float a, float& h, float& s) {
  const float e = __builtin_amdgcn_exp2f(-fabsf(a * 144.26950408889634f));   // exp(-|100 a|)
  const float ope = 1.0f + e;
  // log1p(e): series below 1e-4 keeps relative accuracy where 1+e rounds to 1
  const float l_series = e * fmaf(e, -0.5f, 1.0f);
  const float l_log = __builtin_amdgcn_logf(ope) * 0.6931471805599453f;      // v_log_f32 is log2
  const float l1p = (e < 1e-4f) ? l_series : l_log;
  const float hv = fmaf(0.01f, l1p, fmaxf(a, 0.0f));
  h = (a * 100.0f > 20.0f) ? a : hv;
  const float r = __builtin_amdgcn_rcpf(ope);
  s = (a >= 0.0f) ? r : e * r;
}

// from a saved post-activation h = softplus100(a) >= 0: u = 1 - sigmoid(100 a) = exp(-100 h)
__device__ __forceinline__ float one_minus_sigmoid_from_h(float h) {
  return __builtin_amdgcn_exp2f(h * -144.26950408889634f);
}

// butterfly sum over the 4 lanes {l, l^16, l^32, l^48} (same point, different k-quarter)
__device__ __forceinline__ float sum_over_quarters(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// sum over the 16 lanes sharing the same quarter (the 16 points of a wave tile)
__device__ __forceinline__ float sum_over_points16(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}
