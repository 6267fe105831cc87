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
__device__ __forceinline__ void softplus100(float a, float& h, float& s) {
  const float t = 100.0f * a;
  if (t > 20.0f) {
    h = a;
    s = 1.0f;
  } else {
    const float et = fast_exp(t);
    // log1p(et): series below 1e-4 keeps relative accuracy where 1+et rounds to 1
    const float l1p = (et < 1e-4f) ? et * (1.0f - 0.5f * et) : fast_log(1.0f + et);
    h = 0.01f * l1p;
    s = et / (1.0f + et);
  }
}

// from a saved post-activation h = softplus100(a): u = 1 - sigmoid(100 a) = exp(-100 h)
__device__ __forceinline__ float one_minus_sigmoid_from_h(float h) { return fast_exp(-100.0f * h); }

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
