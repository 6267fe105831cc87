// Weight normalisation of ALL layers of a network in one launch (forward and backward).
//
// The reference wraps every Linear in nn.utils.weight_norm (code/model/network.py:72-73, 239-240,
// 381-382): w = g * v / ||v||_row, recomputed by a forward pre-hook per layer per call, i.e. ~12 small
// launches forward and ~12 backward per network and step.  Here one wave handles one row of one layer:
// forward writes the effective weights (flat, the weight packer's input) and the row norms; backward
// turns d loss/d w into d loss/d v and d loss/d g:
//     dg = <dw, v> / n            dv = (g / n) dw - (g <dw, v> / n^3) v
// Layers without weight norm (has_g == 0) pass through.
#include "common.h"

__device__ __forceinline__ float wn_wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__global__ void __launch_bounds__(256)
msdf_weightnorm_forward_k(const msdf_wn_layer_t* __restrict__ layers, const int* __restrict__ row_layer,
                          const int total_rows, float* __restrict__ flat_w, float* __restrict__ flat_b,
                          float* __restrict__ norms) {
  const int row_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row_g >= total_rows) return;
  const int lane = lane_id();
  const msdf_wn_layer_t L = layers[row_layer[row_g]];
  const int r = row_g - L.row_off;
  const float* v = L.v + (size_t)r * L.cols;
  float* w = flat_w + L.w_off + (size_t)r * L.cols;
  float scale = 1.0f;
  if (L.has_g) {
    float ss = 0.f;
    for (int c = lane; c < L.cols; c += 64) ss += v[c] * v[c];
    const float n = sqrtf(wn_wave_sum(ss));
    scale = L.g[r] / n;
    if (lane == 0) norms[row_g] = n;
  }
  for (int c = lane; c < L.cols; c += 64) w[c] = v[c] * scale;
  if (lane == 0) flat_b[L.b_off + r] = L.b[r];
}

__global__ void __launch_bounds__(256)
msdf_weightnorm_backward_k(const msdf_wn_layer_t* __restrict__ layers, const int* __restrict__ row_layer,
                           const int total_rows, const float* __restrict__ g_flat_w,
                           const float* __restrict__ norms, float* __restrict__ dv_flat,
                           float* __restrict__ dg_rows) {
  const int row_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row_g >= total_rows) return;
  const int lane = lane_id();
  const msdf_wn_layer_t L = layers[row_layer[row_g]];
  const int r = row_g - L.row_off;
  const float* v = L.v + (size_t)r * L.cols;
  const float* dw = g_flat_w + L.w_off + (size_t)r * L.cols;
  float* dv = dv_flat + L.w_off + (size_t)r * L.cols;
  if (!L.has_g) {
    for (int c = lane; c < L.cols; c += 64) dv[c] = dw[c];
    if (lane == 0) dg_rows[row_g] = 0.f;
    return;
  }
  float dot = 0.f;
  for (int c = lane; c < L.cols; c += 64) dot += dw[c] * v[c];
  dot = wn_wave_sum(dot);
  const float n = norms[row_g];
  const float g = L.g[r];
  const float a = g / n, bq = g * dot / (n * n * n);
  for (int c = lane; c < L.cols; c += 64) dv[c] = a * dw[c] - bq * v[c];
  if (lane == 0) dg_rows[row_g] = dot / n;
}

extern "C" int msdf_weightnorm_forward(const msdf_wn_layer_t* layers_dev, const int* row_layer_dev, int total_rows,
                                       float* flat_w, float* flat_b, float* norms, void* stream) {
  if (total_rows < 0) return MSDF_ERR_ARG;
  if (total_rows == 0) return MSDF_OK;
  msdf_weightnorm_forward_k<<<(total_rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(layers_dev, row_layer_dev, total_rows,
                                                                                    flat_w, flat_b, norms);
  return msdf_check_launch();
}

extern "C" int msdf_weightnorm_backward(const msdf_wn_layer_t* layers_dev, const int* row_layer_dev, int total_rows,
                                        const float* g_flat_w, const float* norms, float* dv_flat, float* dg_rows,
                                        void* stream) {
  if (total_rows < 0) return MSDF_ERR_ARG;
  if (total_rows == 0) return MSDF_OK;
  msdf_weightnorm_backward_k<<<(total_rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(
      layers_dev, row_layer_dev, total_rows, g_flat_w, norms, dv_flat, dg_rows);
  return msdf_check_launch();
}
