// Fused scalar loss used by bench.py (BASELINE.md section 2 "probe loss") with its gradients in ONE pass:
//   L = mean|rgb| + w_n mean|normal| + w_d mean(depth) + w_e mean((|g1|-1)^2) + w_s mean|n1 - n2|,
//   n_i = g_i / (|g_i| + 1e-5),  g1 = grad_theta, g2 = grad_theta_nei.
// Every term is a mean of per-row functions, so each row's gradient is local; the loss value is
// reduced per block and summed on the host side of the stream (deterministic).
// The same structure (per-ray terms + eikonal/smooth terms on the SDF gradients) is what the reference's
// MonoSDFLoss has (code/model/loss.py:180-311); porting that loss is the "next" row 8(f)-2.
#include "common.h"

__device__ __forceinline__ float sgnf(const float v) { return (v > 0.f) ? 1.f : (v < 0.f) ? -1.f : 0.f; }

__global__ void __launch_bounds__(256)
msdf_probe_loss_k(const msdf_probe_loss_args_t a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float part = 0.f;
  if (i < a.N) {
    const float inv3n = 1.0f / (3.0f * a.N);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float r = a.rgb[(size_t)i * 3 + c];
      part += fabsf(r) * inv3n;
      a.g_rgb[(size_t)i * 3 + c] = sgnf(r) * inv3n;
      const float m = a.nrm[(size_t)i * 3 + c];
      part += a.w_normal * fabsf(m) * inv3n;
      a.g_nrm[(size_t)i * 3 + c] = a.w_normal * sgnf(m) * inv3n;
    }
    part += a.w_depth * a.depth[i] / a.N;
    a.g_depth[i] = a.w_depth / a.N;
  }
  if (i < a.M) {
    const float invm = 1.0f / a.M;
    const float* g1 = a.g1 + (size_t)i * 3;
    const float* g2 = a.g2 + (size_t)i * 3;
    const float l1 = sqrtf(g1[0] * g1[0] + g1[1] * g1[1] + g1[2] * g1[2]);
    const float l2 = sqrtf(g2[0] * g2[0] + g2[1] * g2[1] + g2[2] * g2[2]);
    const float e = l1 - 1.0f;
    part += a.w_eik * e * e * invm;
    const float d1 = l1 + 1e-5f, d2 = l2 + 1e-5f;
    float n1[3], n2[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { n1[c] = g1[c] / d1; n2[c] = g2[c] / d2; d[c] = n1[c] - n2[c]; }
    const float s = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    part += a.w_smooth * s * invm;
    // u = d / s (0 at s == 0); d loss/d n1 = +k u, d loss/d n2 = -k u; n = g/(|g|+eps)
    const float k = (s > 0.f) ? a.w_smooth * invm / s : 0.f;
    const float ug1 = (d[0] * g1[0] + d[1] * g1[1] + d[2] * g1[2]);
    const float ug2 = (d[0] * g2[0] + d[1] * g2[1] + d[2] * g2[2]);
    const float eik = (l1 > 0.f) ? a.w_eik * 2.0f * e * invm / l1 : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float j1 = d[c] / d1 - ((l1 > 0.f) ? g1[c] * ug1 / (l1 * d1 * d1) : 0.f);
      const float j2 = d[c] / d2 - ((l2 > 0.f) ? g2[c] * ug2 / (l2 * d2 * d2) : 0.f);
      a.g_g1[(size_t)i * 3 + c] = eik * g1[c] + k * j1;
      a.g_g2[(size_t)i * 3 + c] = -k * j2;
    }
  }
  // block reduction of the loss value
  __shared__ float red[4];
#pragma unroll
  for (int dlt = 32; dlt >= 1; dlt >>= 1) part += __shfl_xor(part, dlt, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) a.partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

extern "C" int msdf_probe_loss(const msdf_probe_loss_args_t* a, void* stream) {
  if (a == nullptr || a->N < 1 || a->M < 0) return MSDF_ERR_ARG;
  const int n = a->N > a->M ? a->N : a->M;
  msdf_probe_loss_k<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(*a);
  return msdf_check_launch();
}
