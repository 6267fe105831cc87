// Volume-rendering compositor for gfx950: Laplace density -> free energy -> transmittance
// (wavefront exclusive prefix scan) -> weights -> per-ray reductions, forward and backward.
// One wave per ray, ray-major [N, S] sample layout (coalesced 4-byte-per-lane rows).
//
// Reference: LaplaceDensity.density_func (code/model/density.py:21-30),
// MonoSDFNetwork.volume_rendering (code/model/network.py:626-640) and the composites at
// network.py:552-562, 603-605 (rgb, depth, white background, normal map before the pose rotation).
#include "common.h"

#define CMP_MAX_PER_LANE 4      // supports up to 256 samples per ray

__device__ __forceinline__ float wave_incl_scan(float v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// exclusive prefix sum; `total` = sum over the wave.  (inclusive - self would cancel catastrophically
// against the 1e10 * sigma free energy of the last interval)
__device__ __forceinline__ float wave_excl_scan(const float v, float& total) {
  float prev = __shfl_up(v, 1, 64);
  if (lane_id() == 0) prev = 0.f;
  const float ex = wave_incl_scan(prev);
  total = __shfl(ex, 63, 64) + __shfl(v, 63, 64);
  return ex;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// sigma(s) = (1/beta) (0.5 + 0.5 sign(s) expm1(-|s|/beta))
__device__ __forceinline__ float laplace_density(const float s, const float beta) {
  const float e = expm1f(-fabsf(s) / beta);
  const float sg = (s > 0.f) ? 1.f : (s < 0.f) ? -1.f : 0.f;
  return (1.0f / beta) * (0.5f + 0.5f * sg * e);
}

typedef msdf_composite_args_t CompositeArgs;

__global__ void __launch_bounds__(256)
msdf_composite_forward_k(const CompositeArgs a) {
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int lane = lane_id();
  const int S = a.S;
  const float beta = a.beta[0];
  const float* z = a.z + (size_t)ray * S;
  const float* sd = a.sdf + (size_t)ray * S;
  const float dscale = a.depth_scale[(size_t)ray * (a.depth_scale_stride > 0 ? a.depth_scale_stride : 1)];
  float carry = 0.f;                 // inclusive sum of free energy of earlier chunks
  float r0 = 0.f, r1 = 0.f, r2 = 0.f, wz = 0.f, ws = 0.f, m0 = 0.f, m1 = 0.f, m2 = 0.f;
  for (int base = 0; base < S; base += 64) {
    const int i = base + lane;
    const bool ok = i < S;
    float fe = 0.f, w = 0.f;
    if (ok) {
      const float zi = z[i];
      const float dist = (i + 1 < S) ? (z[i + 1] - zi) : 1e10f;
      fe = dist * laplace_density(sd[i], beta);
    }
    float chunk_total;
    const float ex = wave_excl_scan(fe, chunk_total);
    if (ok) {
      const float excl = carry + ex;
      const float alpha = 1.0f - expf(-fe);
      const float trans = expf(-excl);
      w = alpha * trans;
      a.weights[(size_t)ray * S + i] = w;
      if (a.depth_vals != nullptr) a.depth_vals[(size_t)ray * S + i] = z[i] * dscale;
      const float* c = a.rgb + ((size_t)ray * S + i) * 3;
      const float* n = a.nrm + ((size_t)ray * S + i) * 3;
      r0 += w * c[0]; r1 += w * c[1]; r2 += w * c[2];
      wz += w * z[i];
      ws += w;
      const float nn = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) + 1e-6f;
      m0 += w * (n[0] / nn); m1 += w * (n[1] / nn); m2 += w * (n[2] / nn);
    }
    carry += chunk_total;
  }
  r0 = wave_sum(r0); r1 = wave_sum(r1); r2 = wave_sum(r2);
  wz = wave_sum(wz); ws = wave_sum(ws);
  m0 = wave_sum(m0); m1 = wave_sum(m1); m2 = wave_sum(m2);
  if (lane == 0) {
    if (a.white_bkgd) {
      r0 += (1.f - ws) * a.bg0; r1 += (1.f - ws) * a.bg1; r2 += (1.f - ws) * a.bg2;
    }
    a.rgb_values[(size_t)ray * 3 + 0] = r0;
    a.rgb_values[(size_t)ray * 3 + 1] = r1;
    a.rgb_values[(size_t)ray * 3 + 2] = r2;
    a.depth_values[ray] = dscale * (wz / (ws + 1e-8f));
    if (a.pose != nullptr) {
      // rotate into the camera frame: out = R^T m, R = pose[:3,:3]
      const float* R = a.pose + (size_t)ray * a.pose_stride;
      const float c0 = R[0] * m0 + R[4] * m1 + R[8] * m2;
      const float c1 = R[1] * m0 + R[5] * m1 + R[9] * m2;
      const float c2 = R[2] * m0 + R[6] * m1 + R[10] * m2;
      m0 = c0; m1 = c1; m2 = c2;
    }
    a.normal_map[(size_t)ray * 3 + 0] = m0;
    a.normal_map[(size_t)ray * 3 + 1] = m1;
    a.normal_map[(size_t)ray * 3 + 2] = m2;
    a.wsum[ray] = ws;
  }
}

typedef msdf_composite_bwd_args_t CompositeBwdArgs;

// exclusive suffix sum: sum of v over lanes > lane; `total` = sum over the wave.  Built from a shifted
// reverse scan so that a lane with nothing after it gets EXACTLY 0 (total - inclusive would leave
// rounding noise, which the 1e10 last-interval length then multiplies into the sdf / beta gradients).
__device__ __forceinline__ float wave_excl_suffix(const float v, float& total) {
  const int lane = lane_id();
  float s = __shfl_down(v, 1, 64);
  if (lane == 63) s = 0.f;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_down(s, d, 64);
    if (lane + d < 64) s += o;
  }
  total = __shfl(s, 0, 64) + __shfl(v, 0, 64);
  return s;
}

__global__ void __launch_bounds__(256)
msdf_composite_backward_k(const CompositeBwdArgs a) {
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int lane = lane_id();
  const int S = a.S;
  const float beta = a.beta[0];
  const float* z = a.z + (size_t)ray * S;
  const float* sd = a.sdf + (size_t)ray * S;
  const float ws = a.wsum[ray];
  const float ds = a.depth_scale[(size_t)ray * (a.depth_scale_stride > 0 ? a.depth_scale_stride : 1)];
  float gR0 = 0.f, gR1 = 0.f, gR2 = 0.f, gD = 0.f, gM0 = 0.f, gM1 = 0.f, gM2 = 0.f;
  if (a.g_rgb_values) {
    gR0 = a.g_rgb_values[(size_t)ray * 3 + 0];
    gR1 = a.g_rgb_values[(size_t)ray * 3 + 1];
    gR2 = a.g_rgb_values[(size_t)ray * 3 + 2];
  }
  if (a.g_depth) gD = a.g_depth[ray];
  if (a.g_normal) {
    gM0 = a.g_normal[(size_t)ray * 3 + 0];
    gM1 = a.g_normal[(size_t)ray * 3 + 1];
    gM2 = a.g_normal[(size_t)ray * 3 + 2];
    if (a.pose != nullptr) {
      // adjoint of out = R^T m:  g_m = R g_out
      const float* R = a.pose + (size_t)ray * a.pose_stride;
      const float w0 = R[0] * gM0 + R[1] * gM1 + R[2] * gM2;
      const float w1 = R[4] * gM0 + R[5] * gM1 + R[6] * gM2;
      const float w2 = R[8] * gM0 + R[9] * gM1 + R[10] * gM2;
      gM0 = w0; gM1 = w1; gM2 = w2;
    }
  }
  // depth = ds * wz / (ws + eps):  d/dw_i = ds * (z_i - wz/(ws+eps)) / (ws + eps)
  const float inv = 1.0f / (ws + 1e-8f);
  const float wz_over = (ds != 0.f) ? a.depth_values[ray] / ds : 0.f;   // = wz / (ws + eps)
  float bg_term = 0.f;
  if (a.white_bkgd) bg_term = -(gR0 * a.bg0 + gR1 * a.bg1 + gR2 * a.bg2);

  // pass 1 (descending chunks): need suffix sums of C-bar over samples after i.
  // Per sample keep (per lane, per chunk) fe, trans, alpha, w-bar; S <= 64*CMP_MAX_PER_LANE.
  float fe_[CMP_MAX_PER_LANE], tr_[CMP_MAX_PER_LANE], wb_[CMP_MAX_PER_LANE], dist_[CMP_MAX_PER_LANE];
  float carry = 0.f;
  const int nchunk = (S + 63) >> 6;
#pragma unroll
  for (int ch = 0; ch < CMP_MAX_PER_LANE; ++ch) {
    fe_[ch] = tr_[ch] = wb_[ch] = dist_[ch] = 0.f;
    if (ch < nchunk) {
      const int i = ch * 64 + lane;
      const bool ok = i < S;
      float fe = 0.f, dist = 0.f;
      if (ok) {
        const float zi = z[i];
        dist = (i + 1 < S) ? (z[i + 1] - zi) : 1e10f;
        fe = dist * laplace_density(sd[i], beta);
      }
      float chunk_total;
      const float ex = wave_excl_scan(fe, chunk_total);
      if (ok) {
        const float excl = carry + ex;
        const float trans = expf(-excl);
        const float w = a.weights[(size_t)ray * S + i];
        const float* c = a.rgb + ((size_t)ray * S + i) * 3;
        const float* n = a.nrm + ((size_t)ray * S + i) * 3;
        const float nl = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        const float nn = nl + 1e-6f;
        const float gdot = gM0 * n[0] + gM1 * n[1] + gM2 * n[2];
        float wbar = gR0 * c[0] + gR1 * c[1] + gR2 * c[2] + gD * ds * (z[i] - wz_over) * inv + gdot / nn + bg_term;
        if (a.g_weights) wbar += a.g_weights[(size_t)ray * S + i];
        // colour and normal gradients are local
        float* gc = a.g_rgb + ((size_t)ray * S + i) * 3;
        gc[0] = w * gR0; gc[1] = w * gR1; gc[2] = w * gR2;
        float* gn = a.g_nrm + ((size_t)ray * S + i) * 3;
        const float k = (nl > 0.f) ? gdot / (nl * nn * nn) : 0.f;
        gn[0] = w * (gM0 / nn - n[0] * k);
        gn[1] = w * (gM1 / nn - n[1] * k);
        gn[2] = w * (gM2 / nn - n[2] * k);
        fe_[ch] = fe; tr_[ch] = trans; wb_[ch] = wbar; dist_[ch] = dist;
      }
      carry += chunk_total;
    }
  }
  // pass 2 (last chunk to first): E-bar_i = alpha-bar_i e^{-E_i} + sum_{k>i} C-bar_k,  C-bar_k = -w-bar_k alpha_k T_k
  float after = 0.f;   // sum of C-bar over later chunks
  float gbeta = 0.f;
#pragma unroll
  for (int ch = CMP_MAX_PER_LANE - 1; ch >= 0; --ch) {
    if (ch < nchunk) {
      const int i = ch * 64 + lane;
      const bool ok = i < S;
      const float fe = fe_[ch], trans = tr_[ch], wbar = wb_[ch];
      const float efe = expf(-fe);
      const float alpha = 1.0f - efe;
      const float cbar = ok ? (-wbar * alpha * trans) : 0.f;
      float total;
      const float suffix = wave_excl_suffix(cbar, total);
      if (ok) {
        const float ebar = wbar * trans * efe + suffix + after;
        const float sbar = ebar * dist_[ch];                 // d loss / d sigma_i
        const float s = sd[i];
        const float ex = expf(-fabsf(s) / beta);
        // d sigma / d s = -(0.5/beta^2) e^{-|s|/beta};  d sigma / d beta = -sigma/beta + 0.5 s e^{-|s|/beta} / beta^3
        a.g_sdf[(size_t)ray * S + i] = sbar * (-0.5f / (beta * beta)) * ex;
        const float sigma = laplace_density(s, beta);
        gbeta += sbar * (-sigma / beta + 0.5f * s * ex / (beta * beta * beta));
      }
      after += total;
    }
  }
  gbeta = wave_sum(gbeta);
  if (lane == 0) a.g_beta_part[ray] = gbeta;
}

// |beta| + beta_min, and its adjoint with the sum over the compositor's per-ray partials (one wave, fixed order)
__global__ void msdf_beta_eff_k(const float* __restrict__ raw, const float beta_min, float* __restrict__ out) {
  if (threadIdx.x == 0) out[0] = fabsf(raw[0]) + beta_min;
}
__global__ void __launch_bounds__(64)
msdf_beta_grad_k(const float* __restrict__ raw, const float* __restrict__ part, const int n, float* __restrict__ g_raw) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += part[i];
  s = wave_sum(s);
  const float r = raw[0];
  if (threadIdx.x == 0) g_raw[0] = ((r > 0.f) ? 1.f : (r < 0.f) ? -1.f : 0.f) * s;
}
extern "C" int msdf_beta_eff(const float* beta_raw, float beta_min, float* out, void* stream) {
  if (beta_raw == nullptr || out == nullptr) return MSDF_ERR_ARG;
  msdf_beta_eff_k<<<1, 64, 0, (hipStream_t)stream>>>(beta_raw, beta_min, out);
  return msdf_check_launch();
}
extern "C" int msdf_beta_grad(const float* beta_raw, const float* g_part, int n, float* g_raw, void* stream) {
  if (beta_raw == nullptr || g_raw == nullptr || n < 0 || (n > 0 && g_part == nullptr)) return MSDF_ERR_ARG;
  msdf_beta_grad_k<<<1, 64, 0, (hipStream_t)stream>>>(beta_raw, g_part, n, g_raw);
  return msdf_check_launch();
}

extern "C" int msdf_composite_forward(const msdf_composite_args_t* a, void* stream) {
  if (a == nullptr || a->N < 0 || a->S < 1 || a->S > 64 * CMP_MAX_PER_LANE) return MSDF_ERR_ARG;
  if (a->N == 0) return MSDF_OK;
  msdf_composite_forward_k<<<(a->N + 3) / 4, 256, 0, (hipStream_t)stream>>>(*a);
  return msdf_check_launch();
}

extern "C" int msdf_composite_backward(const msdf_composite_bwd_args_t* a, void* stream) {
  if (a == nullptr || a->N < 0 || a->S < 1 || a->S > 64 * CMP_MAX_PER_LANE) return MSDF_ERR_ARG;
  if (a->N == 0) return MSDF_OK;
  msdf_composite_backward_k<<<(a->N + 3) / 4, 256, 0, (hipStream_t)stream>>>(*a);
  return msdf_check_launch();
}
