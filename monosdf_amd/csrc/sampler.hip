// Error-bounded ray sampler (VolSDF Algorithm 1) for gfx950: one wave per ray, the ray's
// sorted samples live in LDS, every cumulative sum is a wavefront prefix scan.
//
// Reference: UniformSampler.near_far_from_cube / get_z_vals (code/model/ray_sampler.py:48-83),
// ErrorBoundSampler.get_z_vals (110-262) and get_error_bound (264-272).  The reference runs ~15
// small launches per error-bound evaluation, 11 evaluations per round, with boolean-mask
// indexing syncs; here a round is two launches (the batch-global `beta.max() > beta0` test of
// ray_sampler.py:179 needs one grid-wide dependency) and one 4-byte device->host read.
//
//   msdf_sampler_init      uniform samples (+ stratified jitter), Lemma-2 beta, first batch of points
//   [msdf_sdf_forward on the new points]
//   msdf_sampler_beta      merge in the new sdf values, d*, per-ray bisection on beta, global max
//   msdf_sampler_resample  density / transmittance / pdf / inverse CDF, then either merge 128 new
//                          samples into the sorted set (another round) or emit the final 64
//   msdf_sampler_finish    final 64 + near + far + 32 extra -> sorted [N, 98], eikonal sample, points
//
// The round decisions live on the device: flags[2r] = bits of max beta after round r, flags[2r+1] = 1 when round
// r asks for another one.  The kernels of round r > 0 return at once when round r-1 did not ask for them, and the
// finish kernel derives the size of the dense set from the flags.  A host that enqueues more rounds than needed
// (it may choose not to read the flags back between rounds) therefore gets the same result as one that stops in
// time; it only has to check afterwards that the last round it enqueued did not ask for one more.
#include "common.h"

#define SMP_WAVES 4

__device__ __forceinline__ float smp_scan_incl(float v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ float smp_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ float smp_max(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ float smp_density(const float s, const float beta) {
  const float e = expm1f(-fabsf(s) / beta);
  const float sg = (s > 0.f) ? 1.f : (s < 0.f) ? -1.f : 0.f;
  return (1.0f / beta) * (0.5f + 0.5f * sg * e);
}
// torch.linspace(0, 1, n)[j] (symmetric evaluation, like ATen's kernel)
__device__ __forceinline__ float smp_linspace01(const int j, const int n) {
  const float step = 1.0f / (float)(n - 1);
  return (j < n / 2) ? step * (float)j : 1.0f - step * (float)(n - 1 - j);
}

typedef msdf_sampler_args_t SmpArgs;

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64 * SMP_WAVES) smp_init_k(const SmpArgs a) {
  const int ray = blockIdx.x * SMP_WAVES + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int lane = lane_id();
  // the round flags start at zero: no launch of this call reads them before the round kernels behind this one
  if (ray == 0 && a.flags != nullptr)
    for (int i = lane; i < 2 * a.max_rounds; i += 64) a.flags[i] = 0u;
  const int n = a.n_eval;
  const float o0 = a.ray_o[ray * 3 + 0], o1 = a.ray_o[ray * 3 + 1], o2 = a.ray_o[ray * 3 + 2];
  const float d0 = a.ray_d[ray * 3 + 0], d1 = a.ray_d[ray * 3 + 1], d2 = a.ray_d[ray * 3 + 2];
  // slab test against the cube [-bound, bound]^3
  const float ta0 = (-a.bound - o0) / (d0 + 1e-15f), tb0 = (a.bound - o0) / (d0 + 1e-15f);
  const float ta1 = (-a.bound - o1) / (d1 + 1e-15f), tb1 = (a.bound - o1) / (d1 + 1e-15f);
  const float ta2 = (-a.bound - o2) / (d2 + 1e-15f), tb2 = (a.bound - o2) / (d2 + 1e-15f);
  float near_c = fmaxf(fmaxf(fminf(ta0, tb0), fminf(ta1, tb1)), fminf(ta2, tb2));
  float far_c = fminf(fminf(fmaxf(ta0, tb0), fmaxf(ta1, tb1)), fmaxf(ta2, tb2));
  if (far_c < near_c) far_c = 1e9f;
  // bound <= 0: no cube test, the constant far of UniformSampler(take_sphere_intersection=False) (ray_sampler.py:64-65)
  const float far = (a.bound > 0.f) ? fminf(far_c, a.far) : a.far;
  const float near = a.near;
  float sumsq = 0.f;
  float* zrow = a.z + (size_t)ray * a.m_max;
  auto zu = [&](int k) { const float t = smp_linspace01(k, n); return near * (1.f - t) + far * t; };
  auto zfin = [&](int k) {
    float z = zu(k);
    if (a.jitter != nullptr) {
      const float zl = (k > 0) ? 0.5f * (zu(k) + zu(k - 1)) : zu(0);
      const float zh = (k + 1 < n) ? 0.5f * (zu(k + 1) + zu(k)) : zu(n - 1);
      z = zl + (zh - zl) * a.jitter[(size_t)ray * n + k];
    }
    return z;
  };
  for (int j = lane; j < n; j += 64) {
    const float z = zfin(j);
    zrow[j] = z;
    a.new_z[(size_t)ray * n + j] = z;
    a.new_pos[(size_t)ray * n + j] = j;
    float* p = a.pts + ((size_t)ray * n + j) * 3;
    p[0] = o0 + z * d0; p[1] = o1 + z * d1; p[2] = o2 + z * d2;
    if (j + 1 < n) {
      const float dz = zfin(j + 1) - z;
      sumsq += dz * dz;
    }
  }
  sumsq = smp_sum(sumsq);
  if (lane == 0) {
    a.beta[ray] = sqrtf(a.lemma * sumsq);
    if (a.far_out != nullptr) a.far_out[ray] = far;
  }
}

// ---------------------------------------------------------------------------
// per-wave LDS view
struct SmpLds {
  float* z; float* sdf; float* dist; float* dstar; float* tmp; float* snew;
};
__device__ __forceinline__ SmpLds smp_lds(float* base, const int m_max, const int n_eval) {
  const int w = threadIdx.x >> 6;
  const int per = 5 * (m_max + 1) + n_eval;
  float* p = base + (size_t)w * per;
  SmpLds l;
  l.z = p; l.sdf = p + (m_max + 1); l.dist = p + 2 * (m_max + 1); l.dstar = p + 3 * (m_max + 1);
  l.tmp = p + 4 * (m_max + 1); l.snew = p + 5 * (m_max + 1);
  return l;
}
__device__ __forceinline__ void smp_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// round r runs only if round r-1 asked for it (flags are zero-initialised by the host)
__device__ __forceinline__ bool smp_round_active(const SmpArgs& a) {
  return a.round_idx == 0 || a.flags[2 * (a.round_idx - 1) + 1] != 0u;
}

// load z / sdf of the ray (after scattering the freshly evaluated sdf values), build dists and d*
__device__ __forceinline__ void smp_load_ray(const SmpArgs& a, const SmpLds& l, const int ray, const bool scatter) {
  const int lane = lane_id();
  const int M = a.M;
  const float* zrow = a.z + (size_t)ray * a.m_max;
  float* srow = a.sdf + (size_t)ray * a.m_max;
  for (int i = lane; i < M; i += 64) { l.z[i] = zrow[i]; l.sdf[i] = srow[i]; }
  smp_sync();
  if (scatter) {
    for (int j = lane; j < a.n_eval; j += 64)
      l.sdf[a.new_pos[(size_t)ray * a.n_eval + j]] = a.new_sdf[(size_t)ray * a.n_eval + j];
    smp_sync();
    for (int i = lane; i < M; i += 64) srow[i] = l.sdf[i];
  }
  for (int i = lane; i + 1 < M; i += 64) {
    const float aa = l.z[i + 1] - l.z[i];
    const float b = fabsf(l.sdf[i]), c = fabsf(l.sdf[i + 1]);
    const bool first = aa * aa + b * b <= c * c;
    const bool second = aa * aa + c * c <= b * b;
    float ds = 0.f;
    if (first) ds = b;
    if (second) ds = c;
    if (!first && !second && (b + c - aa > 0.f)) {
      const float s = (aa + b + c) / 2.0f;
      const float area = s * (s - aa) * (s - b) * (s - c);
      ds = (2.0f * sqrtf(area)) / aa;
    }
    const float s0 = l.sdf[i], s1 = l.sdf[i + 1];
    const float sg0 = (s0 > 0.f) ? 1.f : (s0 < 0.f) ? -1.f : 0.f;
    const float sg1 = (s1 > 0.f) ? 1.f : (s1 < 0.f) ? -1.f : 0.f;
    l.dist[i] = aa;
    const float dsv = (sg0 * sg1 == 1.f) ? ds : 0.f;
    l.dstar[i] = dsv;
    if (scatter && a.dbg_dstar != nullptr) a.dbg_dstar[(size_t)ray * a.m_max + i] = dsv;
  }
  smp_sync();
}

// max_j (min(exp(sum_{i<=j} e_i), 1e6) - 1) exp(-sum_{i<j} dist_i sigma_i)   (ray_sampler.py:264-272)
__device__ __forceinline__ float smp_error_bound(const SmpLds& l, const int M, const float beta) {
  const int lane = lane_id();
  float carry_fe = 0.f, carry_err = 0.f, best = -1e30f;
  const float inv4b2 = 1.0f / (4.0f * beta * beta);
  for (int base = 0; base < M - 1; base += 64) {
    const int i = base + lane;
    const bool ok = i < M - 1;
    float fe = 0.f, er = 0.f;
    if (ok) {
      const float d = l.dist[i];
      fe = d * smp_density(l.sdf[i], beta);
      er = expf(-l.dstar[i] / beta) * (d * d) * inv4b2;
    }
    const float fe_incl = smp_scan_incl(fe);
    const float er_incl = smp_scan_incl(er);
    if (ok) {
      const float integral = carry_fe + fe_incl - fe;          // exclusive
      const float errint = carry_err + er_incl;                // inclusive
      const float bound = (fminf(expf(errint), 1.0e6f) - 1.0f) * expf(-integral);
      best = fmaxf(best, bound);
    }
    carry_fe += __shfl(fe_incl, 63, 64);
    carry_err += __shfl(er_incl, 63, 64);
  }
  return smp_max(best);
}

__global__ void __launch_bounds__(64 * SMP_WAVES) smp_beta_k(const SmpArgs a) {
  extern __shared__ float smp_lds_mem[];
  const int ray = blockIdx.x * SMP_WAVES + (threadIdx.x >> 6);
  if (ray >= a.N || !smp_round_active(a)) return;
  const SmpLds l = smp_lds(smp_lds_mem, a.m_max, a.n_eval);
  smp_load_ray(a, l, ray, true);
  const float beta0 = a.beta0[0];
  float beta = a.beta[ray];
  const float e0 = smp_error_bound(l, a.M, beta0);
  if (a.dbg_err0 != nullptr && lane_id() == 0) a.dbg_err0[ray] = e0;
  if (e0 <= a.eps) beta = beta0;
  float lo = beta0, hi = beta;
  // a ray already inside the bound at beta0 has lo == hi == beta0: every bisection step would evaluate the bound at
  // beta0 again and leave hi where it is (the reference bisects all rays and masks, ray_sampler.py:152-167) --
  // one wave is one ray, so skipping them is a uniform branch
  const int iters = (e0 <= a.eps) ? 0 : a.beta_iters;
  for (int it = 0; it < iters; ++it) {
    const float mid = (lo + hi) / 2.0f;
    const float e = smp_error_bound(l, a.M, mid);
    if (e <= a.eps) hi = mid;
    if (e > a.eps) lo = mid;
  }
  if (lane_id() == 0) {
    a.beta[ray] = hi;
    atomicMax(a.flags + 2 * a.round_idx, __float_as_uint(hi));
  }
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64 * SMP_WAVES) smp_resample_k(const SmpArgs a) {
  extern __shared__ float smp_lds_mem[];
  const int ray = blockIdx.x * SMP_WAVES + (threadIdx.x >> 6);
  if (ray >= a.N || !smp_round_active(a)) return;
  const int lane = lane_id();
  const int M = a.M;
  const SmpLds l = smp_lds(smp_lds_mem, a.m_max, a.n_eval);
  const float beta0 = a.beta0[0];
  const float maxbeta = __uint_as_float(a.flags[2 * a.round_idx]);
  const bool unconverged = maxbeta > beta0;
  const bool more = unconverged && (a.round_idx + 1 < a.max_rounds);
  if (ray == 0 && lane == 0) a.flags[2 * a.round_idx + 1] = more ? 1u : 0u;
  smp_load_ray(a, l, ray, false);
  const float beta = a.beta[ray];
  const float inv4b2 = 1.0f / (4.0f * beta * beta);

  // pdf over the M-1 intervals into tmp[0..M-2]
  float carry_fe = 0.f, carry_err = 0.f, psum = 0.f;
  for (int base = 0; base < M; base += 64) {
    const int i = base + lane;
    const bool ok = i < M;
    float fe = 0.f, er = 0.f;
    if (ok) {
      const float d = (i + 1 < M) ? l.dist[i] : 1e10f;
      fe = d * smp_density(l.sdf[i], beta);
      if (more && i + 1 < M) er = expf(-l.dstar[i] / beta) * (l.dist[i] * l.dist[i]) * inv4b2;
    }
    float fprev = __shfl_up(fe, 1, 64);
    if (lane == 0) fprev = 0.f;
    const float fe_excl = smp_scan_incl(fprev);
    const float fe_incl = fe_excl + fe;
    const float er_incl = smp_scan_incl(er);
    if (ok && i + 1 < M) {
      const float trans = expf(-(carry_fe + fe_excl));
      float p;
      if (more) p = (fminf(expf(carry_err + er_incl), 1.0e6f) - 1.0f) * trans + a.add_tiny;
      else p = (1.0f - expf(-fe)) * trans + 1e-5f;
      l.tmp[i] = p;
      psum += p;
    }
    carry_fe += __shfl(fe_incl, 63, 64);
    carry_err += __shfl(er_incl, 63, 64);
  }
  psum = smp_sum(psum);
  smp_sync();
  // cdf[0] = 0, cdf[i+1] = cumsum(pdf / sum)[i]  -> stored in dstar-independent buffer: reuse tmp shifted
  // (cdf kept in `dstar` array from here on: d* is no longer needed)
  float carry = 0.f;
  for (int base = 0; base < M - 1; base += 64) {
    const int i = base + lane;
    const float p = (i < M - 1) ? l.tmp[i] / psum : 0.f;
    const float incl = smp_scan_incl(p);
    if (i < M - 1) l.dstar[i + 1] = carry + incl;
    carry += __shfl(incl, 63, 64);
  }
  if (lane == 0) l.dstar[0] = 0.f;
  smp_sync();
  const float* cdf = l.dstar;
  if (a.dbg_cdf != nullptr)
    for (int i = lane; i < M; i += 64) a.dbg_cdf[(size_t)ray * a.m_max + i] = cdf[i];

  // inverse CDF
  const int n_new = more ? a.n_eval : a.n_final;
  for (int j = lane; j < n_new; j += 64) {
    float u;
    if (more || !a.training || a.u_final == nullptr) u = smp_linspace01(j, n_new);
    else u = a.u_final[(size_t)ray * a.n_final + j];
    // searchsorted(right=True): number of cdf entries <= u
    int lo = 0, hi = M;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int below = max(lo - 1, 0), above = min(lo, M - 1);
    float denom = cdf[above] - cdf[below];
    if (denom < 1e-5f) denom = 1.0f;
    const float t = (u - cdf[below]) / denom;
    const float s = l.z[below] + t * (l.z[above] - l.z[below]);
    l.snew[j] = s;
  }
  smp_sync();

  if (!more) {
    for (int j = lane; j < n_new; j += 64) a.final_z[(size_t)ray * a.n_final + j] = l.snew[j];
    return;
  }
  // merge: stable ranks by counting (robust even if the new samples are not perfectly monotone)
  const float o0 = a.ray_o[ray * 3 + 0], o1 = a.ray_o[ray * 3 + 1], o2 = a.ray_o[ray * 3 + 2];
  const float d0 = a.ray_d[ray * 3 + 0], d1 = a.ray_d[ray * 3 + 1], d2 = a.ray_d[ray * 3 + 2];
  float* zrow = a.z + (size_t)ray * a.m_max;
  float* srow = a.sdf + (size_t)ray * a.m_max;
  for (int i = lane; i < M; i += 64) {
    const float zi = l.z[i];
    int cnt = 0;
    for (int k = 0; k < n_new; ++k) cnt += (l.snew[k] < zi) ? 1 : 0;
    zrow[i + cnt] = zi;
    srow[i + cnt] = l.sdf[i];
  }
  for (int j = lane; j < n_new; j += 64) {
    const float sj = l.snew[j];
    int cnt = 0;
    for (int i = 0; i < M; ++i) cnt += (l.z[i] <= sj) ? 1 : 0;
    for (int k = 0; k < n_new; ++k) cnt += (l.snew[k] < sj || (l.snew[k] == sj && k < j)) ? 1 : 0;
    zrow[cnt] = sj;
    a.new_z[(size_t)ray * a.n_eval + j] = sj;
    a.new_pos[(size_t)ray * a.n_eval + j] = cnt;
    float* p = a.pts + ((size_t)ray * a.n_eval + j) * 3;
    p[0] = o0 + sj * d0; p[1] = o1 + sj * d1; p[2] = o2 + sj * d2;
  }
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64 * SMP_WAVES) smp_finish_k(const SmpArgs a) {
  extern __shared__ float smp_lds_mem[];
  const int ray = blockIdx.x * SMP_WAVES + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int lane = lane_id();
  const int S = a.n_final + a.n_extra + 2;
  float* v = smp_lds_mem + (size_t)(threadIdx.x >> 6) * 2 * S;
  float* sorted = v + S;
  const float* zrow = a.z + (size_t)ray * a.m_max;
  // rounds that ran = 1 + the leading rounds that asked for another one; the dense set holds n_eval of them each.
  // Row (rounds - 1) of the extra_idx table holds the columns drawn for that size (ray_sampler.py:242-247).
  int rounds = 1;
  while (rounds < a.max_rounds && a.flags[2 * (rounds - 1) + 1] != 0u) ++rounds;
  const int64_t* extra = a.extra_idx + (size_t)(rounds - 1) * a.n_extra;
  if (a.rounds_out != nullptr && ray == 0 && lane == 0) a.rounds_out[0] = rounds;
  for (int j = lane; j < S; j += 64) {
    float x;
    if (j < a.n_final) x = a.final_z[(size_t)ray * a.n_final + j];
    else if (j == a.n_final) x = a.near;
    else if (j == a.n_final + 1) x = a.far;
    else x = zrow[min((int)extra[j - a.n_final - 2], rounds * a.n_eval - 1)];
    v[j] = x;
  }
  smp_sync();
  const float o0 = a.ray_o[ray * 3 + 0], o1 = a.ray_o[ray * 3 + 1], o2 = a.ray_o[ray * 3 + 2];
  const float d0 = a.ray_d[ray * 3 + 0], d1 = a.ray_d[ray * 3 + 1], d2 = a.ray_d[ray * 3 + 2];
  for (int j = lane; j < S; j += 64) {
    const float x = v[j];
    int r = 0;
    for (int k = 0; k < S; ++k) r += (v[k] < x || (v[k] == x && k < j)) ? 1 : 0;
    sorted[r] = x;
    a.z_out[(size_t)ray * S + r] = x;
    if (a.pts_out != nullptr) {
      float* p = a.pts_out + ((size_t)ray * S + r) * 3;
      p[0] = o0 + x * d0; p[1] = o1 + x * d1; p[2] = o2 + x * d2;
    }
  }
  smp_sync();
  if (lane == 0) {
    const int idx = (a.eik_idx != nullptr) ? (int)a.eik_idx[ray]
                    : (a.eik_u != nullptr) ? min(S - 1, (int)(a.eik_u[ray] * (float)S)) : 0;
    const float ze = sorted[idx];
    if (a.z_eik != nullptr) a.z_eik[ray] = ze;
    if (a.pts_out != nullptr && a.eik_uniform != nullptr) {
      // eikonal block (reference network.py:583-594): [uniform | near-surface | their jittered neighbours]
      float* e = a.pts_out + (size_t)a.N * S * 3;
      const float od[3] = {o0 + ze * d0, o1 + ze * d1, o2 + ze * d2};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float u = a.eik_uniform[(size_t)ray * 3 + c];
        if (a.eik_unit) u = (2.0f * u - 1.0f) * a.bound;
        e[((size_t)ray) * 3 + c] = u;
        e[((size_t)a.N + ray) * 3 + c] = od[c];
        e[((size_t)2 * a.N + ray) * 3 + c] = u + (a.nei_rand[(size_t)ray * 3 + c] - 0.5f) * 0.01f;
        e[((size_t)3 * a.N + ray) * 3 + c] = od[c] + (a.nei_rand[((size_t)a.N + ray) * 3 + c] - 0.5f) * 0.01f;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// ErrorBoundSampler.get_error_bound as an operator of its own (ray_sampler.py:264-272): one wave per ray
__global__ void __launch_bounds__(64 * SMP_WAVES) smp_error_bound_k(const float* __restrict__ z,
                                                                     const float* __restrict__ sdf,
                                                                     const float* __restrict__ dstar,
                                                                     const float* __restrict__ beta, const int beta_stride,
                                                                     const int N, const int M, float* __restrict__ out) {
  extern __shared__ float smp_lds_mem[];
  const int ray = blockIdx.x * SMP_WAVES + (threadIdx.x >> 6);
  if (ray >= N) return;
  const int lane = lane_id();
  const SmpLds l = smp_lds(smp_lds_mem, M, 0);
  for (int i = lane; i < M; i += 64) { l.z[i] = z[(size_t)ray * M + i]; l.sdf[i] = sdf[(size_t)ray * M + i]; }
  smp_sync();
  for (int i = lane; i + 1 < M; i += 64) {
    l.dist[i] = l.z[i + 1] - l.z[i];
    l.dstar[i] = dstar[(size_t)ray * (M - 1) + i];
  }
  smp_sync();
  const float e = smp_error_bound(l, M, beta[(size_t)ray * beta_stride]);
  if (lane == 0) out[ray] = e;
}

// LaplaceDensity.density_func and its derivatives (model/density.py:21-26); beta is one value (stride 0) or one per
// row of `cols` values (the per-ray beta of the sampler, ray_sampler.py:162,168)
__global__ void __launch_bounds__(256) laplace_density_k(const float* __restrict__ sdf, const float* __restrict__ beta,
                                                         const int beta_stride, const int64_t n, const int cols,
                                                         float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = smp_density(sdf[i], beta[(i / cols) * beta_stride]);
}
// g_sdf = g * d sigma / d s,  g_beta_part[i] = g * d sigma / d beta (summed per beta by the caller)
__global__ void __launch_bounds__(256) laplace_density_bwd_k(const float* __restrict__ sdf, const float* __restrict__ beta,
                                                             const int beta_stride, const int64_t n, const int cols,
                                                             const float* __restrict__ g, float* __restrict__ g_sdf,
                                                             float* __restrict__ g_beta_elem) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float s = sdf[i], b = beta[(i / cols) * beta_stride], gi = g[i];
    const float a = fabsf(s);
    const float ex = expf(-a / b);                       // d expm1(t)/dt = exp(t), t = -|s|/beta
    const float sg = (s > 0.f) ? 1.f : (s < 0.f) ? -1.f : 0.f;
    // sigma = (1/b)(1/2 + 1/2 sg expm1(-a/b));  |s|' = sg, sg^2 = 1 away from 0 (torch: sign(0) = 0 -> grad 0)
    g_sdf[i] = gi * (1.0f / b) * 0.5f * sg * ex * (-sg / b);
    const float em1 = expm1f(-a / b);
    g_beta_elem[i] = gi * (-(1.0f / (b * b)) * (0.5f + 0.5f * sg * em1) + (1.0f / b) * 0.5f * sg * ex * (a / (b * b)));
  }
}

// ---------------------------------------------------------------------------
static int smp_check(const msdf_sampler_args_t* a) {
  if (a == nullptr || a->N < 0 || a->n_eval < 2 || a->m_max < a->n_eval) return MSDF_ERR_ARG;
  return MSDF_OK;
}
static size_t smp_lds_bytes(const msdf_sampler_args_t* a) {
  return (size_t)SMP_WAVES * (5 * (a->m_max + 1) + a->n_eval) * sizeof(float);
}
template <typename K>
static int smp_launch(K kernel, const msdf_sampler_args_t* a, size_t lds, void* stream) {
  if (a->N == 0) return MSDF_OK;
  if (lds > 160 * 1024) return MSDF_ERR_UNSUPPORTED;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return MSDF_ERR_LAUNCH;
  kernel<<<(a->N + SMP_WAVES - 1) / SMP_WAVES, 64 * SMP_WAVES, lds, (hipStream_t)stream>>>(*a);
  return msdf_check_launch();
}

extern "C" int msdf_sampler_init(const msdf_sampler_args_t* a, void* stream) {
  if (smp_check(a)) return MSDF_ERR_ARG;
  return smp_launch(smp_init_k, a, 0, stream);
}
static int smp_round_check(const msdf_sampler_args_t* a) {
  return a->flags == nullptr || a->round_idx < 0 || a->round_idx >= a->max_rounds || a->M > a->m_max || a->M < 2;
}
extern "C" int msdf_sampler_beta(const msdf_sampler_args_t* a, void* stream) {
  if (smp_check(a) || smp_round_check(a)) return MSDF_ERR_ARG;
  return smp_launch(smp_beta_k, a, smp_lds_bytes(a), stream);
}
extern "C" int msdf_sampler_resample(const msdf_sampler_args_t* a, void* stream) {
  if (smp_check(a) || smp_round_check(a) || a->n_final > a->n_eval) return MSDF_ERR_ARG;
  return smp_launch(smp_resample_k, a, smp_lds_bytes(a), stream);
}
extern "C" int msdf_sampler_finish(const msdf_sampler_args_t* a, void* stream) {
  if (smp_check(a) || a->flags == nullptr || (a->extra_idx == nullptr && a->n_extra > 0) ||
      a->m_max < a->n_eval * a->max_rounds)
    return MSDF_ERR_ARG;
  const size_t lds = (size_t)SMP_WAVES * 2 * (a->n_final + a->n_extra + 2) * sizeof(float);
  return smp_launch(smp_finish_k, a, lds, stream);
}

extern "C" int msdf_sampler_error_bound(const float* z, const float* sdf, const float* dstar, const float* beta,
                                        int beta_stride, int N, int M, float* out, void* stream) {
  if (N < 0 || M < 2 || (beta_stride != 0 && beta_stride != 1)) return MSDF_ERR_ARG;
  if (N == 0) return MSDF_OK;
  const size_t lds = (size_t)SMP_WAVES * 5 * (M + 1) * sizeof(float);
  if (lds > 160 * 1024) return MSDF_ERR_UNSUPPORTED;
  if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)smp_error_bound_k,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return MSDF_ERR_LAUNCH;
  smp_error_bound_k<<<(N + SMP_WAVES - 1) / SMP_WAVES, 64 * SMP_WAVES, lds, (hipStream_t)stream>>>(
      z, sdf, dstar, beta, beta_stride, N, M, out);
  return msdf_check_launch();
}

static int dens_grid(int64_t n) { return (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048); }
extern "C" int msdf_laplace_density(const float* sdf, const float* beta, int beta_stride, int64_t n, int cols,
                                    float* out, void* stream) {
  if (n < 0 || cols < 1 || (beta_stride != 0 && beta_stride != 1)) return MSDF_ERR_ARG;
  if (n == 0) return MSDF_OK;
  laplace_density_k<<<dens_grid(n), 256, 0, (hipStream_t)stream>>>(sdf, beta, beta_stride, n, cols, out);
  return msdf_check_launch();
}
extern "C" int msdf_laplace_density_backward(const float* sdf, const float* beta, int beta_stride, int64_t n, int cols,
                                             const float* g, float* g_sdf, float* g_beta_elem, void* stream) {
  if (n < 0 || cols < 1 || (beta_stride != 0 && beta_stride != 1)) return MSDF_ERR_ARG;
  if (n == 0) return MSDF_OK;
  laplace_density_bwd_k<<<dens_grid(n), 256, 0, (hipStream_t)stream>>>(sdf, beta, beta_stride, n, cols, g, g_sdf,
                                                                     g_beta_elem);
  return msdf_check_launch();
}
