// Colour-network kernel bodies, templated on the matrix core (CoreF32 / CoreB16); see color_mlp.hip.
#pragma once
#include "mlp_core.h"

typedef msdf_color_fwd_args_t ColorFwdArgs;

__device__ __forceinline__ int wave_point(int& q) {
  const int lane = lane_id();
  q = lane >> 4;
  return blockIdx.x * MLP_PTS_PER_WG + (threadIdx.x >> 6) * MLP_PTS_PER_WAVE + (lane & 15);
}

// misc block in slot layout: idr: [x(3) | PE(v) | n(3)], nerf: [PE(v)]; then code tiles
__device__ __forceinline__ void color_misc_tiles(v4f (&m)[5], const msdf_plan_t& plan, const float* __restrict__ x,
                                                 const float* __restrict__ dirs, const float* __restrict__ nrm,
                                                 const float* __restrict__ code, const int ptc, const int ray,
                                                 const int q) {
  const int nv = 3 + 6 * plan.n_freqs;
  const float v0 = dirs[(size_t)ray * 3 + 0], v1 = dirs[(size_t)ray * 3 + 1], v2 = dirs[(size_t)ray * 3 + 2];
  float x0 = 0.f, x1 = 0.f, x2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
  if (plan.mode == 1) {
    x0 = x[(size_t)ptc * 3 + 0]; x1 = x[(size_t)ptc * 3 + 1]; x2 = x[(size_t)ptc * 3 + 2];
    n0 = nrm[(size_t)ptc * 3 + 0]; n1 = nrm[(size_t)ptc * 3 + 1]; n2 = nrm[(size_t)ptc * 3 + 2];
  }
  const int pe0 = (plan.mode == 1) ? 3 : 0;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * t + 4 * q + r;
      float val = 0.f;
      if (j < pe0) {
        val = (j == 0) ? x0 : (j == 1) ? x1 : x2;
      } else if (j < pe0 + nv) {
        float dv; int c;
        pe_slot(j - pe0, nv, v0, v1, v2, val, dv, c);
      } else if (plan.mode == 1 && j < pe0 + nv + 3) {
        const int k = j - pe0 - nv;
        val = (k == 0) ? n0 : (k == 1) ? n1 : n2;
      }
      m[t][r] = val;
    }
  }
  m[3] = m[4] = V4ZERO;
  if (plan.aux_tiles > 0 && code != nullptr) {
    const int aw = 16 * plan.aux_tiles;
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (t < plan.aux_tiles) m[3 + t] = *(const v4f*)(code + (size_t)ray * aw + 16 * t + 4 * q);
  }
}

__device__ __forceinline__ void load_bias_c(v4f (&acc)[MT], const float* __restrict__ b, const int ot, const int q) {
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = (t < ot) ? *(const v4f*)(b + 16 * t + 4 * q) : V4ZERO;
}


// ---------------------------------------------------------------------------
// Product hooks (protocol: mlp_core.h, NoHooks); lane pointers as in sdf_kernels.h.  kt == 0 disables a hook.
// ---------------------------------------------------------------------------
struct ColorPending {
  float* dst;
  v4f s0, s1;
  int t0, n;
  __device__ __forceinline__ void issue() {
    if (n > 0) *(v4f*)(dst + 16 * t0) = s0;
    if (n > 1) *(v4f*)(dst + 16 * t0 + 16) = s1;
    n = 0;
  }
};

// forward: h = relu(a) in place, saved to H (input of the next layer)
struct ReluSaveHooks {
  int kt;
  bool save;
  ColorPending st;
  __device__ __forceinline__ ReluSaveHooks(float* H, const int kt_, const bool save_) : kt(kt_), save(save_) {
    st.dst = H; st.n = 0; st.t0 = 0;
  }
  __device__ __forceinline__ void pre(const int, const int) { st.issue(); }
  __device__ __forceinline__ void post(const int o0, const int o1, const bool pair, v4f& a0, v4f& a1) {
    if (o0 < kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a0[r] = fmaxf(a0[r], 0.f);
      st.s0 = a0; st.t0 = o0; st.n = save ? 1 : 0;
      if (pair && o1 < kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a1[r] = fmaxf(a1[r], 0.f);
        st.s1 = a1; st.n = save ? 2 : 0;
      }
    }
  }
  __device__ __forceinline__ void drain() { st.issue(); }
};

// backward: a-bar = h-bar masked by h > 0 (h loaded from H) in place, saved to AB
struct ReluMaskHooks {
  const float* H;
  int kt;
  v4f h0, h1;
  ColorPending st;
  __device__ __forceinline__ ReluMaskHooks(const float* H_, float* AB, const int kt_) : H(H_), kt(kt_) {
    st.dst = AB; st.n = 0; st.t0 = 0;
  }
  __device__ __forceinline__ void pre(const int o0, const int o1) {
    st.issue();
    if (kt > 0) {
      h0 = *(const v4f*)(H + 16 * (o0 < kt ? o0 : kt - 1));
      h1 = *(const v4f*)(H + 16 * (o1 < kt ? o1 : kt - 1));
    }
  }
  __device__ __forceinline__ void post(const int o0, const int o1, const bool pair, v4f& a0, v4f& a1) {
    if (o0 < kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a0[r] = (h0[r] > 0.f) ? a0[r] : 0.f;
      st.s0 = a0; st.t0 = o0; st.n = 1;
      if (pair && o1 < kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a1[r] = (h1[r] > 0.f) ? a1[r] : 0.f;
        st.s1 = a1; st.n = 2;
      }
    }
  }
  __device__ __forceinline__ void drain() { st.issue(); }
};

template <class Core>
__device__ __forceinline__ void color_forward_body(const msdf_plan_t& plan, const ColorFwdArgs& a, void* lds) {
  typedef typename Core::wvec wvec;
  int q;
  const int pt = wave_point(q);
  const bool valid = pt < a.P;
  const int ptc = valid ? pt : a.P - 1;
  const int ray = ptc / a.spr;
  const size_t Pp = (size_t)a.P_pad;
  const int misc_tiles = plan.e_tiles + plan.aux_tiles;
  const int nu = plan.n_layers;             // pack units
  v4f in[MT], acc[MT];
  // ---- first layer, feature part
  const msdf_layer_t U0 = plan.layer[0];
#pragma unroll
  for (int t = 0; t < MT; ++t)
    in[t] = (t < U0.kt) ? *(const v4f*)(a.feat + (size_t)ptc * (16 * U0.kt) + 16 * t + 4 * q) : V4ZERO;
  if constexpr (Core::BIAS_IN_HOOKS) {
    Core::gemm_bias(U0.ktp, acc, in, U0.ot, (const wvec*)a.wpack + U0.wf_off, lds, NoHooks(), a.bpack + U0.bias_off + 4 * q);
  } else {
    load_bias_c(acc, a.bpack + U0.bias_off, U0.ot, q);
    Core::gemm(U0.ktp, acc, in, U0.ot, (const wvec*)a.wpack + U0.wf_off, lds, NoHooks());
  }
  // ---- first layer, misc part (same accumulators)
  {
    v4f m[5];
    color_misc_tiles(m, plan, a.x, a.dirs, a.nrm, a.code, ptc, ray, q);
    place_tiles(in, 0, m, misc_tiles);
    if (a.save) {
#pragma unroll
      for (int t = 0; t < 5; ++t)
        if (t < misc_tiles) *(v4f*)(a.MISC + (size_t)pt * (16 * misc_tiles) + 16 * t + 4 * q) = m[t];
    }
  }
  // every product but the last hands relu(output) to the next layer through its hooks (saved to H on the way)
  auto next_hooks = [&](const int u) {     // hooks of the product that feeds unit u (u >= nu: none)
    const msdf_layer_t Ln = plan.layer[u < nu ? u : 0];
    return ReluSaveHooks(a.H + (size_t)Ln.hpre * Pp + (size_t)pt * (16 * Ln.kt) + 4 * q, u < nu ? Ln.kt : 0,
                         a.save != 0);
  };
  const msdf_layer_t U1 = plan.layer[1];
  Core::gemm(U1.ktp, acc, in, U1.ot, (const wvec*)a.wpack + U1.wf_off, lds, next_hooks(2));
  // ---- hidden layers
#ifndef MSDF_DOT_FWD
#define MSDF_DOT_FWD 1
#endif
  const bool dot_out = MSDF_DOT_FWD && nu >= 3;             // the output layer behind a hidden one: three dot products, not a product
  for (int u = 2; u < (dot_out ? nu - 1 : nu); ++u) {
    const msdf_layer_t L = plan.layer[u];
#pragma unroll
    for (int t = 0; t < MT; ++t) in[t] = (t < L.kt) ? acc[t] : V4ZERO;
    if constexpr (Core::BIAS_IN_HOOKS) {
      Core::gemm_bias(L.ktp, acc, in, L.ot, (const wvec*)a.wpack + L.wf_off, lds, next_hooks(u + 1),
                      a.bpack + L.bias_off + 4 * q);
    } else {
      load_bias_c(acc, a.bpack + L.bias_off, L.ot, q);
      Core::gemm(L.ktp, acc, in, L.ot, (const wvec*)a.wpack + L.wf_off, lds, next_hooks(u + 1));
    }
  }
  // ---- output layer: rgb_c = w_c . relu(h) + b_c over the register-resident activation (a 16-row matrix tile
  // with 13 rows of zeros would be one dependent chain of 64 matrix instructions behind a weight chunk of its own)
  float o0 = acc[0][0], o1 = acc[0][1], o2 = acc[0][2];
  if (dot_out) {
    const msdf_layer_t LO = plan.layer[nu - 1];
    const float* w = a.bpack + plan.wsdf_off + 4 * q;
    const int ld = 16 * LO.kt;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      if (t < LO.kt) {
        const v4f w0 = *(const v4f*)(w + 16 * t), w1 = *(const v4f*)(w + ld + 16 * t), w2 = *(const v4f*)(w + 2 * ld + 16 * t);
        const v4f h = acc[t];
        p0 += w0.x * h.x + w0.y * h.y + w0.z * h.z + w0.w * h.w;
        p1 += w1.x * h.x + w1.y * h.y + w1.z * h.z + w1.w * h.w;
        p2 += w2.x * h.x + w2.y * h.y + w2.z * h.z + w2.w * h.w;
      }
    }
    o0 = sum_over_quarters(p0) + a.bpack[LO.bias_off + 0];
    o1 = sum_over_quarters(p1) + a.bpack[LO.bias_off + 1];
    o2 = sum_over_quarters(p2) + a.bpack[LO.bias_off + 2];
  }
  if (valid && q == 0) {
    const float o[3] = {o0, o1, o2};
#pragma unroll
    for (int r = 0; r < 3; ++r)
      a.rgb[(size_t)pt * 3 + r] = (plan.out_act == 1) ? fmaxf(o[r], 0.f) : 1.0f / (1.0f + fast_exp(-o[r]));
  }
}

typedef msdf_color_bwd_args_t ColorBwdArgs;

template <class Core>
__device__ __forceinline__ void color_backward_body(const msdf_plan_t& plan, const ColorBwdArgs& a, void* lds) {
  typedef typename Core::wvec wvec;
  int q;
  const int pt = wave_point(q);
  const bool valid = pt < a.P;
  const size_t Pp = (size_t)a.P_pad;
  const int nu = plan.n_layers;
  const int misc_tiles = plan.e_tiles + plan.aux_tiles;
  v4f in[MT], acc[MT];
  zero_tiles(in);
  // ---- a-bar of the output layer
  float ab0 = 0.f, ab1 = 0.f, ab2 = 0.f;     // a-bar of the three colour rows, in every quarter lane of the point
  {
    v4f ab = V4ZERO;
    if (valid) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float y = a.rgb[(size_t)pt * 3 + r];
        const float g = a.g_rgb[(size_t)pt * 3 + r];
        ab[r] = (plan.out_act == 1) ? ((y > 0.f) ? g : 0.f) : g * y * (1.0f - y);
      }
    }
    ab0 = ab[0]; ab1 = ab[1]; ab2 = ab[2];
    if (q != 0) ab = V4ZERO;                  // slots 0..2 sit in tile 0, quarter 0
    in[0] = ab;
    const msdf_layer_t LL = plan.layer[nu - 1];
    float* ABl = a.AB + (size_t)LL.abpre * Pp + (size_t)pt * (16 * LL.ot) + 4 * q;
    *(v4f*)ABl = ab;
  }
  // ---- output layer behind a hidden one: h-bar = sum_c a-bar_c w_c as an outer product over the activation's slots
  // (the 3-row matrix product was 8 weight chunks of 8 matrix instructions each), masked and stored like a hook does
#ifndef MSDF_DOT_BWD
#define MSDF_DOT_BWD 1
#endif
  const bool dot_out = MSDF_DOT_BWD && nu >= 3;
  if (dot_out) {
    const msdf_layer_t L = plan.layer[nu - 1];
    const float a0 = ab0, a1 = ab1, a2 = ab2;
    const float* w = a.bpack + plan.wsdf_off + 4 * q;
    const int ld = 16 * L.kt;
    const float* Hl = a.H + (size_t)L.hpre * Pp + (size_t)pt * (16 * L.kt) + 4 * q;
    const msdf_layer_t Lp = plan.layer[(nu - 2 == 1) ? 0 : nu - 2];
    float* ABp = a.AB + (size_t)Lp.abpre * Pp + (size_t)pt * (16 * Lp.ot) + 4 * q;
    const int ktl = L.kt - 1;
    v4f hh[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) hh[t] = *(const v4f*)(Hl + 16 * (t < ktl ? t : ktl));     // every load before a store
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      v4f v = V4ZERO;
      if (t < L.kt) {
        const v4f w0 = *(const v4f*)(w + 16 * t), w1 = *(const v4f*)(w + ld + 16 * t), w2 = *(const v4f*)(w + 2 * ld + 16 * t);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = a0 * w0[r] + a1 * w1[r] + a2 * w2[r];
          v[r] = (hh[t][r] > 0.f) ? g : 0.f;
        }
      }
      in[t] = v;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
      if (t < L.kt) *(v4f*)(ABp + 16 * t) = in[t];
  }
  // ---- hidden layers, last to first
  for (int u = (dot_out ? nu - 2 : nu - 1); u >= 2; --u) {
    const msdf_layer_t L = plan.layer[u];
    zero_tiles(acc);
    // the product yields h-bar of this layer's input = output of unit u-1 (u-1 == 1 means the first layer);
    // its hooks mask it with the saved activation and store it as a-bar of that unit
    const msdf_layer_t Lp = plan.layer[(u - 1 == 1) ? 0 : u - 1];
    Core::gemm(L.otp, acc, in, L.kt, (const wvec*)a.wpack + L.wb_off, lds,
               ReluMaskHooks(a.H + (size_t)L.hpre * Pp + (size_t)pt * (16 * L.kt) + 4 * q,
                             a.AB + (size_t)Lp.abpre * Pp + (size_t)pt * (16 * Lp.ot) + 4 * q, L.kt));
#pragma unroll
    for (int t = 0; t < MT; ++t) in[t] = (t < L.kt) ? acc[t] : V4ZERO;
  }
  // ---- first layer: gradient of the feature tiles and of the misc block
  const msdf_layer_t U0 = plan.layer[0];
  zero_tiles(acc);
  Core::gemm(U0.otp, acc, in, U0.kt, (const wvec*)a.wpack + U0.wb_off, lds, NoHooks());
  if (valid) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
      if (t < U0.kt) *(v4f*)(a.g_feat + (size_t)pt * (16 * U0.kt) + 16 * t + 4 * q) = acc[t];
  }
  const msdf_layer_t U1 = plan.layer[1];
  zero_tiles(acc);
  Core::gemm(U1.otp, acc, in, U1.kt, (const wvec*)a.wpack + U1.wb_off, lds, NoHooks());
  if (valid) {
#pragma unroll
    for (int t = 0; t < 5; ++t)
      if (t < misc_tiles) *(v4f*)(a.g_misc + (size_t)pt * (16 * misc_tiles) + 16 * t + 4 * q) = acc[t];
    if (a.g_nrm != nullptr && plan.mode == 1) {
      // misc slots of mode idr: [x (3) | PE(view) (3 + 6 n_freqs) | normal (3)] -- the normal's gradient once more
      // as a dense [P,3] row, so the caller does not have to copy the strided columns out
      const int n0 = 3 + 3 + 6 * plan.n_freqs;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * t + 4 * q + r - n0;
          if (k >= 0 && k < 3) a.g_nrm[(size_t)pt * 3 + k] = acc[t][r];
        }
    }
  }
}

