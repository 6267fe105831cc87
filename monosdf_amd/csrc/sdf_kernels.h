// SDF-network kernel bodies, templated on the matrix core (CoreF32: fp32 MFMA, mlp_core.h;
// CoreB16: bf16x3 split on the bf16 MFMA, mlp_core_b16.h).  Included by sdf_mlp.hip and sdf_mlp_b16.hip,
// which instantiate the __global__ wrappers.  Math: DESIGN.md "SDF network kernels".
#pragma once
#include "mlp_core.h"

// ---------------------------------------------------------------------------
// shared pieces
// ---------------------------------------------------------------------------
struct PointCtx {
  int pt;        // global point index of this lane's point (may be >= P on the padded tail)
  int ptc;       // clamped to P-1 for loads
  bool valid;
  int q;         // quarter
  float x0, x1, x2;
};

__device__ __forceinline__ PointCtx load_point(const float* __restrict__ x, const int P) {
  PointCtx c;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  c.pt = blockIdx.x * MLP_PTS_PER_WG + wave * MLP_PTS_PER_WAVE + (lane & 15);
  c.valid = c.pt < P;
  c.ptc = c.valid ? c.pt : (P - 1);
  c.q = lane >> 4;
  c.x0 = x[(size_t)c.ptc * 3 + 0];
  c.x1 = x[(size_t)c.ptc * 3 + 1];
  c.x2 = x[(size_t)c.ptc * 3 + 2];
  return c;
}

// Extra input features of a point (hash-grid features) and everything shaped like them (d sdf / d features, their
// gradients): either rows [P, 16 * aux_tiles] (C == 0) or the hash encoder's own level-major tensor [L][P][C] with
// L C = LC valid columns (C = 2: what every configuration of the reference uses; other channel counts take the rows),
// column l C + c of a point = level l, channel c.  A lane owns the columns 16 t + 4 q + r of its point: two levels per
// tile, two 8-byte pieces, 128 contiguous bytes per level over the 16 points of a wave.  Reading and writing the level-major tensors
// here is what removes the three transposes around the encoder and the scatter from the training step.
struct AuxView {
  int C, LC, P;
};
typedef float v2f_aux __attribute__((ext_vector_type(2)));

// LM = false compiles the level-major form out (the bf16 cores: their kernels sit at the register limit and the extra
// address arithmetic cost them 30-100 bytes of scratch per lane; they take rows and the LDS-tiled transposes)
template <bool LM>
__device__ __forceinline__ v4f aux_load_tile(const float* __restrict__ base, const AuxView av, const int aux_tiles,
                                             const int t, const int pt, const int q) {
  const int s0 = 16 * t + 4 * q;
  if (!LM || av.C == 0) return *(const v4f*)(base + (size_t)pt * (16 * aux_tiles) + s0);
  // C == 2: levels s0 / 2 and s0 / 2 + 1 (LC is even: both inside or both outside)
  const int l = (s0 < av.LC) ? (s0 >> 1) : 0;
  const v2f_aux a = *(const v2f_aux*)(base + ((size_t)l * av.P + pt) * 2);
  const v2f_aux b = *(const v2f_aux*)(base + ((size_t)(l + (s0 + 2 < av.LC ? 1 : 0)) * av.P + pt) * 2);
  v4f v = (v4f){a.x, a.y, b.x, b.y};
  if (s0 >= av.LC) v = V4ZERO;
  else if (s0 + 2 >= av.LC) { v.z = 0.f; v.w = 0.f; }
  return v;
}

template <bool LM>
__device__ __forceinline__ void aux_store_tile(float* __restrict__ base, const AuxView av, const int aux_tiles,
                                               const int t, const int pt, const int q, const v4f v) {
  const int s0 = 16 * t + 4 * q;
  if (!LM || av.C == 0) { *(v4f*)(base + (size_t)pt * (16 * aux_tiles) + s0) = v; return; }
  if (s0 >= av.LC) return;
  const int l = s0 >> 1;
  *(v2f_aux*)(base + ((size_t)l * av.P + pt) * 2) = (v2f_aux){v.x, v.y};
  if (s0 + 2 < av.LC) *(v2f_aux*)(base + ((size_t)(l + 1) * av.P + pt) * 2) = (v2f_aux){v.z, v.w};
}

// The hash encoder's Jacobian d feature / d x01, level-major [L][P][3][2] (two features per level), applied inside the
// SDF kernels instead of by two kernels of their own (hg_node_input_gradient / hg_node_second_grad, csrc/hashgrid.hip --
// the reference's kernel_input_backward and kernel_grid_second_backward_grad, hashencoder.cu:346-428): a lane owns four
// levels of its point (two per aux tile) and reads their 6 floats as three 8-byte pieces.
// grid part of d sdf / d x:  m_d = sum over the lane's (level, channel) of r[l,c] * J[l,d,c]   (summed over the quarters by the caller)
__device__ __forceinline__ void aux_jacobian_transpose(const float* __restrict__ dy_dx, const AuxView av,
                                                       const int aux_tiles, const v4f (&r)[2], const int pt, const int q,
                                                       float& m0, float& m1, float& m2) {
  m0 = m1 = m2 = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int s0 = 16 * t + 4 * q + 2 * h;
      if (t < aux_tiles && s0 < av.LC) {
        const float* d = dy_dx + ((size_t)(s0 >> 1) * av.P + pt) * 6;
        const v2f_aux j0 = *(const v2f_aux*)d, j1 = *(const v2f_aux*)(d + 2), j2 = *(const v2f_aux*)(d + 4);
        const float g0 = r[t][2 * h], g1 = r[t][2 * h + 1];
        m0 += g0 * j0.x + g1 * j0.y;
        m1 += g0 * j1.x + g1 * j1.y;
        m2 += g0 * j2.x + g1 * j2.y;
      }
    }
  }
}

// gradient arriving at d sdf / d features from the grid part of d sdf / d x:  out[l,c] = sum_d g_d * J[l,d,c], each product
// and sum rounded separately, in the order of hg_node_second_grad_kernel (bit-identical to that kernel)
__device__ __forceinline__ v4f aux_jacobian_tile(const float* __restrict__ dy_dx, const AuxView av, const int t,
                                                 const int pt, const int q, const float g0, const float g1, const float g2) {
  v4f v = V4ZERO;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int s0 = 16 * t + 4 * q + 2 * h;
    if (s0 < av.LC) {
      const float* d = dy_dx + ((size_t)(s0 >> 1) * av.P + pt) * 6;
      const v2f_aux j0 = *(const v2f_aux*)d, j1 = *(const v2f_aux*)(d + 2), j2 = *(const v2f_aux*)(d + 4);
      v[2 * h] = __fadd_rn(__fadd_rn(__fmul_rn(g0, j0.x), __fmul_rn(g1, j1.x)), __fmul_rn(g2, j2.x));
      v[2 * h + 1] = __fadd_rn(__fadd_rn(__fmul_rn(g0, j0.y), __fmul_rn(g1, j1.y)), __fmul_rn(g2, j2.y));
    }
  }
  return v;
}

// network input tiles: PE tiles then aux tiles (hash-grid features)
template <bool LM>
__device__ __forceinline__ void load_input_tiles(v4f (&in0)[5], const msdf_plan_t& plan,
                                                 const float* __restrict__ aux, const AuxView av, const PointCtx& c) {
  pe_values(in0, c.x0, c.x1, c.x2, plan.n_freqs);
  in0[3] = in0[4] = V4ZERO;
  if (plan.aux_tiles > 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (t < plan.aux_tiles) in0[3 + t] = aux_load_tile<LM>(aux, av, plan.aux_tiles, t, c.ptc, c.q);
  }
}

__device__ __forceinline__ void load_bias(v4f (&acc)[MT], const float* __restrict__ b, const int ot, const int q) {
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = (t < ot) ? *(const v4f*)(b + 16 * t + 4 * q) : V4ZERO;
}

// sdf row of the output layer as a dot product over the last hidden activation
__device__ __forceinline__ float sdf_row_dot(const v4f (&in)[MT], const msdf_plan_t& plan,
                                             const float* __restrict__ bpack, const int q) {
  const msdf_layer_t LL = plan.layer[plan.n_layers - 1];
  float part = 0.f;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    if (t < LL.kt) {
      const v4f w = *(const v4f*)(bpack + plan.wsdf_off + 16 * t + 4 * q);
      part += w.x * in[t].x + w.y * in[t].y + w.z * in[t].z + w.w * in[t].w;
    }
  }
  return sum_over_quarters(part) + bpack[LL.bias_off + plan.sdf_slot];
}

// ---------------------------------------------------------------------------
// Product hooks (protocol: mlp_core.h, NoHooks).  Pointers are lane pointers (row of the lane's point + 4 q);
// tile t of the row sits at p + 16 t.  Results to be stored wait in registers until the next pre() / drain().
// ---------------------------------------------------------------------------
struct PendingStores {
  float* dst;
  v4f s0, s1;
  int t0, n;
  __device__ __forceinline__ void issue() {
    if (n > 0) *(v4f*)(dst + 16 * t0) = s0;
    if (n > 1) *(v4f*)(dst + 16 * t0 + 16) = s1;
    n = 0;
  }
};

// h = softplus(a) in place (forward kernel: nothing is saved)
template <class Core>
struct SoftplusHooks {
  __device__ __forceinline__ void pre(const int, const int) {}
  __device__ __forceinline__ void post(const int, const int, const bool pair, v4f& a0, v4f& a1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a0[r] = Core::softplus(a0[r]);
    if (pair) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a1[r] = Core::softplus(a1[r]);
    }
  }
  __device__ __forceinline__ void drain() {}
};

// h = softplus(a) in place and saved to H (forward chain of the fwd+grad kernel)
template <class Core>
struct SoftplusSaveHooks {
  PendingStores st;
  __device__ __forceinline__ SoftplusSaveHooks(float* H) { st.dst = H; st.n = 0; st.t0 = 0; }
  __device__ __forceinline__ void pre(const int, const int) { st.issue(); }
  __device__ __forceinline__ void post(const int o0, const int, const bool pair, v4f& a0, v4f& a1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a0[r] = Core::softplus(a0[r]);
    st.s0 = a0; st.t0 = o0; st.n = 1;
    if (pair) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a1[r] = Core::softplus(a1[r]);
      st.s1 = a1; st.n = 2;
    }
  }
  __device__ __forceinline__ void drain() { st.issue(); }
};

// gradient sweep of the fwd+grad kernel: g -> p = s g with s = 1 - exp(-100 h), h loaded from H; p saved to PM.
// Tiles >= ot (the input block behind a skip layer's hidden tiles; ot == 0: no epilogue at all) are left alone.
struct GradSweepHooks {
  const float* H;
  int ot;
  bool save;
  v4f h0, h1;
  PendingStores st;
  __device__ __forceinline__ GradSweepHooks(const float* H_, float* PM, const int ot_, const bool save_)
      : H(H_), ot(ot_), save(save_) { st.dst = PM; st.n = 0; st.t0 = 0; }
  __device__ __forceinline__ void pre(const int o0, const int o1) {
    st.issue();
    if (ot > 0) {
      h0 = *(const v4f*)(H + 16 * (o0 < ot ? o0 : ot - 1));
      h1 = *(const v4f*)(H + 16 * (o1 < ot ? o1 : ot - 1));
    }
  }
  __device__ __forceinline__ void post(const int o0, const int o1, const bool pair, v4f& a0, v4f& a1) {
    if (o0 < ot) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a0[r] = (1.0f - one_minus_sigmoid_from_h(h0[r])) * a0[r];
      st.s0 = a0; st.t0 = o0; st.n = save ? 1 : 0;
      if (pair && o1 < ot) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a1[r] = (1.0f - one_minus_sigmoid_from_h(h1[r])) * a1[r];
        st.s1 = a1; st.n = save ? 2 : 0;
      }
    }
  }
  __device__ __forceinline__ void drain() { st.issue(); }
};

// sweep up of the double backward: p-bar -> q-bar = s p-bar in place.  The second-order term t = 100 (1 - s) p p-bar
// is NOT stored: the sweep down forms it again from q-bar (which the next layer's input store keeps anyway) as
// 100 (1 - s) p q-bar / s -- one store of a 1 KB row per layer and point less, and no PM read here.
struct SweepUpHooks {
  const float* H;
  int ot;
  v4f h0, h1;
  __device__ __forceinline__ SweepUpHooks(const float* H_, const int ot_) : H(H_), ot(ot_) {}
  __device__ __forceinline__ void pre(const int o0, const int o1) {
    const int c0 = o0 < ot ? o0 : ot - 1, c1 = o1 < ot ? o1 : ot - 1;
    h0 = *(const v4f*)(H + 16 * c0);
    h1 = *(const v4f*)(H + 16 * c1);
  }
  __device__ __forceinline__ void post(const int, const int, const bool pair, v4f& a0, v4f& a1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a0[r] = (1.0f - one_minus_sigmoid_from_h(h0[r])) * a0[r];
    if (pair) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a1[r] = (1.0f - one_minus_sigmoid_from_h(h1[r])) * a1[r];
    }
  }
  __device__ __forceinline__ void drain() {}
};

// sweep down: h-bar -> a-bar = h-bar s + t in place, saved to AB, with t = 100 (1 - s) p p-bar formed from the saved
// p (PM) and q-bar = s p-bar (the input rows of the next layer: QB, or QLAST above the last hidden layer) as
// 100 (1 - s) p q-bar / s; s == 0 means q-bar == 0 and p == 0: t = 0.  Tiles >= ot are left alone (ot == 0: none).
struct SweepDownHooks {
  const float* H;
  const float* PM;
  const float* Q;      // q-bar rows of the layer above (same tile order as this layer's outputs)
  int ot;
  v4f h0, h1, p0, p1, q0, q1;
  PendingStores st;
  __device__ __forceinline__ SweepDownHooks(const float* H_, const float* PM_, const float* Q_, float* AB, const int ot_)
      : H(H_), PM(PM_), Q(Q_), ot(ot_) { st.dst = AB; st.n = 0; st.t0 = 0; }
  __device__ __forceinline__ void pre(const int o0, const int o1) {
    st.issue();
    if (ot > 0) {
      const int c0 = o0 < ot ? o0 : ot - 1, c1 = o1 < ot ? o1 : ot - 1;
      h0 = *(const v4f*)(H + 16 * c0);
      h1 = *(const v4f*)(H + 16 * c1);
      p0 = *(const v4f*)(PM + 16 * c0);
      p1 = *(const v4f*)(PM + 16 * c1);
      q0 = *(const v4f*)(Q + 16 * c0);
      q1 = *(const v4f*)(Q + 16 * c1);
    }
  }
  static __device__ __forceinline__ void tile(const v4f h, const v4f p, const v4f q, v4f& a) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float u = one_minus_sigmoid_from_h(h[r]);
      const float sg = 1.0f - u;
      const float pb = (sg > 0.f) ? q[r] * __builtin_amdgcn_rcpf(sg) : 0.f;      // p-bar
      a[r] = a[r] * sg + 100.0f * u * p[r] * pb;
    }
  }
  __device__ __forceinline__ void post(const int o0, const int o1, const bool pair, v4f& a0, v4f& a1) {
    if (o0 < ot) {
      tile(h0, p0, q0, a0);
      st.s0 = a0; st.t0 = o0; st.n = 1;
      if (pair && o1 < ot) {
        tile(h1, p1, q1, a1);
        st.s1 = a1; st.n = 2;
      }
    }
  }
  __device__ __forceinline__ void drain() { st.issue(); }
};

// ---------------------------------------------------------------------------
// F: forward only, sdf only (get_sdf_vals; reference network.py:131-137 / 307-309)
// ---------------------------------------------------------------------------
template <class Core>
__device__ __forceinline__ void sdf_forward_body(const msdf_plan_t& plan, const typename Core::wvec* __restrict__ wpack,
                                                 const float* __restrict__ bpack, const float* __restrict__ x,
                                                 const float* __restrict__ aux, const AuxView av, const int P,
                                                 const float clamp_radius, const float sphere_scale,
                                                 float* __restrict__ sdf_out, void* lds) {
  const PointCtx c = load_point(x, P);
  v4f in[MT], acc[MT];
  const int in0_tiles = plan.e_tiles + plan.aux_tiles;
  {
    v4f in0[5];
    load_input_tiles<Core::AUX_LEVEL_MAJOR>(in0, plan, aux, av, c);
    place_tiles(in, 0, in0, in0_tiles);
  }
  const int nl = plan.n_layers;
  for (int l = 0; l < nl - 1; ++l) {
    const msdf_layer_t L = plan.layer[l];
    // skip layer: the network input is appended after the hidden tiles (1/sqrt2 folded into the pack)
    if (L.skip_tile >= 0) {
      v4f in0[5];
      load_input_tiles<Core::AUX_LEVEL_MAJOR>(in0, plan, aux, av, c);
      place_tiles(in, L.skip_tile, in0, in0_tiles);
    }
    if constexpr (Core::BIAS_IN_HOOKS) {
      Core::gemm_bias(L.ktp, acc, in, L.ot, wpack + L.wf_off, lds, SoftplusHooks<Core>(), bpack + L.bias_off + 4 * c.q);
    } else {
      load_bias(acc, bpack + L.bias_off, L.ot, c.q);
      Core::gemm(L.ktp, acc, in, L.ot, wpack + L.wf_off, lds, SoftplusHooks<Core>());
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) in[t] = (t < L.ot) ? acc[t] : V4ZERO;
  }
  float sdf = sdf_row_dot(in, plan, bpack, c.q);
  if (clamp_radius > 0.f) {
    const float nx = sqrtf(c.x0 * c.x0 + c.x1 * c.x1 + c.x2 * c.x2);
    sdf = fminf(sdf, sphere_scale * (clamp_radius - nx));
  }
  if (c.valid && c.q == 0) sdf_out[c.pt] = sdf;
}

// ---------------------------------------------------------------------------
// FG: forward + gradient sweep.  Saves H (post-activations) and PM (p_l = s_l * dsdf/dh_{l+1})
// into the workspace for the backward, and the input block IN0 = [PE | aux] for the weight-gradient GEMM.
// ---------------------------------------------------------------------------
typedef msdf_fg_args_t FgArgs;

template <class Core>
__device__ __forceinline__ void sdf_fwd_grad_body(const msdf_plan_t& plan, const FgArgs& a, void* lds) {
  typedef typename Core::wvec wvec;
  const PointCtx c = load_point(a.x, a.P);
  const AuxView av = {a.aux_C, a.aux_LC, a.P};
  v4f in[MT], acc[MT];
  const int nl = plan.n_layers;
  const int in0_tiles = plan.e_tiles + plan.aux_tiles;
  const size_t Pp = (size_t)a.P_pad;
  {
    v4f in0[5];
    load_input_tiles<Core::AUX_LEVEL_MAJOR>(in0, plan, a.aux, av, c);
    place_tiles(in, 0, in0, in0_tiles);
    if (a.save) {
#pragma unroll
      for (int t = 0; t < 5; ++t)
        if (t < in0_tiles) *(v4f*)(a.IN0 + (size_t)c.pt * (16 * in0_tiles) + 16 * t + 4 * c.q) = in0[t];
    }
  }
  // workgroup-uniform on purpose: the gemm below contains barriers and cooperative weight staging
  const bool want_feat = (int)(blockIdx.x * MLP_PTS_PER_WG) < a.n_feat;

  // ---------------- forward chain ----------------
  for (int l = 0; l < nl - 1; ++l) {
    const msdf_layer_t L = plan.layer[l];
    if (L.skip_tile >= 0) {
      v4f in0[5];
      load_input_tiles<Core::AUX_LEVEL_MAJOR>(in0, plan, a.aux, av, c);
      place_tiles(in, L.skip_tile, in0, in0_tiles);
    }
    float* Hl = a.H + (size_t)L.hpre * Pp + (size_t)c.pt * (16 * L.ot) + 4 * c.q;
    if constexpr (Core::BIAS_IN_HOOKS) {
      Core::gemm_bias(L.ktp, acc, in, L.ot, (const wvec*)a.wpack + L.wf_off, lds, SoftplusSaveHooks<Core>(Hl),
                      a.bpack + L.bias_off + 4 * c.q);
    } else {
      load_bias(acc, a.bpack + L.bias_off, L.ot, c.q);
      Core::gemm(L.ktp, acc, in, L.ot, (const wvec*)a.wpack + L.wf_off, lds, SoftplusSaveHooks<Core>(Hl));
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) in[t] = (t < L.ot) ? acc[t] : V4ZERO;
  }
  // ---------------- output layer ----------------
  const msdf_layer_t LL = plan.layer[nl - 1];
  float sdf;
  if (want_feat) {
    if constexpr (Core::BIAS_IN_HOOKS) {
      Core::gemm_bias(LL.ktp, acc, in, LL.ot, (const wvec*)a.wpack + LL.wf_off, lds, NoHooks(),
                      a.bpack + LL.bias_off + 4 * c.q);
    } else {
      load_bias(acc, a.bpack + LL.bias_off, LL.ot, c.q);
      Core::gemm(LL.ktp, acc, in, LL.ot, (const wvec*)a.wpack + LL.wf_off, lds, NoHooks());
    }
    if (c.valid && c.pt < a.n_feat) {
      float* f = a.feat + (size_t)c.pt * (16 * plan.feat_tiles) + 4 * c.q;
#pragma unroll
      for (int t = 0; t < MT; ++t)
        if (t < plan.feat_tiles) *(v4f*)(f + 16 * t) = acc[t];
    }
    // sdf sits in slot sdf_slot = 16 * feat_tiles (quarter 0, component 0 of that tile)
    float sv = 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t)
      if (t == plan.feat_tiles) sv = acc[t].x;
    sdf = __shfl(sv, lane_id() & 15, 64);
  } else {
    sdf = sdf_row_dot(in, plan, a.bpack, c.q);
  }

  // ---------------- gradient sweep (reverse mode through the sdf row) ----------------
  // acc carries g = d sdf / d h_{l+1} between iterations
#pragma unroll
  for (int t = 0; t < MT; ++t)
    acc[t] = (t < LL.kt) ? *(const v4f*)(a.bpack + plan.wsdf_off + 16 * t + 4 * c.q) : V4ZERO;
  // d sdf / d input arrives twice (behind the skip layer's hidden tiles and from layer 0); each arrival goes
  // through the PE Jacobian at once -- 3 floats + the aux tiles stay live across the products, not 5 tiles
  float n0 = 0.f, n1 = 0.f, n2 = 0.f;
  v4f r_aux[2] = {V4ZERO, V4ZERO};
  auto take_input_grad = [&](const int base) {
    v4f r[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) r[t] = V4ZERO;
    gather_tiles(r, acc, base, in0_tiles);
    float m0, m1, m2;
    pe_jacobian_transpose(r, c.x0, c.x1, c.x2, plan.n_freqs, m0, m1, m2);
    n0 += m0; n1 += m1; n2 += m2;
    r_aux[0] += r[3];
    r_aux[1] += r[4];
  };
  auto row = [&](const msdf_layer_t& L) { return (size_t)L.hpre * Pp + (size_t)c.pt * (16 * L.ot) + 4 * c.q; };
  if (nl >= 2) {
    // p of the last hidden layer: no product precedes it (g is the sdf row itself) -- every H load before
    // the first PM store, tile index clamped instead of guarded, so that no branch separates the loads
    const msdf_layer_t L = plan.layer[nl - 2];
    const float* Hl = a.H + row(L);
    float* Pl = a.PM + row(L);
    const int otl = L.ot - 1;
    v4f hh[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) hh[t] = *(const v4f*)(Hl + 16 * (t < otl ? t : otl));
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      if (t == 0 || t < L.ot) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = (1.0f - one_minus_sigmoid_from_h(hh[t][r])) * acc[t][r];
        if (a.save) *(v4f*)(Pl + 16 * t) = acc[t];
      }
    }
  }
  for (int l = nl - 2; l >= 0; --l) {
    const msdf_layer_t L = plan.layer[l];
    // acc = p_l on the tiles < L.ot; the product below yields g = d sdf / d h_l and its hooks turn that into
    // p_{l-1} (tiles behind a skip layer's hidden part are d sdf / d input: left alone)
#pragma unroll
    for (int t = 0; t < MT; ++t) in[t] = (t < L.ot) ? acc[t] : V4ZERO;
    zero_tiles(acc);
    const msdf_layer_t Lp = plan.layer[l > 0 ? l - 1 : 0];
    Core::gemm(L.otp, acc, in, L.kt, (const wvec*)a.wpack + L.wb_off, lds,
               GradSweepHooks(a.H + row(Lp), a.PM + row(Lp), l > 0 ? Lp.ot : 0, a.save != 0));
    if (l == 0) take_input_grad(0);
    else if (L.skip_tile >= 0) take_input_grad(L.skip_tile);
  }
  // d sdf / d x through the hash grid (its chain-rule factor a.aux_dx_scale in), added as the module's tensor
  // expression adds it: the rounded product, then the sum
  if (Core::AUX_LEVEL_MAJOR && a.dy_dx != nullptr && plan.aux_tiles > 0) {
    float m0, m1, m2;
    aux_jacobian_transpose(a.dy_dx, av, plan.aux_tiles, r_aux, c.ptc, c.q, m0, m1, m2);
    n0 = __fadd_rn(n0, __fmul_rn(sum_over_quarters(m0), a.aux_dx_scale));
    n1 = __fadd_rn(n1, __fmul_rn(sum_over_quarters(m1), a.aux_dx_scale));
    n2 = __fadd_rn(n2, __fmul_rn(sum_over_quarters(m2), a.aux_dx_scale));
  }
  // ---------------- clamp, stores ----------------
  bool is_clamped = false;
  if (a.clamp_radius > 0.f && c.pt < a.n_clamp) {
    const float nx = sqrtf(c.x0 * c.x0 + c.x1 * c.x1 + c.x2 * c.x2);
    const float sph = a.sphere_scale * (a.clamp_radius - nx);
    if (sph < sdf) {
      is_clamped = true;
      sdf = sph;
      const float inv = -a.sphere_scale / nx;
      n0 = c.x0 * inv; n1 = c.x1 * inv; n2 = c.x2 * inv;
    }
  }
  if (c.valid) {
    if (c.q == 0) {
      a.sdf[c.pt] = sdf;
      a.nrm[(size_t)c.pt * 3 + 0] = n0;
      a.nrm[(size_t)c.pt * 3 + 1] = n1;
      a.nrm[(size_t)c.pt * 3 + 2] = n2;
      a.clamped[c.pt] = is_clamped ? 1 : 0;
    }
    if (a.r_aux != nullptr) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (t < plan.aux_tiles) {
          v4f v = r_aux[t];
          if (is_clamped) v = V4ZERO;
          aux_store_tile<Core::AUX_LEVEL_MAJOR>(a.r_aux, av, plan.aux_tiles, t, c.pt, c.q, v);
        }
    }
  }
}

// ---------------------------------------------------------------------------
// B: double backward.  Given d loss/d sdf, d loss/d feat, d loss/d nrm (and d loss/d r_aux),
// sweep up (second-order terms) then down (ordinary backward with the extra a-bar terms),
// leaving the operands of the weight-gradient GEMMs in the workspace:
//   QB_l = q-bar_l (input side),  PM_l (from FG),  AB_l = a-bar_l,  H_l / IN0 (from FG).
// ---------------------------------------------------------------------------
typedef msdf_bw_args_t BwArgs;

template <bool LM>
__device__ __forceinline__ void load_rbar(v4f (&rbar)[5], const msdf_plan_t& plan, const BwArgs& a,
                                          const PointCtx& c, const bool live, const float gn0, const float gn1,
                                          const float gn2) {
  pe_jacobian(rbar, c.x0, c.x1, c.x2, plan.n_freqs, gn0, gn1, gn2);
  rbar[3] = rbar[4] = V4ZERO;
  if (LM && a.dy_dx != nullptr && live) {
    // formed here from the encoder's Jacobian instead of read from a tensor a kernel of its own wrote
    const AuxView av = {a.aux_C, a.aux_LC, a.P};
    const float g0 = __fmul_rn(gn0, a.aux_dx_scale), g1 = __fmul_rn(gn1, a.aux_dx_scale), g2 = __fmul_rn(gn2, a.aux_dx_scale);
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (t < plan.aux_tiles) rbar[3 + t] = aux_jacobian_tile(a.dy_dx, av, t, c.pt, c.q, g0, g1, g2);
  } else if (a.g_raux != nullptr && live) {
    const AuxView av = {a.aux_C, a.aux_LC, a.P};
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (t < plan.aux_tiles) rbar[3 + t] = aux_load_tile<LM>(a.g_raux, av, plan.aux_tiles, t, c.pt, c.q);
  }
}

template <class Core>
__device__ __forceinline__ void sdf_backward_body(const msdf_plan_t& plan, const BwArgs& a, void* lds) {
  typedef typename Core::wvec wvec;
  const PointCtx c = load_point(a.x, a.P);
  const int nl = plan.n_layers;
  const size_t Pp = (size_t)a.P_pad;
  const bool live = c.valid && !(a.clamped != nullptr && a.clamped[c.ptc]);
  float gs = 0.f, gn0 = 0.f, gn1 = 0.f, gn2 = 0.f;
  if (live) {
    // two groups of points with separately produced gradients (ray samples | eikonal points)
    const bool first = c.pt < a.n_split;
    const float* gsp = first ? a.g_sdf : a.g_sdf_b;
    const float* gnp = first ? a.g_nrm : a.g_nrm_b;
    const size_t j = first ? (size_t)c.pt : (size_t)(c.pt - a.n_split);
    if (gsp) gs = gsp[j];
    if (gnp) {
      gn0 = gnp[j * 3 + 0];
      gn1 = gnp[j * 3 + 1];
      gn2 = gnp[j * 3 + 2];
    }
  }
  if (c.q == 0) a.GSDF[c.pt] = gs;
  if (Core::AUX_LEVEL_MAJOR && a.gg_out != nullptr && c.valid && c.q == 0) {
    // the scaled gradient of d sdf / d x, which the embedding scatter's second-order term reads per point (also for
    // clamped points, as hg_node_second_grad_kernel writes it: their d sdf / d features is zero, so nothing comes of it)
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    const bool first = c.pt < a.n_split;
    const float* gnp = first ? a.g_nrm : a.g_nrm_b;
    if (gnp != nullptr) {
      const size_t j = first ? (size_t)c.pt : (size_t)(c.pt - a.n_split);
      r0 = gnp[j * 3 + 0]; r1 = gnp[j * 3 + 1]; r2 = gnp[j * 3 + 2];
    }
    a.gg_out[(size_t)c.pt * 3 + 0] = __fmul_rn(r0, a.aux_dx_scale);
    a.gg_out[(size_t)c.pt * 3 + 1] = __fmul_rn(r1, a.aux_dx_scale);
    a.gg_out[(size_t)c.pt * 3 + 2] = __fmul_rn(r2, a.aux_dx_scale);
  }
  const int in0_tiles = plan.e_tiles + plan.aux_tiles;

  v4f in[MT], acc[MT];
  {
    v4f rbar[5];
    load_rbar<Core::AUX_LEVEL_MAJOR>(rbar, plan, a, c, live, gn0, gn1, gn2);
    place_tiles(in, 0, rbar, in0_tiles);
  }

  // ---------------- sweep up: p-bar_l = W_l q-bar_l ----------------
  for (int l = 0; l < nl - 1; ++l) {
    const msdf_layer_t L = plan.layer[l];
    if (L.skip_tile >= 0) {
      v4f rbar[5];
      load_rbar<Core::AUX_LEVEL_MAJOR>(rbar, plan, a, c, live, gn0, gn1, gn2);
      place_tiles(in, L.skip_tile, rbar, in0_tiles);
    }
    float* Ql = a.QB + (size_t)L.qpre * Pp + (size_t)c.pt * (16 * L.kt) + 4 * c.q;
#pragma unroll
    for (int t = 0; t < MT; ++t)
      if (t < L.kt) *(v4f*)(Ql + 16 * t) = in[t];
    zero_tiles(acc);
    const size_t off = (size_t)L.hpre * Pp + (size_t)c.pt * (16 * L.ot) + 4 * c.q;
    // the hooks turn p-bar into q-bar of the next layer chunk by chunk, loading H / PM and storing T under the
    // product's own matrix instructions
    Core::gemm(L.ktp, acc, in, L.ot, (const wvec*)a.wpack + L.wf_off, lds, SweepUpHooks(a.H + off, L.ot));
#pragma unroll
    for (int t = 0; t < MT; ++t) in[t] = (t < L.ot) ? acc[t] : V4ZERO;
  }
  const msdf_layer_t LL = plan.layer[nl - 1];
  {
    // q-bar of the last layer's input: d W_last[sdf row] = sum_p q-bar (one row of ones on the X side)
    float* Ql = a.QLAST + (size_t)c.pt * (16 * LL.kt) + 4 * c.q;
#pragma unroll
    for (int t = 0; t < MT; ++t)
      if (t < LL.kt) *(v4f*)(Ql + 16 * t) = in[t];
  }

  // ---------------- sweep down ----------------
  // a-bar of the output layer: feature gradient tiles + the sdf slot
  const bool has_feat = (a.g_feat != nullptr) && c.valid && c.pt < a.n_feat;
  {
    float* ABl = a.AB + (size_t)LL.abpre * Pp + (size_t)c.pt * (16 * LL.ot) + 4 * c.q;
    // every load before the first store (as in the hooks: a store next to each load is one memory round trip per tile)
    const float* gf = a.g_feat + (size_t)(has_feat ? c.pt : 0) * (16 * plan.feat_tiles) + 4 * c.q;
    const int ftl = plan.feat_tiles - 1;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      v4f v = V4ZERO;
      if (a.g_feat != nullptr && ftl >= 0) v = *(const v4f*)(gf + 16 * (t < ftl ? t : ftl));
      if (!has_feat || t >= plan.feat_tiles) v = V4ZERO;
      if (t == plan.feat_tiles && c.q == 0) v.x = gs;
      in[t] = v;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
      if (t < LL.ot) *(v4f*)(ABl + 16 * t) = in[t];
  }
  zero_tiles(acc);
  // every product of this sweep yields h-bar of the layer below; its hooks turn that into a-bar of that layer
  // (H / T loaded, AB stored under the product's matrix instructions)
  auto down_hooks = [&](const int l) {     // hooks of the product whose output feeds layer l (l < 0: none)
    const msdf_layer_t Ln = plan.layer[l >= 0 ? l : 0];
    const size_t off = (size_t)Ln.hpre * Pp + (size_t)c.pt * (16 * Ln.ot) + 4 * c.q;
    // q-bar of layer l's outputs = the input rows stored for layer l + 1 (QLAST above the last hidden layer)
    const int ln = (l >= 0 ? l : 0) + 1;
    const msdf_layer_t Lq = plan.layer[ln];
    const float* Q = (ln == nl - 1) ? a.QLAST + (size_t)c.pt * (16 * Lq.kt) + 4 * c.q
                                    : a.QB + (size_t)Lq.qpre * Pp + (size_t)c.pt * (16 * Lq.kt) + 4 * c.q;
    return SweepDownHooks(a.H + off, a.PM + off, Q,
                          a.AB + (size_t)Ln.abpre * Pp + (size_t)c.pt * (16 * Ln.ot) + 4 * c.q, l >= 0 ? Ln.ot : 0);
  };
  Core::gemm(LL.otp, acc, in, LL.kt, (const wvec*)a.wpack + LL.wb_off, lds, down_hooks(nl - 2));
  v4f g_in_aux[5];      // only the aux tiles (hash-grid features) of d loss / d input are an output
#pragma unroll
  for (int t = 0; t < 5; ++t) g_in_aux[t] = V4ZERO;
  for (int l = nl - 2; l >= 0; --l) {
    const msdf_layer_t L = plan.layer[l];
#pragma unroll
    for (int t = 0; t < MT; ++t) in[t] = (t < L.ot) ? acc[t] : V4ZERO;      // a-bar_l
    if (l == 0 && a.g_aux == nullptr) break;   // d loss / d x is not needed: skip the last product
    zero_tiles(acc);
    Core::gemm(L.otp, acc, in, L.kt, (const wvec*)a.wpack + L.wb_off, lds, down_hooks(l - 1));
    if (a.g_aux != nullptr) {
      if (l == 0) gather_tiles(g_in_aux, acc, plan.e_tiles, plan.aux_tiles);
      else if (L.skip_tile >= 0) gather_tiles(g_in_aux, acc, L.skip_tile + plan.e_tiles, plan.aux_tiles);
    }
  }
  if (a.g_aux != nullptr && c.valid) {
    const AuxView av = {a.aux_C, a.aux_LC, a.P};
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (t < plan.aux_tiles) aux_store_tile<Core::AUX_LEVEL_MAJOR>(a.g_aux, av, plan.aux_tiles, t, c.pt, c.q, g_in_aux[t]);
  }
}

// ---------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------
static int mlp_prepare(const void* fn) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_BYTES) == hipSuccess
             ? MSDF_OK : MSDF_ERR_LAUNCH;
}

