// bf16x3 variants of the SDF-network kernels (mlp_core_b16.h).  Same plans / slot maps as the fp32
// kernels of sdf_mlp.hip; the plan handed to these kernels carries K-BLOCK counts (32 slots) in
// ktp / otp and 16-byte offsets of the bf16 hi/lo packs in wf_off / wb_off.
#include "mlp_core_b16.h"

// flat effective weights -> bf16 hi/lo packs in fragment order, both orientations, + fp32 bias / sdf row
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
    const int rt_even = (n_rt + B16_CHUNK_OT - 1) / B16_CHUNK_OT * B16_CHUNK_OT;
    const int total = rt_even * kbp * 64;
    for (int i = t0; i < total; i += stride) {
      const int lane = i & 63;
      const int blk = i >> 6;
      const int kb = blk % kbp, rt = blk / kbp;
      const int rslot = 16 * rt + (lane & 15);
      v8bf hi, lo;
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
        const __bf16 h = (__bf16)w;
        hi[j] = h;
        lo[j] = (__bf16)(w - (float)h);
      }
      v8bf* dst = wpack + off + ((size_t)(rt * kbp + kb) * 2) * 64 + lane;
      dst[0] = hi;
      dst[64] = lo;
    }
  } else {
    for (int i = t0; i < 16 * L.ot; i += stride) {
      const int row = rowmap[i];
      bpack[L.bias_off + i] = (row >= 0) ? flat_b[R.b_off + row] : 0.f;
    }
    if (l == plan.n_layers - 1 && plan.wsdf_off >= 0) {
      const int sdf_row = rowmap[plan.sdf_slot];
      for (int i = t0; i < 16 * L.kt; i += stride) {
        const int col = colmap[i];
        bpack[plan.wsdf_off + i] = (col >= 0) ? R.scale * W[(size_t)sdf_row * R.cols + col] : 0.f;
      }
    }
  }
}

struct PointCtxB {
  int pt, ptc, q;
  bool valid;
  float x0, x1, x2;
};
__device__ __forceinline__ PointCtxB load_point_b(const float* __restrict__ x, const int P) {
  PointCtxB c;
  const int lane = lane_id();
  c.pt = blockIdx.x * B16_PTS_PER_WG + (threadIdx.x >> 6) * MLP_PTS_PER_WAVE + (lane & 15);
  c.valid = c.pt < P;
  c.ptc = c.valid ? c.pt : (P - 1);
  c.q = lane >> 4;
  c.x0 = x[(size_t)c.ptc * 3 + 0];
  c.x1 = x[(size_t)c.ptc * 3 + 1];
  c.x2 = x[(size_t)c.ptc * 3 + 2];
  return c;
}

__device__ __forceinline__ void input_tiles_b(v4f (&in0)[5], const msdf_plan_t& plan, const float* __restrict__ aux,
                                              const PointCtxB& c) {
  pe_values(in0, c.x0, c.x1, c.x2, plan.n_freqs);
  in0[3] = in0[4] = V4ZERO;
  if (plan.aux_tiles > 0) {
    const int aw = 16 * plan.aux_tiles;
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (t < plan.aux_tiles) in0[3 + t] = *(const v4f*)(aux + (size_t)c.ptc * aw + 16 * t + 4 * c.q);
  }
}

// F (bf16x3): forward only, sdf only
__global__ void __launch_bounds__(B16_THREADS, 2)
msdf_sdf_forward_b16_k(const msdf_plan_t plan, const v8bf* __restrict__ wpack, const float* __restrict__ bpack,
                       const float* __restrict__ x, const float* __restrict__ aux, const int P,
                       const float clamp_radius, const float sphere_scale, float* __restrict__ sdf_out) {
  extern __shared__ v8bf lds16[];
  const PointCtxB c = load_point_b(x, P);
  v4f tl[MT];     // input tiles of the current layer, then (in place) its accumulators / activations
  const int in0_tiles = plan.e_tiles + plan.aux_tiles;
  {
    v4f in0[5];
    input_tiles_b(in0, plan, aux, c);
    place_tiles(tl, 0, in0, in0_tiles);
  }
  B16Act act;
  const int nl = plan.n_layers;
  const auto activate = [](const int, v4f& v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#if B16_EXP == 1
      v[r] = fmaxf(v[r], 0.f);
#else
      float hv, s;
      softplus100(v[r], hv, s);
      v[r] = hv;
#endif
    }
  };
  for (int l = 0; l < nl - 1; ++l) {
    const msdf_layer_t L = plan.layer[l];
    if (L.skip_tile >= 0) {
      v4f in0[5];
      input_tiles_b(in0, plan, aux, c);
      place_tiles(tl, L.skip_tile, in0, in0_tiles);
    }
    b16_from_tiles(act, tl, L.kt);
#pragma unroll
    for (int t = 0; t < MT; ++t) tl[t] = (t < L.ot) ? *(const v4f*)(bpack + L.bias_off + 16 * t + 4 * c.q) : V4ZERO;
    gemm_b16_dispatch(L.ktp, tl, act, L.ot, wpack + L.wf_off, lds16, activate);
  }
  // sdf row of the output layer: fp32 dot product on the last hidden activation
  const msdf_layer_t LL = plan.layer[nl - 1];
  float part = 0.f;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    if (t < LL.kt) {
      const v4f w = *(const v4f*)(bpack + plan.wsdf_off + 16 * t + 4 * c.q);
      part += w.x * tl[t].x + w.y * tl[t].y + w.z * tl[t].z + w.w * tl[t].w;
    }
  }
  float sdf = sum_over_quarters(part) + bpack[LL.bias_off + plan.sdf_slot];
  if (clamp_radius > 0.f) {
    const float nx = sqrtf(c.x0 * c.x0 + c.x1 * c.x1 + c.x2 * c.x2);
    sdf = fminf(sdf, sphere_scale * (clamp_radius - nx));
  }
  if (c.valid && c.q == 0) sdf_out[c.pt] = sdf;
}

extern "C" int msdf_pack_weights_b16(const msdf_plan_t* plan, const msdf_packrule_t* rules_dev, const int* maps_dev,
                                     const float* flat_w, const float* flat_b, void* wpack, float* bpack,
                                     void* stream) {
  if (plan == nullptr || plan->n_layers < 1 || plan->n_layers > MSDF_MAX_LAYERS) return MSDF_ERR_ARG;
  const dim3 grid(32, plan->n_layers, 3);
  msdf_pack_b16_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(*plan, rules_dev, maps_dev, flat_w, flat_b,
                                                              (v8bf*)wpack, bpack);
  return msdf_check_launch();
}

extern "C" int msdf_sdf_forward_b16(const msdf_plan_t* plan, const void* wpack, const float* bpack, const float* x,
                                    const float* aux, int P, float clamp_radius, float sphere_scale, float* sdf,
                                    void* stream) {
  if (plan == nullptr || P < 0) return MSDF_ERR_ARG;
  if (P == 0) return MSDF_OK;
  if (plan->aux_tiles > 0 && aux == nullptr) return MSDF_ERR_ARG;
  if (hipFuncSetAttribute((const void*)msdf_sdf_forward_b16_k, hipFuncAttributeMaxDynamicSharedMemorySize,
                          B16_LDS_BYTES) != hipSuccess)
    return MSDF_ERR_LAUNCH;
  const int grid = (P + B16_PTS_PER_WG - 1) / B16_PTS_PER_WG;
  msdf_sdf_forward_b16_k<<<grid, B16_THREADS, B16_LDS_BYTES, (hipStream_t)stream>>>(
      *plan, (const v8bf*)wpack, bpack, x, aux, P, clamp_radius, sphere_scale, sdf);
  return msdf_check_launch();
}
