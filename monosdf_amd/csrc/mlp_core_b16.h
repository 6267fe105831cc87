// bf16-split variants of the register-resident MLP core (see mlp_core.h for the fp32 original): CoreB16N<2> = "bf16x3",
// CoreB16N<3> = "bf16x6".
//
// Every fp32 product a*b is replaced by a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with a = a_hi + a_lo split
// into two bf16 (fp32 accumulation in the MFMA): relative error ~2^-16 per product instead of 2^-8
// for plain bf16 -- measured 1.1e-5 / 1.4e-5 / 1.3e-5 on sdf / features / d sdf/dx of the 8x256
// network (scripts/exp_bf16x3.py), inside the 1e-4 parity bar -- at 3/16 of the fp32-MFMA cycles
// (v_mfma_f32_16x16x32_bf16: 8192 MAC per 16 cycles against 1024 MAC per 32 cycles).
//
// Layout: unchanged on the accumulator side (lane = (point l&15, quarter l>>4), tile t, component r
// <-> slot 16 t + 4 q + r).  The B operand of a K=32 block kb is the pair of tiles (2kb, 2kb+1): lane
// (p, q) supplies k-slots 8q + j, j < 8, = slots 16 (2 kb + j/4) + 4 q + j%4 -- exactly the 8 values it
// already holds -- and the weight pack stores, per (out tile, k block), the matching 8 bf16 per lane,
// hi plane then lo plane (1 KB each).
//
// bf16x6 (NS = 3 planes): a = a1 + a2 + a3 with three bf16 (8 + 8 + 8 = the 24 significand bits of an fp32 value, each
// difference exact in fp32), and a*b ~ a1*b1 + a1*b2 + a2*b1 + a1*b3 + a2*b2 + a3*b1 -- every term down to 2^-16 of the
// product; what is dropped (a2*b3, a3*b2, a3*b3) is below 2^-24, the rounding of an fp32 product itself.  Six
// v_mfma_f32_16x16x32_bf16 per (out tile, 32 k) = 96 cycles against the 256 of eight v_mfma_f32_16x16x4_f32.  Measured
// on an 8 x 256 softplus network against fp64 (emulated on the CPU, scripts/exp_bf16x6.py): fp32 8e-7, bf16x6 1e-7,
// bf16x3 1.3e-5.  Its parity tests are held to the fp32 core's rows of the tolerance table.
#pragma once
#include "mlp_core.h"


typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) v8bf lds_v8bf;

// softplus(beta=100) for the bf16x3 kernels: 2 transcendental + 4 plain VALU instructions per value.
// Against softplus100() it drops the two selects (threshold 20: difference < 2.1e-11; log1p series for
// e < 1e-4: absolute difference < 6e-10 on values whose bf16x3 products carry ~1e-5 relative error).
__device__ __forceinline__ float softplus100_lean(const float a) {
  const float e = __builtin_amdgcn_exp2f(-fabsf(a * 144.26950408889634f));
  // max(a, 0) as one integer max on the bit pattern (fmaxf would add a canonicalising v_max(a, a); inline
  // asm is not an option: the hazard recogniser does not pad MFMA -> inline-asm reads)
  const float relu = __int_as_float(max(__float_as_int(a), 0));
  return fmaf(0.006931471805599453f, __builtin_amdgcn_logf(1.0f + e), relu);
}

#define KB_MAX ((MT + 1) / 2)                       // 9 k-blocks of 32 slots
// Same workgroup shape as the fp32 core: 4 waves x 16 points, two workgroups per CU, so that one workgroup's
// load/store-heavy epilogue overlaps the other's matrix products.  (8 waves sharing 4-tile chunks -- half the
// L2 -> LDS traffic -- measured the same on the forward kernel: the weight stream is not what bounds it, see DESIGN.md.)
#define B16_WAVES 4
#define B16_THREADS (64 * B16_WAVES)
#define B16_PTS_PER_WG (16 * B16_WAVES)

// NS planes per operand.  An LDS weight chunk holds CH out tiles: two with two planes (36 KB per buffer), one with three
// (27 KB) -- either way two workgroups fit a CU and a chunk is 54 matrix instructions at K = 288.
template <int NS>
struct B16Cfg {
  static_assert(NS == 2 || NS == 3, "two planes (bf16x3) or three (bf16x6)");
  static constexpr int CH = (NS == 2) ? 2 : 1;
  static constexpr int BUF_V8 = CH * KB_MAX * NS * 64;      // v8bf (16 B) per LDS buffer
  static constexpr int LDS_BYTES = 2 * BUF_V8 * 16;         // 72 KB / 54 KB: two workgroups per CU
  static constexpr int RING = (NS == 2) ? 3 : 2;            // fragment sets in flight (NS * RING * 4 registers; 3 x 3 spills: 7.7 -> 8.2 ms)
};

template <int NS>
struct B16Act {           // an activation vector as MFMA B operands, plane 0 = the leading bf16
  v8bf p[NS][KB_MAX];
};

template <int NS>
__device__ __forceinline__ void b16_split(const v4f t0, const v4f t1, v8bf (&out)[NS]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float r0 = t0[j], r1 = t1[j];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      const __bf16 h0 = (__bf16)r0;
      const __bf16 h1 = (__bf16)r1;
      out[n][j] = h0;
      out[n][4 + j] = h1;
      if (n + 1 < NS) {
        r0 -= (float)h0;      // exact: the difference of an fp32 value and its leading bits
        r1 -= (float)h1;
      }
    }
  }
}

// all MT tiles of `t` -> K blocks.  Tiles past a layer's input width must hold finite values (the kernels
// keep them at zero): their weights are zero in the pack, so they contribute nothing.
template <int NS>
__device__ __forceinline__ void b16_from_tiles(B16Act<NS>& a, const v4f (&t)[MT]) {
#pragma unroll
  for (int kb = 0; kb < KB_MAX; ++kb) {
    const v4f t0 = t[2 * kb];
    const v4f t1 = (2 * kb + 1 < MT) ? t[(2 * kb + 1 < MT) ? 2 * kb + 1 : 0] : V4ZERO;
    v8bf planes[NS];
    b16_split<NS>(t0, t1, planes);
#pragma unroll
    for (int n = 0; n < NS; ++n) a.p[n][kb] = planes[n];
  }
}

template <int PIECES_MAX>
__device__ __forceinline__ void b16_chunk_issue(const v8bf* __restrict__ src, v8bf* dst, const int n_v8) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int pieces = n_v8 >> 6;
#pragma unroll
  for (int i = 0; i < (PIECES_MAX + B16_WAVES - 1) / B16_WAVES; ++i) {
    const int piece = wave + B16_WAVES * i;
    if (piece < pieces) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 64 + lane),
                                       (__attribute__((address_space(3))) void*)(dst + piece * 64), 16, 0, 0);
    }
  }
}

// c += (sum of the kept cross terms of) W-fragment planes f[] times activation planes of k block kb
template <int NS>
__device__ __forceinline__ v4f b16_step(v4f c, const v8bf (&f)[NS], const B16Act<NS>& act, const int kb) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], act.p[0][kb], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], act.p[1][kb], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1], act.p[0][kb], c, 0, 0, 0);
  if constexpr (NS == 3) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], act.p[2][kb], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1], act.p[1][kb], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[2], act.p[0][kb], c, 0, 0, 0);
  }
  return c;
}

// one out tile against all K blocks (runtime K-block count)
template <int NS>
__device__ __forceinline__ v4f b16_tile_dyn(v4f c, const v8bf* __restrict__ w, const B16Act<NS>& act, const int KB) {
  v8bf a[NS];
#pragma unroll
  for (int n = 0; n < NS; ++n) a[n] = w[n * 64];
#pragma unroll
  for (int kb = 0; kb < KB_MAX; ++kb) {
    if (kb < KB) {
      v8bf nx[NS];
#pragma unroll
      for (int n = 0; n < NS; ++n) nx[n] = a[n];
      if (kb + 1 < KB_MAX && kb + 1 < KB) {
#pragma unroll
        for (int n = 0; n < NS; ++n) nx[n] = w[((kb + 1) * NS + n) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);     // keep the prefetch of the next fragments above the MFMAs
      c = b16_step<NS>(c, a, act, kb);
#pragma unroll
      for (int n = 0; n < NS; ++n) a[n] = nx[n];
    }
  }
  return c;
}

// acc[0..OT) += W * act;  W = bf16 plane pack [ceil2(OT)][KB][NS][64] v8bf;  KB_T > 0: compile-time K-block count.
// The product's hooks (mlp_core.h, NoHooks) are called per PAIR of out tiles: pre() after the barrier that opens the
// pair's (first) chunk, post() after its last MFMA -- the activation's VALU work and the epilogue's memory traffic
// overlap the other wave's MFMAs instead of forming a phase of their own.
// All B16_THREADS threads of the workgroup call this together (barriers inside).
template <int NS, int KB_T, class Hooks>
__device__ __forceinline__ void gemm_b16(v4f (&acc)[MT], const B16Act<NS>& act, const int OT, const int kb_rt,
                                         const v8bf* __restrict__ wsrc, v8bf* lds, Hooks& hk) {
  typedef B16Cfg<NS> Cfg;
  constexpr int CH = Cfg::CH;
  constexpr int RING = Cfg::RING;
  constexpr bool DYN = (KB_T == 0);
  constexpr int KMAX = DYN ? KB_MAX : KB_T;
  const int KB = DYN ? kb_rt : KB_T;
  const int ch_v8 = CH * KB * NS * 64;
  const int lane = threadIdx.x & 63;
  const int nchunks = (OT + CH - 1) / CH;
  constexpr int BUF = Cfg::BUF_V8;
  b16_chunk_issue<CH * KMAX * NS>(wsrc, lds, ch_v8);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < (MT + CH - 1) / CH; ++c) {
    if (c < nchunks) {
      const int buf = c & 1;
      if (c + 1 < nchunks)
        b16_chunk_issue<CH * KMAX * NS>(wsrc + (size_t)(c + 1) * ch_v8, lds + (buf ^ 1) * BUF, ch_v8);
      const int t0 = CH * c;                                // first out tile of this chunk: a compile-time constant
      if ((t0 & 1) == 0) hk.pre(t0, t0 + 1);
      const v8bf* w = lds + buf * BUF + lane;               // [ot in chunk][kb][plane][64]
      // LDS base of this chunk as an opaque 32-bit register: every fragment read below then carries its
      // offset in the instruction's 16-bit immediate instead of a v_add per read (VALU issue slots are
      // what this kernel runs out of: a 16x16x32 MFMA blocks the SIMD's vector issue for 8 of its 16 cycles)
      const lds_v8bf* wl0 = (const lds_v8bf*)w;
      asm volatile("" : "+v"(wl0));
      static_assert(BUF * 16 <= 65536, "fragment offsets must fit the ds_read immediate");
#define B16_FRAG(i, plane) (wl0[((i) * NS + (plane)) * 64])
      if constexpr (DYN) {
#pragma unroll
        for (int o = 0; o < CH; ++o) {
          const int ot = CH * c + o;
          if (ot < MT && ot < OT) {
            const int oi = (ot < MT) ? ot : 0;
            acc[oi] = b16_tile_dyn<NS>(acc[oi], w + o * KB * NS * 64, act, KB);
          }
        }
      } else {
        // the chunk as one flat (out tile, k block) sequence with the fragments of step i + RING - 1 in flight;
        // the pack pads the out tiles to an even count, so the look-ahead never leaves the buffer
        const int n_o = OT - CH * c;
        v8bf fr[RING][NS];
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) {
          if (i < CH * KB_T) {
#pragma unroll
            for (int n = 0; n < NS; ++n) fr[i][n] = B16_FRAG(i, n);
          }
        }
#pragma unroll
        for (int o = 0; o < CH; ++o) {
          const int ot = CH * c + o;
          if (ot < MT && o < n_o) {
            const int oi = (ot < MT) ? ot : 0;
            v4f cc = acc[oi];
#pragma unroll
            for (int kb = 0; kb < KB_T; ++kb) {
              const int i = o * KB_T + kb;
              if (i + RING - 1 < CH * KB_T) {
#pragma unroll
                for (int n = 0; n < NS; ++n) fr[(i + RING - 1) % RING][n] = B16_FRAG(i + RING - 1, n);
              }
              __builtin_amdgcn_sched_barrier(0);
              cc = b16_step<NS>(cc, fr[i % RING], act, kb);
            }
            acc[oi] = cc;
          }
        }
      }
#undef B16_FRAG
      const int t_last = CH * c + CH - 1;                   // last out tile of this chunk
      if ((t_last & 1) == 1 || t_last + 1 >= OT) {
        const int e = t_last & ~1;                          // the pair (e, e + 1) is complete (or e is the odd last tile)
        if (e < MT)
          hk.post(e, e + 1, e + 1 < MT && e + 1 < OT, acc[e < MT ? e : 0], acc[(e + 1 < MT) ? e + 1 : 0]);
      }
      __syncthreads();
    }
  }
}

template <int NS, class Hooks>
__device__ __forceinline__ void gemm_b16_dispatch(const int kbp, v4f (&acc)[MT], const B16Act<NS>& act, const int OT,
                                                  const v8bf* __restrict__ wsrc, v8bf* lds, Hooks& hk) {
  switch (kbp) {
    case 2: gemm_b16<NS, 2>(acc, act, OT, 2, wsrc, lds, hk); break;     // PE (3 tiles)
    case 3: gemm_b16<NS, 3>(acc, act, OT, 3, wsrc, lds, hk); break;     // PE + hash-grid features (5 tiles)
    case 8: gemm_b16<NS, 8>(acc, act, OT, 8, wsrc, lds, hk); break;
    case 9: gemm_b16<NS, 9>(acc, act, OT, 9, wsrc, lds, hk); break;
    default: gemm_b16<NS, 0>(acc, act, OT, kbp, wsrc, lds, hk); break;
  }
  hk.drain();
}

template <int NS>
struct CoreB16N {
  typedef v8bf wvec;
  static constexpr int PLANES = NS;
  // three planes only: the two-plane core keeps its accumulators starting from the bias (its results, and the rows of
  // the tolerance table measured on them, stay as they are; it gains 1 % from the change, the three-plane core 10 %)
  static constexpr bool BIAS_IN_HOOKS = (NS == 3);
  static constexpr bool AUX_LEVEL_MAJOR = false;      // rows + transposes: these kernels have no registers to spare
  // two planes: the lean softplus (its differences are far below that core's product error); three planes carry
  // fp32-grade products, so the activation is the fp32 core's
  static __device__ __forceinline__ float softplus(const float a) {
    if constexpr (NS == 2) return softplus100_lean(a);
    float h, s;
    softplus100(a, h, s);
    return h;
  }
  // kbp: K blocks (32 slots) of the pack, i.e. ktp / otp of the bf16 plan
  template <class Hooks>
  static __device__ __forceinline__ void gemm(const int kbp, v4f (&acc)[MT], const v4f (&in)[MT], const int OT,
                                              const wvec* __restrict__ wsrc, void* lds, Hooks&& hk) {
    B16Act<NS> act;
    b16_from_tiles<NS>(act, in);
    gemm_b16_dispatch<NS>(kbp, acc, act, OT, wsrc, (v8bf*)lds, hk);
  }
  // acc = W in + bias (the bias is added to the finished tiles, BiasHooks of mlp_core.h); tiles >= OT leave as zero
  template <class Hooks>
  static __device__ __forceinline__ void gemm_bias(const int kbp, v4f (&acc)[MT], const v4f (&in)[MT], const int OT,
                                                   const wvec* __restrict__ wsrc, void* lds, Hooks&& hk,
                                                   const float* __restrict__ bias) {
    B16Act<NS> act;
    b16_from_tiles<NS>(act, in);
    zero_tiles(acc);
    BiasHooks<typename std::remove_reference<Hooks>::type> bh(hk, bias, OT);
    gemm_b16_dispatch<NS>(kbp, acc, act, OT, wsrc, (v8bf*)lds, bh);
  }
};
typedef CoreB16N<2> CoreB16;       // "bf16x3"
typedef CoreB16N<3> CoreB16X6;     // "bf16x6"
