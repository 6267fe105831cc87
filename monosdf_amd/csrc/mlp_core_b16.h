// bf16x3 variant of the register-resident MLP core (see mlp_core.h for the fp32 original).
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
// Same workgroup shape as the fp32 core: 4 waves x 16 points, two workgroups per CU (72 KB of LDS each), so
// that one workgroup's load/store-heavy epilogue overlaps the other's matrix products.  (8 waves sharing
// 4-tile chunks -- half the L2 -> LDS traffic -- measured the same on the forward kernel: the weight
// stream is not what bounds it, see DESIGN.md.)
#define B16_WAVES 4
#define B16_THREADS (64 * B16_WAVES)
#define B16_PTS_PER_WG (16 * B16_WAVES)
#define B16_CHUNK_OT 2
#define B16_BUF_V8 (B16_CHUNK_OT * KB_MAX * 2 * 64) // v8bf (16 B) per LDS buffer: 36 KB
#define B16_LDS_BYTES (2 * B16_BUF_V8 * 16)

struct B16Act {           // an activation vector as MFMA B operands
  v8bf hi[KB_MAX];
  v8bf lo[KB_MAX];
};

__device__ __forceinline__ void b16_split2(const v4f t0, const v4f t1, v8bf& hi, v8bf& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 h0 = (__bf16)t0[j];
    const __bf16 h1 = (__bf16)t1[j];
    hi[j] = h0;
    hi[4 + j] = h1;
    lo[j] = (__bf16)(t0[j] - (float)h0);
    lo[4 + j] = (__bf16)(t1[j] - (float)h1);
  }
}

// all MT tiles of `t` -> K blocks.  Tiles past a layer's input width must hold finite values (the kernels
// keep them at zero): their weights are zero in the pack, so they contribute nothing.
__device__ __forceinline__ void b16_from_tiles(B16Act& a, const v4f (&t)[MT]) {
#pragma unroll
  for (int kb = 0; kb < KB_MAX; ++kb) {
    const v4f t0 = t[2 * kb];
    const v4f t1 = (2 * kb + 1 < MT) ? t[(2 * kb + 1 < MT) ? 2 * kb + 1 : 0] : V4ZERO;
    b16_split2(t0, t1, a.hi[kb], a.lo[kb]);
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

// one out tile against all K blocks (runtime K-block count): c += W[ot] * act (3 MFMAs per K block)
__device__ __forceinline__ v4f b16_tile_dyn(v4f c, const v8bf* __restrict__ w, const B16Act& act, const int KB) {
  v8bf ah = w[0], al = w[64];
#pragma unroll
  for (int kb = 0; kb < KB_MAX; ++kb) {
    if (kb < KB) {
      v8bf nh = ah, nl = al;
      if (kb + 1 < KB_MAX && kb + 1 < KB) {
        nh = w[(kb + 1) * 128];
        nl = w[(kb + 1) * 128 + 64];
      }
      __builtin_amdgcn_sched_barrier(0);     // keep the prefetch of the next fragments above the MFMAs
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, act.hi[kb], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, act.lo[kb], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, act.hi[kb], c, 0, 0, 0);
      ah = nh;
      al = nl;
    }
  }
  return c;
}

// acc[0..OT) += W * act;  W = bf16 hi/lo pack [ceil4(OT)][KB][2][64] v8bf;  KB_T > 0: compile-time K-block count.
// The product's hooks (mlp_core.h, NoHooks) are called per chunk of two out tiles: pre() after the barrier that
// opens the chunk, post() after its last MFMA -- the activation's VALU work and the epilogue's memory traffic
// overlap the other wave's MFMAs instead of forming a phase of their own.
// All B16_THREADS threads of the workgroup call this together (barriers inside).
template <int KB_T, class Hooks>
__device__ __forceinline__ void gemm_b16(v4f (&acc)[MT], const B16Act& act, const int OT, const int kb_rt,
                                         const v8bf* __restrict__ wsrc, v8bf* lds, Hooks& hk) {
  constexpr bool DYN = (KB_T == 0);
  constexpr int KMAX = DYN ? KB_MAX : KB_T;
  const int KB = DYN ? kb_rt : KB_T;
  const int ch_v8 = B16_CHUNK_OT * KB * 2 * 64;
  const int lane = threadIdx.x & 63;
  const int nchunks = (OT + B16_CHUNK_OT - 1) / B16_CHUNK_OT;
  b16_chunk_issue<B16_CHUNK_OT * KMAX * 2>(wsrc, lds, ch_v8);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < (MT + B16_CHUNK_OT - 1) / B16_CHUNK_OT; ++c) {
    if (c < nchunks) {
      const int buf = c & 1;
      if (c + 1 < nchunks)
        b16_chunk_issue<B16_CHUNK_OT * KMAX * 2>(wsrc + (size_t)(c + 1) * ch_v8, lds + (buf ^ 1) * B16_BUF_V8, ch_v8);
      static_assert(B16_CHUNK_OT == 2, "the hooks take pairs of out tiles");
      hk.pre(2 * c, 2 * c + 1);
      const v8bf* w = lds + buf * B16_BUF_V8 + lane;      // [ot in chunk][kb][hi|lo][64]
      // LDS base of this chunk as an opaque 32-bit register: every fragment read below then carries its
      // offset in the instruction's 16-bit immediate instead of a v_add per read (VALU issue slots are
      // what this kernel runs out of: a 16x16x32 MFMA blocks the SIMD's vector issue for 8 of its 16 cycles)
      const lds_v8bf* wl0 = (const lds_v8bf*)w;
      asm volatile("" : "+v"(wl0));
      static_assert(B16_BUF_V8 * 16 <= 65536, "fragment offsets must fit the ds_read immediate");
#define B16_FRAG(i, half) (wl0[(i) * 128 + (half) * 64])
      if constexpr (DYN) {
#pragma unroll
        for (int o = 0; o < B16_CHUNK_OT; ++o) {
          const int ot = B16_CHUNK_OT * c + o;
          if (ot < MT && ot < OT) {
            const int oi = (ot < MT) ? ot : 0;
            acc[oi] = b16_tile_dyn(acc[oi], w + o * KB * 128, act, KB);
          }
        }
      } else {
        // the chunk as one flat (out tile, k block) sequence with the fragments of step i+2 in flight;
        // the pack pads every chunk to B16_CHUNK_OT tiles, so the look-ahead never leaves the buffer
        const int n_o = OT - B16_CHUNK_OT * c;
        v8bf fh[3], fl[3];
        fh[0] = B16_FRAG(0, 0); fl[0] = B16_FRAG(0, 1);
        fh[1] = B16_FRAG(1, 0); fl[1] = B16_FRAG(1, 1);
#pragma unroll
        for (int o = 0; o < B16_CHUNK_OT; ++o) {
          const int ot = B16_CHUNK_OT * c + o;
          if (ot < MT && o < n_o) {
            const int oi = (ot < MT) ? ot : 0;
            v4f cc = acc[oi];
#pragma unroll
            for (int kb = 0; kb < KB_T; ++kb) {
              const int i = o * KB_T + kb;
              if (i + 2 < B16_CHUNK_OT * KB_T) {
                fh[(i + 2) % 3] = B16_FRAG(i + 2, 0);
                fl[(i + 2) % 3] = B16_FRAG(i + 2, 1);
              }
              __builtin_amdgcn_sched_barrier(0);
              cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[i % 3], act.hi[kb], cc, 0, 0, 0);
              cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[i % 3], act.lo[kb], cc, 0, 0, 0);
              cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[i % 3], act.hi[kb], cc, 0, 0, 0);
            }
            acc[oi] = cc;
          }
        }
      }
#undef B16_FRAG
      hk.post(2 * c, 2 * c + 1, 2 * c + 1 < MT && 2 * c + 1 < OT, acc[2 * c], acc[(2 * c + 1 < MT) ? 2 * c + 1 : 0]);
      __syncthreads();
    }
  }
}

template <class Hooks>
__device__ __forceinline__ void gemm_b16_dispatch(const int kbp, v4f (&acc)[MT], const B16Act& act, const int OT,
                                                  const v8bf* __restrict__ wsrc, v8bf* lds, Hooks& hk) {
  switch (kbp) {
    case 2: gemm_b16<2>(acc, act, OT, 2, wsrc, lds, hk); break;     // PE (3 tiles)
    case 3: gemm_b16<3>(acc, act, OT, 3, wsrc, lds, hk); break;     // PE + hash-grid features (5 tiles)
    case 8: gemm_b16<8>(acc, act, OT, 8, wsrc, lds, hk); break;
    case 9: gemm_b16<9>(acc, act, OT, 9, wsrc, lds, hk); break;
    default: gemm_b16<0>(acc, act, OT, kbp, wsrc, lds, hk); break;
  }
  hk.drain();
}

struct CoreB16 {
  typedef v8bf wvec;
  static __device__ __forceinline__ float softplus(const float a) { return softplus100_lean(a); }
  // kbp: K blocks (32 slots) of the pack, i.e. ktp / otp of the bf16 plan
  template <class Hooks>
  static __device__ __forceinline__ void gemm(const int kbp, v4f (&acc)[MT], const v4f (&in)[MT], const int OT,
                                              const wvec* __restrict__ wsrc, void* lds, Hooks&& hk) {
    B16Act act;
    b16_from_tiles(act, in);
    gemm_b16_dispatch(kbp, acc, act, OT, wsrc, (v8bf*)lds, hk);
  }
};
