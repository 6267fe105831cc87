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

#define KB_MAX ((MT + 1) / 2)                       // 9 k-blocks of 32 slots
#define B16_BUF_V8 (CHUNK_OT * KB_MAX * 2 * 64)     // v8bf (16 B) per LDS buffer: 36 KB
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

// tiles [0, kt) of `t` -> K blocks (zero padding beyond kt)
__device__ __forceinline__ void b16_from_tiles(B16Act& a, const v4f (&t)[MT], const int kt) {
#pragma unroll
  for (int kb = 0; kb < KB_MAX; ++kb) {
    const v4f t0 = (2 * kb < kt) ? t[2 * kb] : V4ZERO;
    const v4f t1 = (2 * kb + 1 < MT && 2 * kb + 1 < kt) ? t[(2 * kb + 1 < MT) ? 2 * kb + 1 : 0] : V4ZERO;
    b16_split2(t0, t1, a.hi[kb], a.lo[kb]);
  }
}

template <int PIECES_MAX>
__device__ __forceinline__ void b16_chunk_issue(const v8bf* __restrict__ src, v8bf* dst, const int n_v8) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int pieces = n_v8 >> 6;
#pragma unroll
  for (int i = 0; i < (PIECES_MAX + 3) / 4; ++i) {
    const int piece = wave + 4 * i;
    if (piece < pieces) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 64 + lane),
                                       (__attribute__((address_space(3))) void*)(dst + piece * 64), 16, 0, 0);
    }
  }
}

// acc[0..OT) += W * act;  W = bf16 hi/lo pack [ceil2(OT)][KB][2][64] v8bf;  KB_T > 0: compile-time K-block count.
template <int KB_T>
__device__ __forceinline__ void gemm_b16(v4f (&acc)[MT], const B16Act& act, const int OT, const int kb_rt,
                                         const v8bf* __restrict__ wsrc, v8bf* lds) {
  constexpr bool DYN = (KB_T == 0);
  constexpr int KMAX = DYN ? KB_MAX : KB_T;
  const int KB = DYN ? kb_rt : KB_T;
  const int ch_v8 = CHUNK_OT * KB * 2 * 64;
  const int lane = threadIdx.x & 63;
  const int nchunks = (OT + CHUNK_OT - 1) / CHUNK_OT;
  b16_chunk_issue<CHUNK_OT * KMAX * 2>(wsrc, lds, ch_v8);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < (MT + 1) / 2; ++c) {
    if (c < nchunks) {
      const int buf = c & 1;
      if (c + 1 < nchunks)
        b16_chunk_issue<CHUNK_OT * KMAX * 2>(wsrc + (size_t)(c + 1) * ch_v8, lds + (buf ^ 1) * B16_BUF_V8, ch_v8);
      const v8bf* w0 = lds + buf * B16_BUF_V8 + lane;      // out tile 2c:   [kb][hi|lo][64]
      const v8bf* w1 = w0 + KB * 2 * 64;                   // out tile 2c+1
      const int o0 = 2 * c;
      const int o1 = (2 * c + 1 < MT) ? 2 * c + 1 : 0;
      const bool two = (2 * c + 1 < MT) && (2 * c + 1 < OT);
      v4f c0 = acc[o0], c1 = acc[o1];
      if (two) {
        v8bf a0h = w0[0], a0l = w0[64], a1h = w1[0], a1l = w1[64];
#pragma unroll
        for (int kb = 0; kb < KMAX; ++kb) {
          if (!DYN || kb < KB) {
            v8bf n0h = a0h, n0l = a0l, n1h = a1h, n1l = a1l;
            if (kb + 1 < KMAX && (!DYN || kb + 1 < KB)) {
              n0h = w0[(kb + 1) * 128]; n0l = w0[(kb + 1) * 128 + 64];
              n1h = w1[(kb + 1) * 128]; n1l = w1[(kb + 1) * 128 + 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, act.hi[kb], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, act.hi[kb], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, act.lo[kb], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, act.lo[kb], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0l, act.hi[kb], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1l, act.hi[kb], c1, 0, 0, 0);
            a0h = n0h; a0l = n0l; a1h = n1h; a1l = n1l;
          }
        }
      } else {
#pragma unroll
        for (int kb = 0; kb < KMAX; ++kb) {
          if (!DYN || kb < KB) {
            const v8bf ah = w0[kb * 128], al = w0[kb * 128 + 64];
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, act.hi[kb], c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, act.lo[kb], c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, act.hi[kb], c0, 0, 0, 0);
          }
        }
      }
      acc[o0] = c0;
      if (two) acc[o1] = c1;
      __syncthreads();
    }
  }
}

__device__ __forceinline__ void gemm_b16_dispatch(const int kbp, v4f (&acc)[MT], const B16Act& act, const int OT,
                                                  const v8bf* __restrict__ wsrc, v8bf* lds) {
  switch (kbp) {
    case 8: gemm_b16<8>(acc, act, OT, 8, wsrc, lds); break;
    case 9: gemm_b16<9>(acc, act, OT, 9, wsrc, lds); break;
    default: gemm_b16<0>(acc, act, OT, kbp, wsrc, lds); break;
  }
}
