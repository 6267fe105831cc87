// Register-resident MLP core for gfx950 (wave64, fp32-input MFMA).
//
// One wave owns a tile of 16 points.  An activation vector of up to 272 slots
// is held as 17 float4 "tiles": lane l = (point p = l & 15, quarter q = l >> 4),
// tile t, component r  <->  slot 16 t + 4 q + r of point p.  That is exactly the
// C/D layout of v_mfma_f32_16x16x4_f32 computing D[out][pt] = W[out][k] * act[k][pt]
// (col = lane & 15 = point, row = 4 (lane >> 4) + reg), and at the same time its
// B-operand layout for the NEXT layer (B[k = lane >> 4][pt = lane & 15]) with the
// k index permuted -- the permutation is absorbed by packing the weights in
// "fragment order": for every (out tile ot, k tile kt) a 1 KB block whose float4 of
// lane l holds W[16 ot + (l & 15)][16 kt + 4 (l >> 4) + 0..3].  Activations never
// leave registers between layers; weights stream HBM/L2 -> LDS (linear copy of
// the packed image, LDS-DMA) -> one conflict-free ds_read_b128 per 4 MFMAs.
//
// A workgroup is 4 waves (64 points); the 4 waves share every weight chunk.
#pragma once
#include <type_traits>
#include "common.h"
#include "../../include/monosdf_plan.h"

#define MT MSDF_MAX_TILES   // 17
#ifndef MLP_WAVES
#define MLP_WAVES 4                      // waves per workgroup (all of them share every weight chunk).  The library is
                                         // built and tested with 4; 8 was a timing experiment of the fp32 kernels only
                                         // (slower) and is not a supported configuration
#endif
#define MLP_THREADS (64 * MLP_WAVES)
#define MLP_PTS_PER_WAVE 16
#define MLP_PTS_PER_WG (16 * MLP_WAVES)
#define MLP_WGS_PER_CU (8 / MLP_WAVES)   // 256 registers per lane: two waves per SIMD
#define CHUNK_OT 2                       // out tiles per LDS chunk
#define LDS_BUF_F4 (CHUNK_OT * MT * 64)  // float4 per LDS buffer (34 KB)
#define MLP_LDS_BYTES (2 * LDS_BUF_F4 * 16)

#define V4ZERO ((v4f){0.f, 0.f, 0.f, 0.f})

__device__ __forceinline__ v4f mfma4(const v4f a, const v4f b, v4f c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
  return c;
}

// Copy one weight chunk (n_f4 float4, a multiple of 64) global -> LDS with LDS-DMA, linear image.
template <int PIECES_MAX>
__device__ __forceinline__ void chunk_issue(const v4f* __restrict__ src, v4f* dst, const int n_f4) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: keeps the branch below uniform
  const int lane = threadIdx.x & 63;
  const int pieces = n_f4 >> 6;   // 1 KB pieces, one wave-instruction each
#pragma unroll
  for (int i = 0; i < (PIECES_MAX + MLP_WAVES - 1) / MLP_WAVES; ++i) {
    const int piece = wave + MLP_WAVES * i;
    if (piece < pieces) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 64 + lane),
                                       (__attribute__((address_space(3))) void*)(dst + piece * 64), 16, 0, 0);
    }
  }
}

// acc[0..OT) += W * in, W given as the fragment-ordered pack [ceil2(OT)][K][64] float4.
// KT_T > 0: K is the compile-time constant KT_T (fully unrolled, no guards);
// KT_T == 0: K = k_rt at run time (every k tile guarded by a wave-uniform branch).
// All 256 threads of the workgroup must call this together (it contains barriers).
// `lds` points at 2 * LDS_BUF_F4 float4; on return every wave has passed a barrier
// after its last LDS read, so the caller may start the next gemm immediately.
// Hooks of a product: the epilogue of the out tiles is spread over the product instead of forming a phase of
// its own after it.  Out tiles are finished two at a time (one weight chunk); for every chunk gemm_tiles calls
//   hk.pre(o0, o1)                 right after the barrier that opens the chunk and the LDS-DMA of the next
//                                  one: the place to ISSUE memory operations -- the stores a previous post()
//                                  left pending and the loads post() of THIS chunk will need.  Everything
//                                  issued here completes under the chunk's 128 MFMAs (>= 2 us), i.e. before
//                                  the s_waitcnt vmcnt(0) of the next barrier instead of in front of it;
//   hk.post(o0, o1, pair, a0, a1)  after the chunk's last MFMA: a0 = acc[o0], a1 = acc[o1] are complete
//                                  (a1 only if `pair`); transform them in place, keep what has to be stored;
// and the caller runs hk.drain() after the product for the stores of the last chunk.  o0 / o1 are compile-time
// constants at every call (the chunk loop is unrolled), everything else is wave-uniform.
struct NoHooks {
  __device__ __forceinline__ void pre(const int, const int) {}
  __device__ __forceinline__ void post(const int, const int, const bool, v4f&, v4f&) {}
  __device__ __forceinline__ void drain() {}
};

// Adds the bias to a pair of finished out tiles in front of the product's own hooks: the accumulators then start from
// zero (no registers until a tile's first matrix instruction) and the pair's two bias tiles fly under its chunk.
// With all 17 bias tiles loaded before the product -- 68 registers beside the input vector -- the forward kernels of
// the three-plane bf16 core spilled 236 bytes per lane (~1 GB of scratch traffic per launch) and the fp32 forward +
// gradient kernel 208.  (W x) + b instead of b + (W x): the same value up to the last rounding.  The two-plane bf16
// core keeps the old order (mlp_core_b16.h).
template <class Inner>
struct BiasHooks {
  Inner& inner;
  const float* bias;          // the lane's pointer into the packed bias row (+ 4 q)
  int ot;
  v4f b0, b1;
  __device__ __forceinline__ BiasHooks(Inner& in_, const float* bias_, const int ot_) : inner(in_), bias(bias_), ot(ot_) {}
  __device__ __forceinline__ void pre(const int o0, const int o1) {
    inner.pre(o0, o1);
    b0 = *(const v4f*)(bias + 16 * (o0 < ot ? o0 : ot - 1));
    b1 = *(const v4f*)(bias + 16 * (o1 < ot ? o1 : ot - 1));
  }
  __device__ __forceinline__ void post(const int o0, const int o1, const bool pair, v4f& a0, v4f& a1) {
    a0 += b0;
    if (pair) a1 += b1;
    inner.post(o0, o1, pair, a0, a1);
  }
  __device__ __forceinline__ void drain() { inner.drain(); }
};

// PPC = pairs of out tiles per weight chunk: 1 for the wide products; narrow ones (K = 3, 5 tiles: the network
// input layers, the misc block of the colour network) take several pairs per chunk -- a pair is then only 24 / 40
// matrix instructions, too little work between two barriers.
template <int KT_T, int PPC, class Hooks>
__device__ __forceinline__ void gemm_tiles(v4f (&acc)[MT], const v4f (&in)[MT], const int OT, const int k_rt,
                                           const v4f* __restrict__ wsrc, v4f* lds, Hooks& hk) {
  constexpr bool DYN = (KT_T == 0);
  constexpr int KMAX = DYN ? MT : KT_T;
  static_assert(PPC * CHUNK_OT * KMAX * 64 <= LDS_BUF_F4, "chunk larger than an LDS buffer");
  const int K = DYN ? k_rt : KT_T;
  const int pair_f4 = CHUNK_OT * K * 64;
  const int lane = threadIdx.x & 63;
  const int npairs = (OT + CHUNK_OT - 1) / CHUNK_OT;
  const int nchunks = (npairs + PPC - 1) / PPC;
  constexpr int MAXPAIRS = (MT + 1) / 2;
  constexpr int MAXCHUNKS = (MAXPAIRS + PPC - 1) / PPC;
  // Chunks are consumed from the LAST pairs of out tiles down to the first.
  {
    const int p0 = (nchunks - 1) * PPC;
    chunk_issue<PPC * CHUNK_OT * KMAX>(wsrc + (size_t)p0 * pair_f4, lds + ((nchunks - 1) & 1) * LDS_BUF_F4,
                                       min(PPC, npairs - p0) * pair_f4);
  }
  __syncthreads();
#pragma unroll
  for (int cc = MAXCHUNKS - 1; cc >= 0; --cc) {
    if (cc < nchunks) {
      const int buf = cc & 1;
      if (cc > 0)
        chunk_issue<PPC * CHUNK_OT * KMAX>(wsrc + (size_t)(cc - 1) * PPC * pair_f4, lds + (buf ^ 1) * LDS_BUF_F4,
                                           PPC * pair_f4);
#pragma unroll
      for (int pl = PPC - 1; pl >= 0; --pl) {
        const int c = cc * PPC + pl;          // pair of out tiles (2c, 2c + 1): a compile-time constant here
        if (c < MAXPAIRS && c < npairs) {
      hk.pre(2 * c, 2 * c + 1);
      const v4f* w0 = lds + buf * LDS_BUF_F4 + pl * pair_f4 + lane;
      const v4f* w1 = w0 + K * 64;
      const int o0 = 2 * c;
      const int o1 = (2 * c + 1 < MT) ? 2 * c + 1 : 0;   // the dead pair of the last odd tile
      if (2 * c + 1 < MT && 2 * c + 1 < OT) {
        if (!DYN) {
          // software-pipelined: the A fragments of k tile kt+1 are in flight while the 8 MFMAs of kt issue
          v4f a0 = w0[0], a1 = w1[0];
          v4f c0 = acc[o0], c1 = acc[o1];
#pragma unroll
          for (int kt = 0; kt < KMAX; ++kt) {
            v4f n0 = a0, n1 = a1;
            if (kt + 1 < KMAX) {
              n0 = w0[(kt + 1) * 64];
              n1 = w1[(kt + 1) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch above the MFMAs
            const v4f b = in[kt];
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, c1, 0, 0, 0);
            a0 = n0;
            a1 = n1;
          }
          acc[o0] = c0;
          acc[o1] = c1;
        } else {
#pragma unroll
          for (int kt = 0; kt < KMAX; ++kt) {
            if (kt < K) {
              const v4f a0 = w0[kt * 64];
              const v4f a1 = w1[kt * 64];
              const v4f b = in[kt];
              v4f c0 = acc[o0], c1 = acc[o1];
              c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, c1, 0, 0, 0);
              c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, c1, 0, 0, 0);
              c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, c1, 0, 0, 0);
              c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, c1, 0, 0, 0);
              acc[o0] = c0;
              acc[o1] = c1;
            }
          }
        }
      } else {
        if (!DYN) {
          v4f a0 = w0[0];
          v4f c0 = acc[o0];
#pragma unroll
          for (int kt = 0; kt < KMAX; ++kt) {
            v4f n0 = a0;
            if (kt + 1 < KMAX) n0 = w0[(kt + 1) * 64];
            __builtin_amdgcn_sched_barrier(0);
            c0 = mfma4(a0, in[kt], c0);
            a0 = n0;
          }
          acc[o0] = c0;
        } else {
#pragma unroll
          for (int kt = 0; kt < KMAX; ++kt) {
            if (kt < K) acc[o0] = mfma4(w0[kt * 64], in[kt], acc[o0]);
          }
        }
      }
      hk.post(o0, 2 * c + 1, 2 * c + 1 < MT && 2 * c + 1 < OT, acc[o0], acc[o1]);
        }
      }
      __syncthreads();
    }
  }
}

// Specialised for the K values of the 256-wide networks (3 = PE, 5 = PE + hash-grid features, 16, 17 = skip
// layer); anything else takes the guarded path.
template <class Hooks>
__device__ __forceinline__ void gemm_dispatch(const int kp, v4f (&acc)[MT], const v4f (&in)[MT], const int OT,
                                              const v4f* __restrict__ wsrc, v4f* lds, Hooks& hk) {
  switch (kp) {
    case 3: gemm_tiles<3, 4>(acc, in, OT, 3, wsrc, lds, hk); break;
    case 5: gemm_tiles<5, 3>(acc, in, OT, 5, wsrc, lds, hk); break;      // PE + hash-grid features
    case 16: gemm_tiles<16, 1>(acc, in, OT, 16, wsrc, lds, hk); break;
    case 17: gemm_tiles<17, 1>(acc, in, OT, 17, wsrc, lds, hk); break;
    default: gemm_tiles<0, 1>(acc, in, OT, kp, wsrc, lds, hk); break;
  }
  hk.drain();
}

__device__ __forceinline__ void zero_tiles(v4f (&a)[MT]) {
#pragma unroll
  for (int t = 0; t < MT; ++t) a[t] = V4ZERO;
}

// dst[base + j] = src[j] (j < n), dst[t] = 0 for t >= base + n.  `base`, `n` wave-uniform.
__device__ __forceinline__ void place_tiles(v4f (&dst)[MT], const int base, const v4f (&src)[5], const int n) {
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    if (t >= base) {
      v4f v = V4ZERO;
#pragma unroll
      for (int j = 0; j < 5; ++j)
        if (j < n && t == base + j) v = src[j];
      dst[t] = v;
    }
  }
}

// dst[j] += src[base + j] (j < n)
__device__ __forceinline__ void gather_tiles(v4f (&dst)[5], const v4f (&src)[MT], const int base, const int n) {
#pragma unroll
  for (int t = 0; t < MT; ++t) {
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (j < n && t == base + j) dst[j] += src[t];
  }
}

// ---------------------------------------------------------------------------
// Positional encoding in slot layout (reference: code/model/embedder.py:10-50).
// PE index j: j < 3 -> x_j ; else m = j - 3, octave k = m / 6, (m % 6) < 3 -> sin(2^k x_c),
// else cos(2^k x_c), c = m % 3.  A lane owns slots 16 t + 4 q + r, t < 3.
// Stateless on purpose (recomputed where needed) to keep registers for the tiles.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void pe_slot(const int j, const int n_valid, const float x0, const float x1,
                                        const float x2, float& val, float& dval, int& c) {
  val = 0.f; dval = 0.f; c = 3;
  if (j < 3) {
    c = j;
    val = (c == 0) ? x0 : (c == 1) ? x1 : x2;
    dval = 1.f;
  } else if (j < n_valid) {
    const int m = j - 3;
    const int k = m / 6;
    const int w = m - 6 * k;
    c = (w >= 3) ? w - 3 : w;
    const float xc = (c == 0) ? x0 : (c == 1) ? x1 : x2;
    const float f = (float)(1 << k);
    float sn, cs;
    sincosf(xc * f, &sn, &cs);
    if (w < 3) { val = sn; dval = f * cs; } else { val = cs; dval = -f * sn; }
  }
}

// e[t][r] for t < 3
__device__ __forceinline__ void pe_values(v4f (&e)[5], const float x0, const float x1, const float x2,
                                          const int n_freqs) {
  const int q = lane_id() >> 4;
  const int n_valid = 3 + 6 * n_freqs;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v, dv; int c;
      pe_slot(16 * t + 4 * q + r, n_valid, x0, x1, x2, v, dv, c);
      e[t][r] = v;
    }
  }
}

// n_c = sum_j r_j de_j [c(j) == c], reduced over the 4 quarter lanes of the point
__device__ __forceinline__ void pe_jacobian_transpose(const v4f (&r)[5], const float x0, const float x1,
                                                      const float x2, const int n_freqs, float& n0, float& n1,
                                                      float& n2) {
  const int q = lane_id() >> 4;
  const int n_valid = 3 + 6 * n_freqs;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v, dv; int c;
      pe_slot(16 * t + 4 * q + k, n_valid, x0, x1, x2, v, dv, c);
      const float w = r[t][k] * dv;
      a0 += (c == 0) ? w : 0.f;
      a1 += (c == 1) ? w : 0.f;
      a2 += (c == 2) ? w : 0.f;
    }
  }
  n0 = sum_over_quarters(a0);
  n1 = sum_over_quarters(a1);
  n2 = sum_over_quarters(a2);
}

// rbar_j = de_j * nbar_{c(j)}  for the PE tiles (t < 3)
__device__ __forceinline__ void pe_jacobian(v4f (&rbar)[5], const float x0, const float x1, const float x2,
                                            const int n_freqs, const float g0, const float g1, const float g2) {
  const int q = lane_id() >> 4;
  const int n_valid = 3 + 6 * n_freqs;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v, dv; int c;
      pe_slot(16 * t + 4 * q + k, n_valid, x0, x1, x2, v, dv, c);
      const float g = (c == 0) ? g0 : (c == 1) ? g1 : (c == 2) ? g2 : 0.f;
      rbar[t][k] = dv * g;
    }
  }
}


// ---------------------------------------------------------------------------
// The matrix core as the kernel bodies see it (sdf_kernels.h / color_kernels.h are templated on this).
// gemm(): acc[0..OT) += W * in with the product's hooks (see NoHooks) called per pair of out tiles.
// ---------------------------------------------------------------------------
struct CoreF32 {
  typedef v4f wvec;                                  // one 16-byte element of the weight pack
  // the products that carry a bias add it to the finished tiles (BiasHooks) instead of starting the accumulators from
  // it: the forward + gradient kernel spilled 208 bytes per lane with 17 bias tiles live beside two activation vectors,
  // 36 now (2.155 -> 2.08 ms, same box), the colour forward kernel 0.339 -> 0.332 ms
  static constexpr bool BIAS_IN_HOOKS = true;
  // hash-grid features and their gradients as the encoder's level-major tensors, the encoder's Jacobian applied in the
  // kernels (sdf_kernels.h: AuxView)
#ifndef MLP_F32_AUX_LEVEL_MAJOR
#define MLP_F32_AUX_LEVEL_MAJOR true     // timing experiments build a variant with false (rows + transposes, round 3's kernels)
#endif
  static constexpr bool AUX_LEVEL_MAJOR = MLP_F32_AUX_LEVEL_MAJOR;
  static __device__ __forceinline__ float softplus(const float a) {
    float h, s;
    softplus100(a, h, s);
    return h;
  }
  template <class Hooks>
  static __device__ __forceinline__ void gemm(const int kp, v4f (&acc)[MT], const v4f (&in)[MT], const int OT,
                                              const wvec* __restrict__ wsrc, void* lds, Hooks&& hk) {
    gemm_dispatch(kp, acc, in, OT, wsrc, (v4f*)lds, hk);
  }
  template <class Hooks>
  static __device__ __forceinline__ void gemm_bias(const int kp, v4f (&acc)[MT], const v4f (&in)[MT], const int OT,
                                                   const wvec* __restrict__ wsrc, void* lds, Hooks&& hk,
                                                   const float* __restrict__ bias) {
    zero_tiles(acc);
    BiasHooks<typename std::remove_reference<Hooks>::type> bh(hk, bias, OT);
    gemm_dispatch(kp, acc, in, OT, wsrc, (v4f*)lds, bh);
  }
};
