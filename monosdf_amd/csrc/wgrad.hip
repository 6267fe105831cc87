// Weight-gradient GEMMs for the fused MLPs: D[wx x wy] = sum over points of X[p][:]^T Y[p][:].
//
// The contraction runs over points (K ~ 1e5), so one workgroup keeps a whole
// <=256x256 fp32 result in registers (8 waves x 128 accumulator VGPRs,
// v_mfma_f32_32x32x2_f32) and streams its share of the points HBM -> LDS (LDS-DMA,
// double buffered) -> one ds_read_b32 per operand fragment.  Each split writes a
// partial block; msdf_reduce_kernel sums the partials in a fixed order (bitwise
// reproducible, no float atomics) and scatters them back to the original
// [out][in] weight layout.
//
// Replaces the `mm` calls autograd emits for dW in the reference's backward
// (implied by loss.backward(), code/training/monosdf_train.py:431).
#include "common.h"

typedef float v16f __attribute__((ext_vector_type(16)));

#define WG_THREADS 512
#define WG_NP 32                               // points per LDS stage: one barrier per 128 MFMAs of each wave
#define WG_NBUF 2                              // double buffered (a stage is ~8 us of matrix products)
#define WG_TILE_F (WG_NP * 256)                // floats per operand tile
#define WG_STAGE_F (2 * WG_TILE_F + 64)        // X tile, Y tile, v[NP] (+pad)
#define WG_LDS_BYTES (WG_NBUF * WG_STAGE_F * 4)
#define WG_PIECES (WG_NP * 256 / 4 / 64 / 8)    // 1 KB LDS-DMA pieces of one operand tile per wave: 4
#define WG_GLDS_PER_STAGE (2 * WG_PIECES + 1)  // LDS-DMA instructions every wave issues per stage (uniform!)

// Copy [WG_NP x w] (row pitch ld) into a dense LDS image with exactly WG_PIECES 1 KB LDS-DMA pieces per wave;
// a wave whose piece index runs past the tile re-copies the last piece (same bytes, same place), which
// keeps the per-wave instruction count uniform so that a counted s_waitcnt vmcnt(N) is exact.
__device__ __forceinline__ void wg_issue_tile(const float* __restrict__ src, const int ld, const int w,
                                              float* dst, const int wave, const int lane) {
  const int w4 = w >> 2;
  const int pieces = (WG_NP * w4) >> 6;        // >= 2 (w multiple of 16)
#pragma unroll
  for (int i = 0; i < WG_PIECES; ++i) {
    const int piece = min(wave + 8 * i, pieces - 1);
    const int u = piece * 64 + lane;
    const int row = u / w4, c4 = u - row * w4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)row * ld + 4 * c4),
                                     (__attribute__((address_space(3))) void*)(dst + piece * 256), 16, 0, 0);
  }
}

// The same number of LDS-DMA instructions as wg_issue_tile, all of them the tile's first piece (same bytes, same
// place, served by the L1): what an item without a Y operand issues instead of a second copy of its X tile.
__device__ __forceinline__ void wg_issue_first_piece(const float* __restrict__ src, const int ld, const int w,
                                                     float* dst, const int lane) {
  const int w4 = w >> 2;
  const int row = lane / w4, c4 = lane - row * w4;
#pragma unroll
  for (int i = 0; i < WG_PIECES; ++i)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)row * ld + 4 * c4),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

__device__ __forceinline__ void wg_wait_outstanding(const int stages_behind) {
  // wait until at most `stages_behind` later stages are still in flight
  static_assert(WG_GLDS_PER_STAGE == 9 && WG_NBUF == 2, "the immediates below are WG_GLDS_PER_STAGE multiples");
  switch (stages_behind) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
  }
}

// The wave grid is a template parameter, not a branch inside the k loop: with the branch there, the accumulators of
// the two variants met in phi nodes and every k step copied 32 accumulator registers behind an `s_nop 15`
// that waited out the previous MFMA (a third of the matrix pipe's time).
// NBN == 0: wide items, 2 x 4 waves of 128 rows x 64 columns; NBN = 2, 3, 4: items of <= 32 NBN columns (the PE /
// PE + hash-feature / skip blocks), 8 x 1 waves of 32 rows x 32 NBN columns so that all four SIMDs work -- an
// 80-column item on the wide grid keeps the waves of two SIMDs idle and takes as long as a 256-column one.
// NBN == -1: items of <= 32 rows, 1 x 8 waves of 32 rows x 32 columns.
template <int NBN>
__device__ __forceinline__ void wgrad_body(const msdf_wgrad_item_t& it, const int split,
                                           float* __restrict__ part, const int P_pad, float* lds_f,
                                           const float* __restrict__ base0, const float* __restrict__ base1) {
  const int n_splits = it.n_splits;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wide items: 2 x 4 waves, 128 rows x 64 cols each (4 x 2 MFMA tiles per wave);
  // narrow items (wy <= 32 NBN): 8 x 1 waves, 32 rows x 32 NBN cols each, so all 8 waves (all 4 SIMDs) work
  // thin items (wx <= 32 rows, e.g. the colour network's 3-row output layer): 1 x 8 waves, 32 rows x 32 cols each
  constexpr bool narrow = NBN > 0;
  constexpr bool thin = NBN < 0;
  const int wi = thin ? 0 : narrow ? wave : (wave >> 2), wj = thin ? wave : narrow ? 0 : (wave & 3);
  const int i_base = narrow ? 32 * wi : 128 * wi, j_base = thin ? 32 * wj : 64 * wj;
  constexpr int na = (narrow || thin) ? 1 : 4;
  constexpr int nb = thin ? 1 : narrow ? NBN : 2;

  // point range of this split, in stages of WG_NP points
  const int n_stages_total = P_pad / WG_NP;
  const int per = (n_stages_total + n_splits - 1) / n_splits;
  const int s_begin = split * per;
  const int s_end = min(n_stages_total, s_begin + per);
  const int n_st = max(0, s_end - s_begin);

  const float* X = ((it.bufs & 0xff) ? base1 : base0) + it.x;
  const float* Y = (((it.bufs >> 8) & 0xff) ? base1 : base0) + it.y;
  const float* V = (((it.bufs >> 16) & 0xff) == 0xff) ? nullptr : ((((it.bufs >> 16) & 0xff) ? base1 : base0) + it.v);
  const bool do_mm = it.wy > 0;

  v16f acc[na][nb];
#pragma unroll
  for (int a = 0; a < na; ++a)
#pragma unroll
    for (int b = 0; b < nb; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float colsum = 0.f, vrow = 0.f;

  // which of this wave's tiles are inside [wx x wy]
  bool ai[na], bj[nb];
#pragma unroll
  for (int a = 0; a < na; ++a) ai[a] = (i_base + 32 * a) < it.wx;
#pragma unroll
  for (int b = 0; b < nb; ++b) bj[b] = (j_base + 32 * b) < it.wy;
  const bool wave_active = do_mm && ai[0] && bj[0];

  // every wave issues exactly WG_GLDS_PER_STAGE LDS-DMA instructions per stage
  auto issue = [&](int j) {
    float* base = lds_f + (j & (WG_NBUF - 1)) * WG_STAGE_F;
    const size_t p0 = (size_t)(s_begin + j) * WG_NP;
    wg_issue_tile(X + p0 * it.x_ld, it.x_ld, it.wx, base, wave, lane);
    if (do_mm) wg_issue_tile(Y + p0 * it.y_ld, it.y_ld, it.wy, base + WG_TILE_F, wave, lane);
    else wg_issue_first_piece(X + p0 * it.x_ld, it.x_ld, it.wx, base, lane);   // column sums only: keeps the count uniform
    // per-point scalar (or, without one, a re-copy of 64 floats of X): one 4-byte-per-lane piece
    const float* vsrc = (V != nullptr) ? V + p0 + min(lane, WG_NP - 1) : X + p0 * it.x_ld + min(lane, 15);
    float* vdst = (V != nullptr) ? base + 2 * WG_TILE_F : base + 2 * WG_TILE_F;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)vsrc,
                                     (__attribute__((address_space(3))) void*)vdst, 4, 0, 0);
  };

  // prologue: WG_NBUF - 1 stages in flight
#pragma unroll
  for (int j = 0; j < WG_NBUF - 1; ++j)
    if (j < n_st) issue(j);

  for (int j = 0; j < n_st; ++j) {
    // stage j has landed once at most min(WG_NBUF - 2, n_st-1-j) later stages are outstanding
    wg_wait_outstanding(min(WG_NBUF - 2, n_st - 1 - j));
    __builtin_amdgcn_s_barrier();              // every wave's pieces of stage j are in LDS; stage j-1 is fully consumed
    if (j + WG_NBUF - 1 < n_st) issue(j + WG_NBUF - 1);   // refill the buffer stage j-1 used
    const float* xt = lds_f + (j & (WG_NBUF - 1)) * WG_STAGE_F;
    const float* yt = xt + WG_TILE_F;
    const float* vt = xt + 2 * WG_TILE_F;
    if (wave_active) {
      const float* xa = xt + (lane >> 5) * it.wx + i_base + (lane & 31);
      const float* yb = yt + (lane >> 5) * it.wy + j_base + (lane & 31);
      // software-pipelined: the fragments of k-step k+1 are in flight while the 8 MFMAs of step k issue
      float af[2][na], bf[2][nb];
#pragma unroll
      for (int a = 0; a < na; ++a) af[0][a] = xa[32 * a];
#pragma unroll
      for (int b = 0; b < nb; ++b) bf[0][b] = yb[32 * b];
#pragma unroll
      for (int k = 0; k < WG_NP / 2; ++k) {
        const int cur = k & 1, nxt = cur ^ 1;
        if (k + 1 < WG_NP / 2) {
#pragma unroll
          for (int a = 0; a < na; ++a) af[nxt][a] = xa[2 * (k + 1) * it.wx + 32 * a];
#pragma unroll
          for (int b = 0; b < nb; ++b) bf[nxt][b] = yb[2 * (k + 1) * it.wy + 32 * b];
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch above the MFMAs (the scheduler would sink it)
#pragma unroll
        for (int a = 0; a < na; ++a)
#pragma unroll
          for (int b = 0; b < nb; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][a], bf[cur][b], acc[a][b], 0, 0, 0);
      }
    }
    if (it.colsum_off >= 0 && tid < it.wx) {
#pragma unroll 8
      for (int p = 0; p < WG_NP; ++p) colsum += xt[p * it.wx + tid];
    }
    if (it.vrow_off >= 0 && tid >= 256 && (tid - 256) < it.wy) {
#pragma unroll 8
      for (int p = 0; p < WG_NP; ++p) vrow += vt[p] * yt[p * it.wy + (tid - 256)];
    }
  }

  // ---------------- write this split's partials ----------------
  if (wave_active) {
    float* out = part + it.part_off + (size_t)split * it.wx * it.wy;
#pragma unroll
    for (int a = 0; a < na; ++a) {
#pragma unroll
      for (int b = 0; b < nb; ++b) {
        if (ai[a] && bj[b]) {
          const int j = j_base + 32 * b + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = i_base + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (i < it.wx && j < it.wy) out[(size_t)i * it.wy + j] = acc[a][b][r];
          }
        }
      }
    }
  }
  if (it.colsum_off >= 0 && tid < it.wx) part[it.colsum_off + (size_t)split * it.wx + tid] = colsum;
  if (it.vrow_off >= 0 && tid >= 256 && (tid - 256) < it.wy)
    part[it.vrow_off + (size_t)split * it.wy + (tid - 256)] = vrow;
}

__global__ void __launch_bounds__(WG_THREADS, 2)
msdf_wgrad_k(const msdf_wgrad_item_t* __restrict__ items, const int* __restrict__ wg_map,
             float* __restrict__ part, const int P_pad, const float* __restrict__ base0,
             const float* __restrict__ base1) {
  extern __shared__ float lds_f[];
  const msdf_wgrad_item_t it = items[wg_map[2 * blockIdx.x]];
  const int split = wg_map[2 * blockIdx.x + 1];
  // wide items: 2 x 4 waves of 128 x 64; items of <= 128 columns: 8 x 1 waves of 32 x (64 | 96 | 128)
  if (it.wx <= 32 && it.wy > 32) wgrad_body<-1>(it, split, part, P_pad, lds_f, base0, base1);
  else if (it.wy <= 64) wgrad_body<2>(it, split, part, P_pad, lds_f, base0, base1);
  else if (it.wy <= 96) wgrad_body<3>(it, split, part, P_pad, lds_f, base0, base1);
  else if (it.wy <= 128) wgrad_body<4>(it, split, part, P_pad, lds_f, base0, base1);
  else wgrad_body<0>(it, split, part, P_pad, lds_f, base0, base1);
}


// ---------------------------------------------------------------------------
// bf16x3 variant: the same items on v_mfma_f32_16x16x32_bf16 (x*y ~ x_hi*y_hi + x_hi*y_lo + x_lo*y_hi).
//
// Both operands of this GEMM are activations (K = points), so the hi/lo split cannot be precomputed the
// way the weight packs are.  Each element is split ONCE per workgroup: a thread loads the 8 points x 1 slot
// an MFMA lane needs (row-wise coalesced dword loads, issued one stage ahead), splits them, and writes the
// two 16-byte fragments-to-be into a fragment-ordered LDS image; every wave then fetches its A / B fragments
// with single ds_read_b128s.  Image of one stage (32 points): [X | Y][16 slot tiles][hi | lo][64 lanes] x 16 B
// = 64 KB, double buffered; lane = 16 * (point octet) + (slot & 15), element j = point 8 * octet + j.
// ---------------------------------------------------------------------------
typedef __bf16 wv8bf __attribute__((ext_vector_type(8)));
typedef float wv4f __attribute__((ext_vector_type(4)));

#define WB_NP 32
#define WB_IMG_V8 (2 * 16 * 2 * 64)              // 16-byte elements per stage image
#define WB_LDS_BYTES (2 * WB_IMG_V8 * 16)        // 128 KB

__device__ __forceinline__ void wb_split8(const float (&v)[8], wv8bf& hi, wv8bf& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

// NARROW is a template parameter and tiles are never skipped inside the stage loop (slots past wx / wy are
// zero in the image; only the final store is guarded): straight-line MFMAs instead of a scalar branch per tile.
template <bool NARROW>
__device__ __forceinline__ void wgrad_b16_body(const msdf_wgrad_item_t& it, const int split,
                                               float* __restrict__ part,
                                               const int P_pad, wv8bf* lds_img,
                                               const float* __restrict__ base0, const float* __restrict__ base1) {
  const int n_splits = it.n_splits;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wide items: 2 x 4 waves of 128 rows x 64 cols (8 x 4 tiles); narrow (wy <= 64): 8 x 1 waves of 32 x 64
  constexpr bool narrow = NARROW;
  const int wi = narrow ? wave : (wave >> 2), wj = narrow ? 0 : (wave & 3);
  const int i_base = narrow ? 32 * wi : 128 * wi, j_base = 64 * wj;
  constexpr int na = narrow ? 2 : 8;

  const int n_stages_total = P_pad / WB_NP;
  const int per = (n_stages_total + n_splits - 1) / n_splits;
  const int s_begin = split * per;
  const int s_end = min(n_stages_total, s_begin + per);
  const int n_st = max(0, s_end - s_begin);

  const float* X = ((it.bufs & 0xff) ? base1 : base0) + it.x;
  const float* Y = (((it.bufs >> 8) & 0xff) ? base1 : base0) + it.y;
  const float* V = (((it.bufs >> 16) & 0xff) == 0xff) ? nullptr : ((((it.bufs >> 16) & 0xff) ? base1 : base0) + it.v);
  const bool do_mm = it.wy > 0;
  const bool want_vrow = it.vrow_off >= 0;

  wv4f acc[na][4];
#pragma unroll
  for (int a = 0; a < na; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (wv4f){0.f, 0.f, 0.f, 0.f};
  bool bj[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) bj[b] = (j_base + 16 * b) < it.wy;
  const bool wave_active = do_mm && i_base < it.wx && bj[0];

  // loader role: slot `ls`, point octets kh and kh + 2 of both operands
  const int ls = tid & 255, kh = tid >> 8;
  const bool lx = ls < it.wx, ly = do_mm && ls < it.wy;
  float xr[2][8], yr[2][8];
  float colsum = 0.f, vrow = 0.f;

  auto load = [&](const int j) {
    const size_t p0 = (size_t)(s_begin + j) * WB_NP;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const size_t pu = p0 + 8 * (kh + 2 * u);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xr[u][e] = lx ? X[(pu + e) * it.x_ld + ls] : 0.f;
        yr[u][e] = ly ? Y[(pu + e) * it.y_ld + ls] : 0.f;
      }
    }
  };
  auto store = [&](const int j) {
    wv8bf* img = lds_img + (j & 1) * WB_IMG_V8;
    const size_t p0 = (size_t)(s_begin + j) * WB_NP;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int kq = kh + 2 * u;
      const int fl = (ls >> 4) * 128 + kq * 16 + (ls & 15);       // [tile][hi|lo][lane]
      wv8bf hi, lo;
      wb_split8(xr[u], hi, lo);
      img[fl] = hi;
      img[fl + 64] = lo;
      if (do_mm) {
        wb_split8(yr[u], hi, lo);
        img[16 * 128 + fl] = hi;
        img[16 * 128 + fl + 64] = lo;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) colsum += xr[u][e];
      if (want_vrow) {
        const wv4f v0 = *(const wv4f*)(V + p0 + 8 * kq), v1 = *(const wv4f*)(V + p0 + 8 * kq + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) vrow += v0[e] * yr[u][e] + v1[e] * yr[u][4 + e];
      }
    }
  };

  if (n_st > 0) load(0);
  for (int j = 0; j < n_st; ++j) {
    store(j);
    if (j + 1 < n_st) load(j + 1);          // in flight behind this stage's matrix products
    __syncthreads();                        // image j complete; image j-1's readers passed the previous barrier
    if (wave_active) {
      const wv8bf* img = lds_img + (j & 1) * WB_IMG_V8 + lane;
      wv8bf bh[4], bl[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int tn = (j_base >> 4) + b;
        bh[b] = img[16 * 128 + tn * 128];
        bl[b] = img[16 * 128 + tn * 128 + 64];
      }
      wv8bf ah = img[(i_base >> 4) * 128], al = img[(i_base >> 4) * 128 + 64];
#pragma unroll
      for (int a = 0; a < na; ++a) {
        wv8bf nh = ah, nl = al;
        if (a + 1 < na) {
          nh = img[((i_base >> 4) + a + 1) * 128];
          nl = img[((i_base >> 4) + a + 1) * 128 + 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[b], acc[a][b], 0, 0, 0);
        }
        ah = nh;
        al = nl;
      }
    }
  }

  // ---------------- write this split's partials ----------------
  if (wave_active) {
    float* out = part + it.part_off + (size_t)split * it.wx * it.wy;
#pragma unroll
    for (int a = 0; a < na; ++a) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (bj[b]) {
          const int n = j_base + 16 * b + (lane & 15);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = i_base + 16 * a + 4 * (lane >> 4) + r;
            if (m < it.wx && n < it.wy) out[(size_t)m * it.wy + n] = acc[a][b][r];
          }
        }
      }
    }
  }
  // column sums / v-weighted sums: two loader threads per slot, added in a fixed order through LDS
  if (it.colsum_off >= 0 || want_vrow) {
    __syncthreads();
    float* red = (float*)lds_img;
    red[kh * 256 + ls] = colsum;
    red[512 + kh * 256 + ls] = vrow;
    __syncthreads();
    if (tid < 256) {
      if (it.colsum_off >= 0 && tid < it.wx)
        part[it.colsum_off + (size_t)split * it.wx + tid] = red[tid] + red[256 + tid];
      if (want_vrow && tid < it.wy)
        part[it.vrow_off + (size_t)split * it.wy + tid] = red[512 + tid] + red[768 + tid];
    }
  }
}

__global__ void __launch_bounds__(WG_THREADS, 2)
msdf_wgrad_b16_k(const msdf_wgrad_item_t* __restrict__ items, const int* __restrict__ wg_map,
                 float* __restrict__ part, const int P_pad, const float* __restrict__ base0,
                 const float* __restrict__ base1) {
  extern __shared__ wv8bf lds_img[];
  const msdf_wgrad_item_t it = items[wg_map[2 * blockIdx.x]];
  const int split = wg_map[2 * blockIdx.x + 1];
  if (it.wy <= 64) wgrad_b16_body<true>(it, split, part, P_pad, lds_img, base0, base1);
  else wgrad_b16_body<false>(it, split, part, P_pad, lds_img, base0, base1);
}


// dst[rowmap[i]*ld + colmap[j]] = scale * sum_b PART[b][i][j]   (fixed summation order: b ascending, so the
// result is bitwise reproducible).  Four consecutive elements per thread (16-byte loads; wx*wy is a multiple
// of 16) and the block loop unrolled so that 8 loads are in flight before the first add.
__global__ void __launch_bounds__(256)
msdf_reduce_k(const msdf_reduce_rule_t* __restrict__ rules, const int* __restrict__ maps,
                   const float* __restrict__ part, float* __restrict__ dst) {
  typedef float rv4f __attribute__((ext_vector_type(4)));
  const msdf_reduce_rule_t R = rules[blockIdx.y];
  const int n = R.wx * R.wy;
  const int n4 = n >> 2;
  const rv4f* src = (const rv4f*)(part + R.part_off);
  for (int e4 = blockIdx.x * blockDim.x + threadIdx.x; e4 < n4; e4 += gridDim.x * blockDim.x) {
    rv4f s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int b = 0; b < R.n_blocks; ++b) s += src[(size_t)b * n4 + e4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = 4 * e4 + c;
      const int i = e / R.wy, jj = e - i * R.wy;
      const int row = (R.rowmap_off >= 0) ? maps[R.rowmap_off + i] : R.fixed_row;
      const int col = (R.colmap_off >= 0) ? maps[R.colmap_off + jj] : 0;
      if (row >= 0 && col >= 0) dst[R.dst_off + (size_t)row * R.dst_ld + col] = R.scale * s[c];
    }
  }
}

extern "C" int msdf_wgrad(const msdf_wgrad_item_t* items_dev, const int32_t* wg_map_dev, int n_wgs,
                          float* partials, int P_pad, int precision, const float* base0, const float* base1,
                          void* stream) {
  if (n_wgs < 0 || P_pad < 0 || (P_pad % WB_NP) != 0) return MSDF_ERR_ARG;
  if (n_wgs > 0 && P_pad > 0 && base0 == nullptr) return MSDF_ERR_ARG;
  if (n_wgs == 0 || P_pad == 0) return MSDF_OK;
  if (precision == MSDF_PRECISION_BF16X3) {
    if (hipFuncSetAttribute((const void*)msdf_wgrad_b16_k, hipFuncAttributeMaxDynamicSharedMemorySize,
                            WB_LDS_BYTES) != hipSuccess)
      return MSDF_ERR_LAUNCH;
    msdf_wgrad_b16_k<<<n_wgs, WG_THREADS, WB_LDS_BYTES, (hipStream_t)stream>>>(items_dev, wg_map_dev, partials,
                                                                               P_pad, base0, base1);
    return msdf_check_launch();
  }
  // BF16X6 networks take the fp32 kernel: both operands are saved fp32 activations, a three-plane image of a stage
  // would not fit the LDS double buffer, and the fp32 matrix instructions are exact where bf16x6 is fp32-grade
  if (precision != MSDF_PRECISION_F32 && precision != MSDF_PRECISION_BF16X6) return MSDF_ERR_ARG;
  if (hipFuncSetAttribute((const void*)msdf_wgrad_k, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS_BYTES) !=
      hipSuccess)
    return MSDF_ERR_LAUNCH;
  msdf_wgrad_k<<<n_wgs, WG_THREADS, WG_LDS_BYTES, (hipStream_t)stream>>>(items_dev, wg_map_dev, partials,
                                                                          P_pad, base0, base1);
  return msdf_check_launch();
}

extern "C" int msdf_reduce(const msdf_reduce_rule_t* rules_dev, int n_rules, const int* maps_dev,
                           const float* partials, float* dst, void* stream) {
  if (n_rules < 0) return MSDF_ERR_ARG;
  if (n_rules == 0) return MSDF_OK;
  const dim3 grid(64, n_rules);
  msdf_reduce_k<<<grid, 256, 0, (hipStream_t)stream>>>(rules_dev, maps_dev, partials, dst);
  return msdf_check_launch();
}
