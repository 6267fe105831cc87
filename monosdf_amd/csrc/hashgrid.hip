// Multi-resolution hash-grid encoder for gfx950: forward (+ dy/dx), grid backward,
// input backward and the two second-order kernels.
//
// Same arithmetic as the reference's CUDA kernels (code/hashencoder/src/hashencoder.cu:
// index function 35-72, kernel_grid 103-254, kernel_grid_backward 257-343,
// kernel_input_backward 346-372, kernel_grid_second_backward_grad 375-428,
// kernel_grid_second_backward_embedding 431-595), written for wave64: one lane per
// (point, level), level-major launch so that a level's table slice (<= 4 MiB) is the
// working set of the blocks in flight, the 8 corner reads of a point reused for the
// output AND its three directional derivatives (the reference gathers them twice),
// [L,B,C] output rows written as 8-byte-per-lane coalesced stores.
// Layouts are the reference's: inputs [B,3] in [0,1], embeddings [n,C], offsets int32 [L+1],
// outputs [L,B,C], dy_dx [B, L*3*C].  fp32 only (the reference's fp16 path is dead code).
#include "common.h"
#include <cstdlib>

#define HG_THREADS 256

__device__ __forceinline__ uint32_t hg_index(const uint32_t px, const uint32_t py, const uint32_t pz,
                                             const uint32_t hashmap_size, const uint32_t resolution) {
  // dense while the running stride still fits the level, hashed otherwise (cu:54-72)
  uint32_t stride = 1, index = 0;
  const uint32_t p[3] = {px, py, pz};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (stride <= hashmap_size) {
      index += p[d] * stride;
      stride *= resolution;
    }
  }
  if (stride > hashmap_size) index = px ^ (py * 2654435761u) ^ (pz * 805459861u);
  return (index < hashmap_size) ? index : index % hashmap_size;     // same value; dense levels skip the division
}

struct HgCell {
  float scale;
  uint32_t res, hsize;
  uint32_t gx, gy, gz;
  float sx, sy, sz;       // smoothstep(frac)
  float dx, dy, dz;       // smoothstep'(frac)
  bool oob;
};

// per-level constants (scale, resolution, table size): uniform over a workgroup of the level-major launches.
// The index form of cu:54-72 is decided here, once per level instead of once per corner: `dense` = the running
// stride never exceeds the table (index = x + y res + z res^2), otherwise the xor hash; `mask` = hsize - 1 when
// the table size is a power of two (every hashed level of the reference's configurations: the modulo becomes an
// AND instead of a ~40-instruction integer division per corner).
struct HgLevel {
  float scale;
  uint32_t res, hsize;
  uint32_t s1, s2, mask;
  bool dense;
};
__device__ __forceinline__ HgLevel hg_level(const int* __restrict__ offsets, const uint32_t level, const float S,
                                            const uint32_t H) {
  HgLevel v;
  v.hsize = (uint32_t)(offsets[level + 1] - offsets[level]);
  // exp2f(level * S) as the correctly rounded float (device exp2f is only ~1 ulp; one ulp of scale moves a
  // fine-level sample by 1e-4 of a cell)
  v.scale = (float)exp2((double)((float)level * S)) * (float)H - 1.0f;
  v.res = (uint32_t)ceilf(v.scale) + 1u;
  // the loop of hg_index with the coordinates left out: which strides are taken, and whether the last one fits
  uint32_t stride = 1;
  v.s1 = v.s2 = 0;
  if (stride <= v.hsize) stride *= v.res;                        // d = 0 (stride 1)
  if (stride <= v.hsize) { v.s1 = stride; stride *= v.res; }     // d = 1
  if (stride <= v.hsize) { v.s2 = stride; stride *= v.res; }     // d = 2
  v.dense = !(stride > v.hsize);
  v.mask = ((v.hsize & (v.hsize - 1u)) == 0u) ? v.hsize - 1u : 0u;
  return v;
}

// hg_index with the level's decisions taken from HgLevel (same value for every input)
__device__ __forceinline__ uint32_t hg_index_lv(const HgLevel& lv, const uint32_t px, const uint32_t py,
                                                const uint32_t pz) {
  uint32_t index;
  if (lv.dense) index = px + py * lv.s1 + pz * lv.s2;
  else index = px ^ (py * 2654435761u) ^ (pz * 805459861u);
  if (lv.mask) return index & lv.mask;
  return (index < lv.hsize) ? index : index % lv.hsize;
}

__device__ __forceinline__ HgCell hg_locate_xyz(const float x, const float y, const float z, const HgLevel& lv) {
  HgCell c;
  c.oob = (x < 0.f || x > 1.f || y < 0.f || y > 1.f || z < 0.f || z > 1.f);
  c.hsize = lv.hsize;
  c.scale = lv.scale;
  c.res = lv.res;
  float px = x * c.scale, py = y * c.scale, pz = z * c.scale;
  const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
  c.gx = (uint32_t)fx; c.gy = (uint32_t)fy; c.gz = (uint32_t)fz;
  px -= fx; py -= fy; pz -= fz;
  c.dx = 6.f * px * (1.f - px); c.dy = 6.f * py * (1.f - py); c.dz = 6.f * pz * (1.f - pz);
  c.sx = px * px * (3.f - 2.f * px); c.sy = py * py * (3.f - 2.f * py); c.sz = pz * pz * (3.f - 2.f * pz);
  return c;
}

__device__ __forceinline__ HgCell hg_locate(const float* __restrict__ inputs, const HgLevel& lv, const uint32_t b) {
  return hg_locate_xyz(inputs[(size_t)b * 3 + 0], inputs[(size_t)b * 3 + 1], inputs[(size_t)b * 3 + 2], lv);
}

__device__ __forceinline__ HgCell hg_locate(const float* __restrict__ inputs, const int* __restrict__ offsets,
                                            const uint32_t b, const uint32_t level, const float S,
                                            const uint32_t H) {
  return hg_locate(inputs, hg_level(offsets, level, S, H), b);
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
// one (point, level): the 8 corner reads feed the output and its three directional derivatives
template <int C>
__device__ __forceinline__ void hg_forward_cell(const HgCell& c, const float* __restrict__ grid,
                                                const int* __restrict__ offsets, float* __restrict__ out,
                                                const HgLevel& lv, const uint32_t level, const int calc_grad_inputs,
                                                float* __restrict__ dy) {
  if (c.oob) {
#pragma unroll
    for (int ch = 0; ch < C; ++ch) out[ch] = 0.f;
    if (calc_grad_inputs) {
#pragma unroll
      for (int k = 0; k < 3 * C; ++k) dy[k] = 0.f;
    }
    return;
  }
  const float* table = grid + (size_t)(uint32_t)offsets[level] * C;
  // gather the 8 corners once
  float v[8][C];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t idx = hg_index_lv(lv, c.gx + (k & 1), c.gy + ((k >> 1) & 1), c.gz + ((k >> 2) & 1));
#pragma unroll
    for (int ch = 0; ch < C; ++ch) v[k][ch] = table[(size_t)idx * C + ch];
  }
  const float wx[2] = {1.f - c.sx, c.sx}, wy[2] = {1.f - c.sy, c.sy}, wz[2] = {1.f - c.sz, c.sz};
  float res[C];
#pragma unroll
  for (int ch = 0; ch < C; ++ch) res[ch] = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float w = wx[k & 1] * wy[(k >> 1) & 1] * wz[(k >> 2) & 1];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) res[ch] += w * v[k][ch];
  }
#pragma unroll
  for (int ch = 0; ch < C; ++ch) out[ch] = res[ch];
  if (calc_grad_inputs) {
    // d/dx: pairs differing in bit 0; weights of the other two axes; times scale * smoothstep'
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int a = m & 1, bq = (m >> 1) & 1;
        gx += (c.scale * wy[a] * wz[bq]) * (v[1 | (a << 1) | (bq << 2)][ch] - v[0 | (a << 1) | (bq << 2)][ch]) * c.dx;
        gy += (c.scale * wx[a] * wz[bq]) * (v[a | 2 | (bq << 2)][ch] - v[a | 0 | (bq << 2)][ch]) * c.dy;
        gz += (c.scale * wx[a] * wy[bq]) * (v[a | (bq << 1) | 4][ch] - v[a | (bq << 1) | 0][ch]) * c.dz;
      }
      dy[0 * C + ch] = gx;
      dy[1 * C + ch] = gy;
      dy[2 * C + ch] = gz;
    }
  }
}

template <int C>
__global__ void __launch_bounds__(HG_THREADS)
hg_forward_kernel(const float* __restrict__ inputs, const float* __restrict__ grid, const int* __restrict__ offsets,
                  float* __restrict__ outputs, const uint32_t B, const uint32_t L, const float S, const uint32_t H,
                  const int calc_grad_inputs, float* __restrict__ dy_dx) {
  const uint32_t b = blockIdx.x * HG_THREADS + threadIdx.x;
  if (b >= B) return;
  const uint32_t level = blockIdx.y;
  const HgLevel lv = hg_level(offsets, level, S, H);
  float* out = outputs + ((size_t)level * B + b) * C;
  // calc_grad_inputs == 2: dy_dx level-major [L, B, 3 C] (a wave writes 64 x 3 C contiguous floats) instead of the
  // reference's [B, L, 3 C] (3 C floats every L 3 C: partial lines written by 16 different launches' blocks)
  float* dy = (calc_grad_inputs == 2) ? dy_dx + ((size_t)level * B + b) * 3 * C
                                      : dy_dx + (size_t)b * 3 * L * C + (size_t)level * 3 * C;
  hg_forward_cell<C>(hg_locate(inputs, lv, b), grid, offsets, out, lv, level, calc_grad_inputs, dy);
}

// ---------------------------------------------------------------------------
// "Node" forms (ops.GridSdfFunction, the sampler's evaluations): the same arithmetic with the tensors laid out as
// the fused MLP kernels read and write them, and the elementwise steps around the encoder done here instead of as
// PyTorch launches:
//   * points arrive in world coordinates; x01 = (x * inv_divide + 1) * 0.5 is formed per lane exactly as the
//     module's tensor expression rounds it (a multiply by the reciprocal -- formed in double, rounded to fp32 --
//     an add, a multiply) and stored once;
//   * features, d sdf / d features and their gradients: pitch == 0 keeps the kernels' own level-major [L, B, C] layout
//     (coalesced: a wave touches 64 C contiguous floats); pitch > 0: point-major rows of `pitch` floats ([B, pitch],
//     level l channel c at column l C + c, columns >= L C zeroed), the layout of the fused SDF kernels' input tiles.
//     Point-major is free in the two small kernels (hg_node_input_gradient / hg_node_second_grad) and costly in the
//     two large ones (8-byte pieces at a 128-byte stride: forward 0.115 -> 0.148 ms per step, the scatter 0.120 ->
//     0.163), so ops.GridSdfFunction uses it for the former and msdf_hash_transpose (LDS tiles, both sides coalesced)
//     around the latter -- instead of five strided tensor copies of 11-48 us per training step.
// ---------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(HG_THREADS)
hg_node_forward_kernel(const float* __restrict__ x, const float inv_divide, float* __restrict__ x01_out,
                       const float* __restrict__ grid, const int* __restrict__ offsets, float* __restrict__ feat,
                       const uint32_t pitch, const uint32_t B, const uint32_t L, const float S, const uint32_t H,
                       float* __restrict__ dy_dx) {
  const uint32_t b = blockIdx.x * HG_THREADS + threadIdx.x;
  if (b >= B) return;
  const uint32_t level = blockIdx.y;
  const HgLevel lv = hg_level(offsets, level, S, H);
  const float u0 = (x[(size_t)b * 3 + 0] * inv_divide + 1.0f) * 0.5f;
  const float u1 = (x[(size_t)b * 3 + 1] * inv_divide + 1.0f) * 0.5f;
  const float u2 = (x[(size_t)b * 3 + 2] * inv_divide + 1.0f) * 0.5f;
  if (level == 0) {
    if (x01_out != nullptr) {
      x01_out[(size_t)b * 3 + 0] = u0; x01_out[(size_t)b * 3 + 1] = u1; x01_out[(size_t)b * 3 + 2] = u2;
    }
    for (uint32_t k = L * C; k < pitch; ++k) feat[(size_t)b * pitch + k] = 0.f;
  }
  float* dy = (dy_dx != nullptr) ? dy_dx + ((size_t)level * B + b) * 3 * C : nullptr;
  float* out = pitch ? feat + (size_t)b * pitch + level * C : feat + ((size_t)level * B + b) * C;
  hg_forward_cell<C>(hg_locate_xyz(u0, u1, u2, lv), grid, offsets, out, lv, level, dy != nullptr ? 2 : 0, dy);
}

// inout[b,d] += scale * sum_{l,c} g[b, l C + c] * dy_dx[l,b,d,c]   (the grid part of d sdf / d x, chain rule factor in)
// One thread per (point, input dimension) -- consecutive lanes read consecutive 8-byte (C = 2) pieces of dy_dx and write
// consecutive floats; the sum over (level, channel) runs in the reference's order (hashencoder.cu:347-372, also one thread per (b, d)).
template <int C>
__global__ void __launch_bounds__(HG_THREADS)
hg_node_input_gradient_kernel(const float* __restrict__ g, const uint32_t pitch, const float* __restrict__ dy_dx,
                              const uint32_t B, const uint32_t L, const float scale, float* __restrict__ inout) {
  const uint32_t t = blockIdx.x * HG_THREADS + threadIdx.x;
  if (t >= 3 * B) return;
  const uint32_t b = t / 3, d = t - 3 * b;
  float r = 0.f;
#pragma unroll 8
  for (uint32_t l = 0; l < L; ++l) {
    const float* dy = dy_dx + ((size_t)l * B + b) * 3 * C + d * C;
    const float* gp = pitch ? g + (size_t)b * pitch + l * C : g + ((size_t)l * B + b) * C;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) r += gp[ch] * dy[ch];
  }
  inout[t] += r * scale;      // the rounded product, then the sum (as the tensor expression nrm + through * k rounds them)
}

// gg[b] = scale * (b < n_split ? g_a[b] : g_b[b - n_split])  (a missing part is zero), stored once;
// grad_grad[b, l C + c] = sum_d gg[b,d] * dy_dx[l,b,d,c]
template <int C>
__global__ void __launch_bounds__(HG_THREADS)
hg_node_second_grad_kernel(const float* __restrict__ g_a, const float* __restrict__ g_b, const uint32_t n_split,
                           const float scale, float* __restrict__ gg_out, const float* __restrict__ dy_dx,
                           float* __restrict__ grad_grad, const uint32_t pitch, const uint32_t B, const uint32_t L) {
  const uint32_t b = blockIdx.x * HG_THREADS + threadIdx.x;
  if (b >= B) return;
  const uint32_t level = blockIdx.y;
  const float* src = (b < n_split) ? g_a : g_b;
  const size_t j = (b < n_split) ? b : b - n_split;
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (src != nullptr) { g0 = src[j * 3 + 0] * scale; g1 = src[j * 3 + 1] * scale; g2 = src[j * 3 + 2] * scale; }
  if (level == 0) {
    gg_out[(size_t)b * 3 + 0] = g0; gg_out[(size_t)b * 3 + 1] = g1; gg_out[(size_t)b * 3 + 2] = g2;
    for (uint32_t k = L * C; k < pitch; ++k) grad_grad[(size_t)b * pitch + k] = 0.f;
  }
  const float* dy = dy_dx + ((size_t)level * B + b) * 3 * C;
  float* out = pitch ? grad_grad + (size_t)b * pitch + level * C : grad_grad + ((size_t)level * B + b) * C;
#pragma unroll
  for (int ch = 0; ch < C; ++ch) out[ch] = g0 * dy[0 * C + ch] + g1 * dy[1 * C + ch] + g2 * dy[2 * C + ch];
}

// ---------------------------------------------------------------------------
// grid backward and second backward (embedding): scatter into the 8 corners (float atomics, like the
// reference; one code path for both kernels, they differ in the per-corner coefficient only).
// Lane mapping: C consecutive lanes own the C channels of ONE (point, level), so a wave-instruction
// adds 64/C entries of C contiguous floats each -- half (C=2) as many distinct memory segments per
// instruction as one-point-per-lane, which is what the memory-side atomic units are paced by
// (MI355X_MICROARCH.md "Global float atomics", access-shape row).
// ---------------------------------------------------------------------------
// value added to corner k of (point b, level, channel ch):
//   SECOND = false: w_k * grad                                         (kernel_grid_backward, cu:257-343)
//   SECOND = true : (sum over axes of +/- prod of the other axes' weights * gg_x * dsmooth * scale) * grad
//                                                                      (kernel_grid_second_backward_embedding, cu:431-595)
template <bool SECOND>
__device__ __forceinline__ void hg_corner_values(const HgCell& c, const float g, const float* __restrict__ gg_inputs,
                                                 const uint32_t b, float (&val)[8], uint32_t (&idx)[8]) {
  const float wx[2] = {1.f - c.sx, c.sx}, wy[2] = {1.f - c.sy, c.sy}, wz[2] = {1.f - c.sz, c.sz};
  float q0 = 0.f, q1 = 0.f, q2 = 0.f;
  if (SECOND) {
    q0 = gg_inputs[(size_t)b * 3 + 0] * c.dx * c.scale;
    q1 = gg_inputs[(size_t)b * 3 + 1] * c.dy * c.scale;
    q2 = gg_inputs[(size_t)b * 3 + 2] * c.dz * c.scale;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int bx = k & 1, by = (k >> 1) & 1, bz = (k >> 2) & 1;
    float w;
    if (SECOND)
      w = (bx ? 1.f : -1.f) * wy[by] * wz[bz] * q0 + (by ? 1.f : -1.f) * wx[bx] * wz[bz] * q1 +
          (bz ? 1.f : -1.f) * wx[bx] * wy[by] * q2;
    else
      w = wx[bx] * wy[by] * wz[bz];
    val[k] = w * g;
    idx[k] = hg_index(c.gx + bx, c.gy + by, c.gz + bz, c.hsize, c.res);
  }
}

// levels [level_base, level_base + gridDim.y): straight to memory
template <int C, bool SECOND>
__global__ void __launch_bounds__(HG_THREADS)
hg_scatter_kernel(const float* __restrict__ grad, const float* __restrict__ inputs, const int* __restrict__ offsets,
                  const float* __restrict__ gg_inputs, float* __restrict__ grad_grid, const uint32_t B,
                  const uint32_t level_base, const float S, const uint32_t H) {
  const uint32_t t = blockIdx.x * HG_THREADS + threadIdx.x;
  const uint32_t b = t / C, ch = t % C;
  if (b >= B) return;
  const uint32_t level = level_base + blockIdx.y;
  const HgCell c = hg_locate(inputs, offsets, b, level, S, H);
  if (c.oob) return;
  const float g = grad[((size_t)level * B + b) * C + ch];
  float* table = grad_grid + (size_t)(uint32_t)offsets[level] * C + ch;
  float val[8];
  uint32_t idx[8];
  hg_corner_values<SECOND>(c, g, gg_inputs, b, val, idx);
#pragma unroll
  for (int k = 0; k < 8; ++k) unsafeAtomicAdd(table + (size_t)idx[k] * C, val[k]);
}

// The coarsest levels (4,096 and 12,167 entries for the 16 -> 2048 pyramid) take every point of the batch, so
// hundreds of atomics land on each entry and serialise at the memory-side atomic units: alone, level 0 costs
// 5x and level 1 3x a fine level (scripts/diag_hash_levels.py).  Their whole table fits in LDS, so a workgroup
// accumulates its share of the points with LDS atomics and adds the non-zero entries to memory once.
#define HG_LDS_THREADS 1024
#define HG_LDS_WGS 128                     // workgroups per level
#define HG_LDS_MAX_BYTES (100 * 1024)

template <int C, bool SECOND>
__global__ void __launch_bounds__(HG_LDS_THREADS)
hg_scatter_lds_kernel(const float* __restrict__ grad, const float* __restrict__ inputs,
                      const int* __restrict__ offsets, const float* __restrict__ gg_inputs,
                      float* __restrict__ grad_grid, const uint32_t B, const float S, const uint32_t H,
                      const uint32_t lds_floats) {
  extern __shared__ float hg_tab[];
  const uint32_t level = blockIdx.y;
  const uint32_t hsize = (uint32_t)(offsets[level + 1] - offsets[level]);
  const uint32_t n = hsize * C;
  const bool in_lds = n <= lds_floats;       // host-side sizing is an estimate: fall back if the level is larger
  float* table = grad_grid + (size_t)(uint32_t)offsets[level] * C;
  // LDS atomics (~64 lanes per 32 cycles per CU, several same-address lanes per instruction on these levels) are
  // what bounds a workgroup, so the level is spread over many of them; each flushes only the entries its few
  // rays touched (the non-zero ones), which keeps the memory-side adds far below the direct count
  const uint32_t n_wg = gridDim.x;
  if (in_lds) {
    for (uint32_t i = threadIdx.x; i < n; i += HG_LDS_THREADS) hg_tab[i] = 0.f;
    __syncthreads();
  }
  const uint32_t per = (B + n_wg - 1) / n_wg;
  const uint32_t b_end = min(B, (blockIdx.x + 1) * per);
  const HgLevel lv = hg_level(offsets, level, S, H);
  for (uint32_t t = blockIdx.x * per * C + threadIdx.x; t < b_end * C; t += HG_LDS_THREADS) {
    const uint32_t b = t / C, ch = t % C;
    const HgCell c = hg_locate(inputs, lv, b);
    if (c.oob) continue;
    const float g = grad[((size_t)level * B + b) * C + ch];
    float val[8];
    uint32_t idx[8];
    hg_corner_values<SECOND>(c, g, gg_inputs, b, val, idx);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (in_lds) atomicAdd(&hg_tab[idx[k] * C + ch], val[k]);
      else unsafeAtomicAdd(table + (size_t)idx[k] * C + ch, val[k]);
    }
  }
  if (in_lds) {
    __syncthreads();
    // every workgroup starts its flush at a different entry, so they do not walk the table in lockstep
    const uint32_t start = (uint32_t)(((uint64_t)blockIdx.x * n) / n_wg);
    for (uint32_t i = threadIdx.x; i < n; i += HG_LDS_THREADS) {
      uint32_t e = i + start;
      e = (e >= n) ? e - n : e;
      const float v = hg_tab[e];
      if (v != 0.f) unsafeAtomicAdd(table + e, v);
    }
  }
}

// number of leading levels whose dense table fits the LDS budget, from the kernel's own resolution formula
static uint32_t hg_small_levels(const uint32_t C, const uint32_t L, const float S, const uint32_t H, uint32_t* lds_bytes) {
  uint32_t n = 0, bytes = 0;
  for (uint32_t l = 0; l < L; ++l) {
    const float scale = (float)exp2((double)((float)l * S)) * (float)H - 1.0f;
    const double res = (double)((uint32_t)ceilf(scale) + 1u);
    const double b = res * res * res * C * 4.0;
    if (b > (double)HG_LDS_MAX_BYTES) break;
    bytes = (uint32_t)b > bytes ? (uint32_t)b : bytes;
    ++n;
  }
  *lds_bytes = (bytes + 255u) & ~255u;
  return n;
}

template <int C, bool SECOND>
static int hg_launch_scatter(const float* grad, const float* inputs, const int* offsets, const float* gg_inputs,
                             float* grad_grid, const uint32_t B, const uint32_t L, const float S, const uint32_t H,
                             hipStream_t st) {
  uint32_t lds_bytes = 0;
  const uint32_t n_small = hg_small_levels(C, L, S, H, &lds_bytes);
  if (n_small > 0) {
    if (hipFuncSetAttribute((const void*)hg_scatter_lds_kernel<C, SECOND>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess)
      return MSDF_ERR_LAUNCH;
    hg_scatter_lds_kernel<C, SECOND><<<dim3(HG_LDS_WGS, n_small), HG_LDS_THREADS, lds_bytes, st>>>(
        grad, inputs, offsets, gg_inputs, grad_grid, B, S, H, lds_bytes / 4);
  }
  if (L > n_small) {
    const dim3 grid_c((B * C + HG_THREADS - 1) / HG_THREADS, L - n_small);   // one lane per (point, channel)
    hg_scatter_kernel<C, SECOND><<<grid_c, HG_THREADS, 0, st>>>(grad, inputs, offsets, gg_inputs, grad_grid, B, n_small,
                                                                S, H);
  }
  return MSDF_OK;
}

// ---------------------------------------------------------------------------
// Binned scatter: the same sums WITHOUT one memory-side atomic per corner.
//
// The direct kernels above issue B * L * 8 scattered float-atomic requests per call; the memory-side atomic units
// retire ~20 G requests/s whatever the schedule (MI355X_MICROARCH.md "Global float atomics": one 64-B request per
// distinct row), which is where they sit (0.59 ms for the 14 fine levels at B = 104,448).  Here every level's table
// is cut into slices of HB_SLICE_FLOATS floats (one LDS accumulator), and
//   hb_count_k       counts the corner contributions per (level, slice) bin      [LDS histogram per workgroup]
//   hb_scan_k        turns the counts into bin offsets and a list of work items  [one workgroup]
//   hb_place_k       writes each contribution as a record {entry in slice, C values} into its bin
//   hb_accumulate_k  one workgroup per bin (or per HB_CHUNK records of a crowded bin) sums its records into LDS
//                    with ds_add_f32 and adds the slice to the table once: plain read-modify-write when the
//                    bin has one workgroup, contiguous float atomics (the fast shape) when it has several.
// Coarse levels (hundreds of contributions per entry) and fine hashed levels (mostly unique entries) take the same
// path; the records are the only extra traffic (B * L * 8 * (4 + 4 C) bytes written once, read once).
// The sum order inside a bin follows the order in which workgroups reserved their runs: like the atomics it replaces
// it is not fixed from run to run (differences at fp32 rounding).
// ---------------------------------------------------------------------------
#ifndef HB_SLICE_FLOATS
#define HB_SLICE_FLOATS 8192
#endif
#define HB_CHUNK 8192
#define HB_THREADS 256
#define HB_PTS 4                      // points per thread in the count / place kernels (1,024 per workgroup)
#define HB_MAX_SLICES 1024            // per level, in the LDS histogram (2^19 entries x C = 8 -> 512)
#define HB_HDR_INTS 16

struct HbLayout {                      // int32 offsets into the workspace
  int nb_max, work_max, slice_base, bin_count, bin_base, bin_cursor, work, hdr_ints;
  size_t rec_off_bytes, total_bytes;
};
static HbLayout hb_layout(const uint32_t B, const uint32_t C, const uint32_t L, const uint64_t n_entries) {
  HbLayout y;
  y.nb_max = (int)((n_entries * C + HB_SLICE_FLOATS - 1) / HB_SLICE_FLOATS + L);
  y.work_max = y.nb_max + (int)(((uint64_t)B * L * 8 + HB_CHUNK - 1) / HB_CHUNK);
  y.slice_base = HB_HDR_INTS;
  y.bin_count = y.slice_base + (int)L + 1;
  y.bin_base = y.bin_count + y.nb_max;
  y.bin_cursor = y.bin_base + y.nb_max + 1;
  y.work = (y.bin_cursor + y.nb_max + 3) & ~3;          // int4 descriptors, 16-byte aligned
  y.hdr_ints = y.work + 4 * y.work_max;
  y.rec_off_bytes = (((size_t)y.hdr_ints * 4) + 255) & ~(size_t)255;
  y.total_bytes = y.rec_off_bytes + (size_t)B * L * 8 * (4 + 4 * C);
  return y;
}

__global__ void __launch_bounds__(HB_THREADS)
hb_setup_k(int* __restrict__ ws, const HbLayout y, const int* __restrict__ offsets, const uint32_t L, const uint32_t C) {
  for (int i = threadIdx.x; i < y.nb_max; i += HB_THREADS) ws[y.bin_count + i] = 0;
  if (threadIdx.x == 0) {
    const uint32_t epb = HB_SLICE_FLOATS / C;
    int base = 0;
    for (uint32_t l = 0; l < L; ++l) {
      ws[y.slice_base + l] = base;
      const uint32_t hsize = (uint32_t)(offsets[l + 1] - offsets[l]);
      base += (int)((hsize + epb - 1) / epb);
    }
    ws[y.slice_base + L] = base;
    ws[0] = base;                      // number of bins in use
  }
}

template <int C>
__global__ void __launch_bounds__(HB_THREADS)
hb_count_k(const float* __restrict__ inputs, const int* __restrict__ offsets, int* __restrict__ ws, const HbLayout y,
           const uint32_t B, const float S, const uint32_t H) {
  __shared__ int hist[HB_MAX_SLICES];
  const uint32_t level = blockIdx.y;
  const int sb = ws[y.slice_base + level];
  const int ns = ws[y.slice_base + level + 1] - sb;
  const bool local = ns <= HB_MAX_SLICES;
  if (local) {
    for (int i = threadIdx.x; i < ns; i += HB_THREADS) hist[i] = 0;
    __syncthreads();
  }
  constexpr uint32_t epb = HB_SLICE_FLOATS / C;
  const HgLevel lv = hg_level(offsets, level, S, H);
#pragma unroll
  for (int p = 0; p < HB_PTS; ++p) {
    const uint32_t b = (blockIdx.x * HB_PTS + p) * HB_THREADS + threadIdx.x;
    if (b >= B) continue;
    const HgCell c = hg_locate(inputs, lv, b);
    if (c.oob) continue;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t idx = hg_index(c.gx + (k & 1), c.gy + ((k >> 1) & 1), c.gz + ((k >> 2) & 1), c.hsize, c.res);
      const int s = (int)(idx / epb);
      if (local) atomicAdd(&hist[s], 1);
      else atomicAdd(&ws[y.bin_count + sb + s], 1);
    }
  }
  if (local) {
    __syncthreads();
    for (int i = threadIdx.x; i < ns; i += HB_THREADS) {
      const int n = hist[i];
      if (n) atomicAdd(&ws[y.bin_count + sb + i], n);
    }
  }
}

// exclusive scans over the bins: record offsets, and the list of work items for hb_accumulate_k -- one per
// HB_CHUNK records of a bin: {first record, end record, first float of the slice in the table, floats | shared flag}
__global__ void __launch_bounds__(1024)
hb_scan_k(int* __restrict__ ws, const HbLayout y, const int* __restrict__ offsets, const uint32_t L, const uint32_t C,
          const float S, const uint32_t H) {
  __shared__ int part[1024], partw[1024];
  const int nb = ws[0];
  const int t = threadIdx.x;
  const int per = (nb + 1023) / 1024;
  const int lo = min(nb, t * per), hi = min(nb, lo + per);
  int s = 0, w = 0;
  for (int i = lo; i < hi; ++i) {
    const int n = ws[y.bin_count + i];
    s += n;
    w += (n + HB_CHUNK - 1) / HB_CHUNK;
  }
  part[t] = s; partw[t] = w;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {            // Hillis-Steele inclusive scan of the per-thread totals
    const int a = (t >= d) ? part[t - d] : 0, aw = (t >= d) ? partw[t - d] : 0;
    __syncthreads();
    part[t] += a; partw[t] += aw;
    __syncthreads();
  }
  int run = part[t] - s, runw = partw[t] - w;
  const uint32_t epb = HB_SLICE_FLOATS / C;
  int level = 0;
  for (int i = lo; i < hi; ++i) {
    const int n = ws[y.bin_count + i];
    ws[y.bin_base + i] = run;
    ws[y.bin_cursor + i] = run;
    while (level + 1 < (int)L && ws[y.slice_base + level + 1] <= i) ++level;
    const uint32_t e0 = (uint32_t)(i - ws[y.slice_base + level]) * epb;
    const uint32_t hsize = (uint32_t)(offsets[level + 1] - offsets[level]);
    const int nf = (int)(min(epb, hsize - e0) * C);
    const int chunks = (n + HB_CHUNK - 1) / HB_CHUNK;
    for (int c = 0; c < chunks; ++c) {
      int* d = ws + y.work + 4 * (runw + c);
      d[0] = run + c * HB_CHUNK;
      d[1] = min(run + n, run + (c + 1) * HB_CHUNK);
      d[2] = (int)(((uint32_t)offsets[level] + e0) * C);   // < 2^31 floats: tables of up to 8 GB
      // bit 30: several workgroups share the slice; bit 29: hashed level (its records rarely repeat an entry)
      const uint64_t res = (uint64_t)ceilf((float)exp2((double)((float)level * S)) * (float)H - 1.0f) + 1u;
      const bool hashed = res * res * res > (uint64_t)hsize;
      d[3] = nf | (chunks > 1 ? (int)0x40000000 : 0) | (hashed ? (int)0x20000000 : 0);
    }
    run += n;
    runw += chunks;
  }
  if (t == 1023) {
    ws[y.bin_base + nb] = part[1023];
    ws[1] = partw[1023];               // number of work items
  }
}

// MODE 0: w_k * grad  (kernel_grid_backward);  MODE 1: second-order coefficient * grad
// (kernel_grid_second_backward_embedding);  MODE 2: w_k * grad + coefficient * grad2, both in one pass.
// A workgroup takes 1,024 points of one level: phase 1 ranks every corner inside (workgroup, bin) with an LDS
// histogram, one returning global atomic per non-empty bin then reserves the workgroup's run in the bin, phase 2
// writes the records.  (index in level | rank << 19) is all that is kept per corner between the phases.
template <int C, int MODE>
__global__ void __launch_bounds__(HB_THREADS)
hb_place_k(const float* __restrict__ grad, const float* __restrict__ grad2, const float* __restrict__ inputs,
           const int* __restrict__ offsets, const float* __restrict__ gg_inputs, int* __restrict__ ws,
           const HbLayout y, const uint32_t B, const float S, const uint32_t H) {
  __shared__ int hist[HB_MAX_SLICES];
  const uint32_t level = blockIdx.y;
  const int sb = ws[y.slice_base + level];
  const int ns = ws[y.slice_base + level + 1] - sb;
  const HgLevel lv = hg_level(offsets, level, S, H);
  // packed (index | rank << 19) needs index < 2^19 and rank < 2^13 (8 * 1,024 records per workgroup)
  const bool local = ns <= HB_MAX_SLICES && lv.hsize <= (1u << 19);
  if (local) {
    for (int i = threadIdx.x; i < ns; i += HB_THREADS) hist[i] = 0;
    __syncthreads();
  }
  constexpr uint32_t epb = HB_SLICE_FLOATS / C;
  uint32_t packed[HB_PTS][8];
  uint32_t live = 0;
#pragma unroll
  for (int p = 0; p < HB_PTS; ++p) {
    const uint32_t b = (blockIdx.x * HB_PTS + p) * HB_THREADS + threadIdx.x;
    if (b >= B) continue;
    const HgCell c = hg_locate(inputs, lv, b);
    if (c.oob) continue;
    live |= 1u << p;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t idx = hg_index(c.gx + (k & 1), c.gy + ((k >> 1) & 1), c.gz + ((k >> 2) & 1), c.hsize, c.res);
      const int s = (int)(idx / epb);
      if (local) packed[p][k] = idx | ((uint32_t)atomicAdd(&hist[s], 1) << 19);
      else packed[p][k] = (uint32_t)atomicAdd(&ws[y.bin_cursor + sb + s], 1);     // absolute record position
    }
  }
  if (local) {
    __syncthreads();
    for (int i = threadIdx.x; i < ns; i += HB_THREADS) {
      const int n = hist[i];
      hist[i] = n ? atomicAdd(&ws[y.bin_cursor + sb + i], n) : 0;       // start of this workgroup's run
    }
    __syncthreads();
  }
  uint32_t* rec = (uint32_t*)((char*)ws + y.rec_off_bytes);
#pragma unroll
  for (int p = 0; p < HB_PTS; ++p) {
    if (!(live & (1u << p))) continue;
    const uint32_t b = (blockIdx.x * HB_PTS + p) * HB_THREADS + threadIdx.x;
    const HgCell c = hg_locate(inputs, lv, b);
    const float wx[2] = {1.f - c.sx, c.sx}, wy[2] = {1.f - c.sy, c.sy}, wz[2] = {1.f - c.sz, c.sz};
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
    if (MODE != 0) {
      q0 = gg_inputs[(size_t)b * 3 + 0] * c.dx * c.scale;
      q1 = gg_inputs[(size_t)b * 3 + 1] * c.dy * c.scale;
      q2 = gg_inputs[(size_t)b * 3 + 2] * c.dz * c.scale;
    }
    float g1[C], g2[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      g1[ch] = grad[((size_t)level * B + b) * C + ch];
      g2[ch] = (MODE == 2) ? grad2[((size_t)level * B + b) * C + ch] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int bx = k & 1, by = (k >> 1) & 1, bz = (k >> 2) & 1;
      const float wk = wx[bx] * wy[by] * wz[bz];
      const float qk = (bx ? 1.f : -1.f) * wy[by] * wz[bz] * q0 + (by ? 1.f : -1.f) * wx[bx] * wz[bz] * q1 +
                       (bz ? 1.f : -1.f) * wx[bx] * wy[by] * q2;
      uint32_t idx, pos;
      if (local) {
        idx = packed[p][k] & ((1u << 19) - 1);
        pos = (uint32_t)hist[idx / epb] + (packed[p][k] >> 19);
      } else {
        idx = hg_index(c.gx + bx, c.gy + by, c.gz + bz, c.hsize, c.res);
        pos = packed[p][k];
      }
      uint32_t* r = rec + (size_t)pos * (1 + C);
      r[0] = idx % epb;
#pragma unroll
      for (int ch = 0; ch < C; ++ch) {
        float v;
        if (MODE == 0) v = wk * g1[ch];
        else if (MODE == 1) v = qk * g1[ch];
        else v = wk * g1[ch] + qk * g2[ch];
        r[1 + ch] = __float_as_uint(v);
      }
    }
  }
}

// acc += v in LDS as a compare-and-swap loop (for addresses that rarely collide)
__device__ __forceinline__ void lds_add_cas(float* p, const float v) {
  uint32_t* u = (uint32_t*)p;
  uint32_t old = *u;
  while (true) {
    // the sum goes through an opaque instruction: left visible, the compiler recognises the loop as an atomic float
    // add and turns it back into ds_add_f32
    float sum;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(sum) : "v"(__uint_as_float(old)), "v"(v));
    const uint32_t got = atomicCAS(u, old, __float_as_uint(sum));
    if (got == old) break;
    old = got;
  }
}

template <int C>
__global__ void __launch_bounds__(HB_THREADS)
hb_accumulate_k(const int* __restrict__ ws, const HbLayout y, float* __restrict__ grad_grid) {
  __shared__ float acc[HB_SLICE_FLOATS];
  const int w = blockIdx.x;
  if (w >= ws[1]) return;
  const int4 d = *(const int4*)(ws + y.work + 4 * w);
  const int r0 = d.x, r1 = d.y;
  const uint32_t nf = (uint32_t)(d.w & 0x1fffffff);
  const bool shared_slice = (d.w & 0x40000000) != 0;
  const bool hashed = (d.w & 0x20000000) != 0;
  constexpr uint32_t epb = HB_SLICE_FLOATS / C;
  for (uint32_t i = threadIdx.x; i < HB_SLICE_FLOATS; i += HB_THREADS) acc[i] = 0.f;
  __syncthreads();
  const uint32_t* rec = (const uint32_t*)((const char*)ws + y.rec_off_bytes);
  // eight records per lane in flight: the loop is a chain of (HBM load -> LDS add) otherwise
  constexpr int U = 8;
  for (int i0 = r0 + (int)threadIdx.x; i0 < r1; i0 += U * HB_THREADS) {
    uint32_t e[U];
    float v[U][C];
    // every load is issued (index clamped to the last record): a branch around a load makes the compiler wait for
    // each one in turn (cdna_hip_programming.md, "Projection GEMM" item 4c)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(i0 + u * HB_THREADS, r1 - 1);
      const uint32_t* r = rec + (size_t)i * (1 + C);
      e[u] = r[0];
#pragma unroll
      for (int ch = 0; ch < C; ++ch) v[u][ch] = __uint_as_float(r[1 + ch]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u * HB_THREADS < r1) {
        // channel planes: the 64 lanes of one ds_add_f32 spread over all 32 banks.  These LDS float atomics are what
        // bounds the kernel (0.133 of its 0.155 ms at B = 104,448; loads 0.02, flush 0.005).  ds_add_f32 costs ~170
        // cycles per wave-instruction on gfx950 against 8 for ds_add_u32 (scripts/dbg/lds_atomics.hip); a
        // compare-and-swap loop is 7x faster in that microbenchmark but slower here when used for every level (0.30 ->
        // 0.38 ms per step: the ray samples' coarse-level records repeat entries, every repeat is a retry), so only
        // the hashed levels take it -- profiles/r02_hash_scatter.md
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
          if (hashed) lds_add_cas(&acc[ch * epb + e[u]], v[u][ch]);
          else atomicAdd(&acc[ch * epb + e[u]], v[u][ch]);
        }
      }
    }
  }
  __syncthreads();
  float* table = grad_grid + (size_t)(uint32_t)d.z;
  if (!shared_slice) {
    // this workgroup owns the slice: plain read-modify-write, one entry (C floats) per lane and step.  Level offsets
    // are arbitrary entry counts (12,167 ...), so a table row is aligned to one entry, not to 16 bytes.
    typedef float vcf __attribute__((ext_vector_type(C)));
    vcf* tc = (vcf*)table;
    const uint32_t ne = nf / C;
    constexpr int UF = 8;
    for (uint32_t i0 = threadIdx.x; i0 < ne; i0 += UF * HB_THREADS) {
      vcf t[UF];
#pragma unroll
      for (int u = 0; u < UF; ++u) t[u] = tc[min(i0 + u * HB_THREADS, ne - 1)];      // all loads issued, see above
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        const uint32_t i = i0 + u * HB_THREADS;
        if (i < ne) {
          vcf a;
#pragma unroll
          for (int ch = 0; ch < C; ++ch) a[ch] = acc[ch * epb + i];
          tc[i] = t[u] + a;
        }
      }
    }
  } else {
    for (uint32_t i = threadIdx.x; i < nf; i += HB_THREADS) {
      const float v = acc[(i % C) * epb + i / C];
      if (v != 0.f) unsafeAtomicAdd(table + i, v);  // neighbouring lanes, neighbouring floats: the fast atomic shape
    }
  }
}

// ---------------------------------------------------------------------------
// Binned scatter, second form ("hb2", round 3): no counting pass, no scan kernel, no global cursors.
//
//   hb2_place_k       a workgroup takes 1,024 consecutive points of one level and owns a FIXED region of 8,192
//                     records: it counting-sorts its own corner contributions by table slice in LDS, writes them to
//                     its region in slice order and writes the (slices + 1) run offsets of the region to a run table
//                     [level][workgroup][slice].  Consecutive points that sit in the same cell (consecutive samples of
//                     a ray at the coarse levels) are summed on the way -- a segmented reduction over 16-lane rows with
//                     DPP -- so that one record leaves per run, not per point: fewer records, and the LDS adds of the
//                     second kernel rarely meet.
//   hb2_accumulate_k  one workgroup per (level, slice, group of place workgroups): reads the group's runs for its
//                     slice through the run table, adds them into an LDS accumulator with compare-and-swap float
//                     adds (both channels of an entry in one 64-bit swap; ds_add_f32 costs ~170 cycles per
//                     wave-instruction on gfx950, the swap loop ~25) and adds the slice to the table once.
// What the first form paid for exact bin sizes (hb_setup_k + hb_count_k + hb_scan_k, ~45 us of 255 at B = 104,448)
// is gone: the work list is a function of the level sizes alone.  Levels with few slices are cut by place workgroups
// instead (a coarse level of one slice becomes n_wg work items of one run each).
// ---------------------------------------------------------------------------
#define HB2_NS_MAX 8192               // slices per level the place kernel's LDS histogram is sized for
#define HB2_TILE 256                  // runs per pass of the accumulate kernel

struct Hb2Layout {
  int n_wg, ns_bound, rt_stride, work_max;
  size_t rec_off_bytes, total_bytes;
};
static Hb2Layout hb2_layout(const uint32_t B, const uint32_t C, const uint32_t L, const uint64_t n_entries) {
  Hb2Layout y;
  y.n_wg = (int)((B + HB_PTS * HB_THREADS - 1) / (HB_PTS * HB_THREADS));
  const uint64_t ns = (n_entries * C + HB_SLICE_FLOATS - 1) / HB_SLICE_FLOATS + 1;   // >= slices of any one level
  y.ns_bound = (int)(ns < (uint64_t)(1 << 30) ? ns : (uint64_t)(1 << 30));
  y.rt_stride = y.ns_bound + 1;
  // items of a level: slices x ceil(n_wg / min(slices, n_wg)) <= slices + n_wg
  y.work_max = (int)((n_entries * C + HB_SLICE_FLOATS - 1) / HB_SLICE_FLOATS + L) + (int)L * y.n_wg;
  y.rec_off_bytes = (((size_t)L * y.n_wg * y.rt_stride * 4) + 255) & ~(size_t)255;
  y.total_bytes = y.rec_off_bytes + (size_t)L * y.n_wg * (8 * HB_PTS * HB_THREADS) * (4 + 4 * C);
  return y;
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(const uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
#define DPP_ROW_SHL(n) (0x100 + (n))       // lane i reads lane i + n of its 16-lane row (0 past the row's end)
#define DPP_ROW_SHR(n) (0x110 + (n))       // lane i reads lane i - n

// v_i += v_{i+d} where lane i + d continues lane i's run (m = 1.0f) -- four steps sum a run into its first lane
template <int CN>
__device__ __forceinline__ void run_sum(float (&v)[CN], const float m1, const float m2, const float m4, const float m8) {
#pragma unroll
  for (int ch = 0; ch < CN; ++ch) {
    v[ch] = __builtin_fmaf(m1, __uint_as_float(dpp_u32<DPP_ROW_SHL(1)>(__float_as_uint(v[ch]))), v[ch]);
    v[ch] = __builtin_fmaf(m2, __uint_as_float(dpp_u32<DPP_ROW_SHL(2)>(__float_as_uint(v[ch]))), v[ch]);
    v[ch] = __builtin_fmaf(m4, __uint_as_float(dpp_u32<DPP_ROW_SHL(4)>(__float_as_uint(v[ch]))), v[ch]);
    v[ch] = __builtin_fmaf(m8, __uint_as_float(dpp_u32<DPP_ROW_SHL(8)>(__float_as_uint(v[ch]))), v[ch]);
  }
}

template <int C, int MODE>
__global__ void __launch_bounds__(HB_THREADS)
hb2_place_k(const float* __restrict__ grad, const float* __restrict__ grad2, const float* __restrict__ inputs,
            const int* __restrict__ offsets, const float* __restrict__ gg_inputs, int* __restrict__ ws,
            const Hb2Layout y, const uint32_t B, const float S, const uint32_t H, float* __restrict__ zero_grid,
            const uint32_t pitch) {       // pitch > 0: grad / grad2 are point-major [B, pitch] (level l, channel c at l C + c)
  extern __shared__ int hb2_lds[];
  int* hist = hb2_lds;                       // [ns]: counts, then run starts
  int* part = hb2_lds + y.ns_bound + 1;      // [HB_THREADS] scan scratch
  const uint32_t level = blockIdx.y;
  const HgLevel lv = hg_level(offsets, level, S, H);
  constexpr uint32_t epb = HB_SLICE_FLOATS / C;
  const int ns = (int)((lv.hsize + epb - 1) / epb);
  const int tid = threadIdx.x;
  const uint32_t lane = tid & 63;
  for (int i = tid; i < ns; i += HB_THREADS) hist[i] = 0;
  if (zero_grid != nullptr && ns < y.n_wg) {
    // "=" instead of "+=" (msdf_hash_encode_backward_fused_out): the slices of this level are shared by several
    // accumulate workgroups, which add with atomics -- the level is zeroed here, one share per place workgroup
    // (this kernel has finished before the accumulate kernel starts)
    const uint32_t nfl = lv.hsize * C, share = (nfl + y.n_wg - 1) / y.n_wg;
    float* t = zero_grid + (size_t)(uint32_t)offsets[level] * C;
    const uint32_t lo = blockIdx.x * share, hi = min(nfl, lo + share);
    for (uint32_t i = lo + tid; i < hi; i += HB_THREADS) t[i] = 0.f;
  }
  __syncthreads();

  // the gradient values of this thread's points are needed in phase 3 only, but they come from tensors the SDF kernels
  // wrote long before (cold: HBM latency): issued here, they arrive under phases 1 and 2
  float g1v[HB_PTS][C], g2v[HB_PTS][C], ggv[HB_PTS][3];
#pragma unroll
  for (int p = 0; p < HB_PTS; ++p) {
    const uint32_t b = (blockIdx.x * HB_PTS + p) * HB_THREADS + tid;
    const uint32_t bc = b < B ? b : B - 1;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      const size_t gi = pitch ? (size_t)bc * pitch + level * C + ch : ((size_t)level * B + bc) * C + ch;
      g1v[p][ch] = grad[gi];
      g2v[p][ch] = (MODE == 2) ? grad2[gi] : 0.f;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) ggv[p][d] = (MODE != 0) ? gg_inputs[(size_t)bc * 3 + d] : 0.f;
  }

  // ---- phase 1: runs of equal cells, rank of every run head's corners inside (workgroup, slice) ----
  uint32_t rank2[HB_PTS][4];                 // two 13-bit ranks per word
  uint32_t rid[HB_PTS];                      // run id inside the 16-lane row (>= 1), 255 = no contribution
  uint32_t heads = 0;
#pragma unroll
  for (int p = 0; p < HB_PTS; ++p) {
    const uint32_t b = (blockIdx.x * HB_PTS + p) * HB_THREADS + tid;
    const HgCell c = hg_locate(inputs, lv, b < B ? b : B - 1);
    const bool live = b < B && !c.oob;
    const uint32_t k1 = live ? (c.gx | (c.gy << 16)) : 0xffffffffu, k2 = live ? c.gz : 0xffffffffu;
    const uint32_t p1 = dpp_u32<DPP_ROW_SHR(1)>(k1), p2 = dpp_u32<DPP_ROW_SHR(1)>(k2);
    const bool head = live && ((lane & 15) == 0 || p1 != k1 || p2 != k2);
    const uint64_t hb = __ballot(head);
    const uint32_t row = (uint32_t)(hb >> (lane & 48)) & 0xffffu;
    rid[p] = live ? (uint32_t)__popc(row & ((2u << (lane & 15)) - 1u)) : 255u;
    if (head) heads |= 1u << p;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      uint32_t r0 = 0, r1 = 0;
      if (head) {
        const uint32_t i0 = hg_index_lv(lv, c.gx + (k & 1), c.gy + ((k >> 1) & 1), c.gz + ((k >> 2) & 1));
        const uint32_t i1 = hg_index_lv(lv, c.gx + ((k + 1) & 1), c.gy + (((k + 1) >> 1) & 1), c.gz + (((k + 1) >> 2) & 1));
        r0 = (uint32_t)atomicAdd(&hist[i0 / epb], 1);
        r1 = (uint32_t)atomicAdd(&hist[i1 / epb], 1);
      }
      rank2[p][k >> 1] = r0 | (r1 << 16);
    }
  }
  __syncthreads();

  // ---- phase 2: exclusive scan of the slice counts -> run starts; the run table row of this workgroup ----
  {
    const int per = (ns + HB_THREADS - 1) / HB_THREADS;
    const int lo = min(ns, tid * per), hi = min(ns, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += hist[i];
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < HB_THREADS; d <<= 1) {
      const int a = (tid >= d) ? part[tid - d] : 0;
      __syncthreads();
      part[tid] += a;
      __syncthreads();
    }
    int run = part[tid] - sum;
    int* rt = ws + ((size_t)level * y.n_wg + blockIdx.x) * y.rt_stride;
    for (int i = lo; i < hi; ++i) {
      const int n = hist[i];
      hist[i] = run;
      rt[i] = run;
      run += n;
    }
    if (tid == HB_THREADS - 1) rt[ns] = part[HB_THREADS - 1];
    __syncthreads();
  }

  // ---- phase 3: the records ----
  uint32_t* rec = (uint32_t*)((char*)ws + y.rec_off_bytes) +
                  ((size_t)level * y.n_wg + blockIdx.x) * (8 * HB_PTS * HB_THREADS) * (1 + C);
#pragma unroll
  for (int p = 0; p < HB_PTS; ++p) {
    const uint32_t b = (blockIdx.x * HB_PTS + p) * HB_THREADS + tid;
    const uint32_t bc = b < B ? b : B - 1;
    const HgCell c = hg_locate(inputs, lv, bc);
    const bool live = rid[p] != 255u;
    // lane i + d continues lane i's run?  (run ids of live lanes are >= 1, a read past the row's end gives 0)
    const float m1 = (live && dpp_u32<DPP_ROW_SHL(1)>(rid[p]) == rid[p]) ? 1.f : 0.f;
    const float m2 = (live && dpp_u32<DPP_ROW_SHL(2)>(rid[p]) == rid[p]) ? 1.f : 0.f;
    const float m4 = (live && dpp_u32<DPP_ROW_SHL(4)>(rid[p]) == rid[p]) ? 1.f : 0.f;
    const float m8 = (live && dpp_u32<DPP_ROW_SHL(8)>(rid[p]) == rid[p]) ? 1.f : 0.f;
    const float wx[2] = {1.f - c.sx, c.sx}, wy[2] = {1.f - c.sy, c.sy}, wz[2] = {1.f - c.sz, c.sz};
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
    if (MODE != 0) {
      q0 = ggv[p][0] * c.dx * c.scale;
      q1 = ggv[p][1] * c.dy * c.scale;
      q2 = ggv[p][2] * c.dz * c.scale;
    }
    float g1[C], g2[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      g1[ch] = live ? g1v[p][ch] : 0.f;
      g2[ch] = (MODE == 2 && live) ? g2v[p][ch] : 0.f;
    }
    const bool head = (heads >> p) & 1u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int bx = k & 1, by = (k >> 1) & 1, bz = (k >> 2) & 1;
      const float wk = wx[bx] * wy[by] * wz[bz];
      const float qk = (bx ? 1.f : -1.f) * wy[by] * wz[bz] * q0 + (by ? 1.f : -1.f) * wx[bx] * wz[bz] * q1 +
                       (bz ? 1.f : -1.f) * wx[bx] * wy[by] * q2;
      float v[C];
#pragma unroll
      for (int ch = 0; ch < C; ++ch) {
        if (MODE == 0) v[ch] = wk * g1[ch];
        else if (MODE == 1) v[ch] = qk * g1[ch];
        else v[ch] = wk * g1[ch] + qk * g2[ch];
      }
      run_sum<C>(v, m1, m2, m4, m8);
      if (head) {
        const uint32_t idx = hg_index_lv(lv, c.gx + bx, c.gy + by, c.gz + bz);
        const uint32_t pos = (uint32_t)hist[idx / epb] + ((rank2[p][k >> 1] >> ((k & 1) * 16)) & 0xffffu);
        uint32_t* r = rec + (size_t)pos * (1 + C);
        r[0] = idx % epb;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) r[1 + ch] = __float_as_uint(v[ch]);
      }
    }
  }
}

// entry e of the LDS accumulator += v[0..C): float adds as compare-and-swap loops, two channels per 64-bit swap
template <int C>
__device__ __forceinline__ void lds_add_entry(float* acc, const uint32_t e, const float (&v)[C]) {
  if (C == 1) {
    lds_add_cas(acc + e, v[0]);
  } else {
#pragma unroll
    for (int h = 0; h < C / 2; ++h) {
      unsigned long long* u = (unsigned long long*)(acc + (size_t)e * C + 2 * h);
      unsigned long long old = *u;
      while (true) {
        const float a = __uint_as_float((uint32_t)old) + v[2 * h];
        const float b = __uint_as_float((uint32_t)(old >> 32)) + v[2 * h + 1];
        const unsigned long long want = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
        const unsigned long long got = atomicCAS(u, old, want);
        if (got == old) break;
        old = got;
      }
    }
  }
}

template <int C>
__global__ void __launch_bounds__(HB_THREADS)
hb2_accumulate_k(const int* __restrict__ ws, const Hb2Layout y, const int* __restrict__ offsets, const uint32_t L,
                 float* __restrict__ grad_grid, const int overwrite) {
  __shared__ __attribute__((aligned(16))) float acc[HB_SLICE_FLOATS];      // [entry][C]
  __shared__ int pre[HB2_TILE + 1], st[HB2_TILE];
  constexpr uint32_t epb = HB_SLICE_FLOATS / C;
  // work item -> (level, slice, group of place workgroups): a function of the level sizes alone
  int w = blockIdx.x;
  uint32_t level = 0, hsize = 0;
  int ns = 0, G = 0;
  for (; level < L; ++level) {
    hsize = (uint32_t)(offsets[level + 1] - offsets[level]);
    ns = (int)((hsize + epb - 1) / epb);
    G = min(ns, y.n_wg);
    const int items = (G > 0) ? ns * ((y.n_wg + G - 1) / G) : 0;
    if (w < items) break;
    w -= items;
  }
  if (level >= L) return;
  const int slice = w % ns, wg0 = (w / ns) * G, wg1 = min(y.n_wg, wg0 + G);
  const bool shared_slice = G < y.n_wg;
  const int tid = threadIdx.x;
  for (uint32_t i = tid; i < HB_SLICE_FLOATS / 4; i += HB_THREADS) ((v4f*)acc)[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  const uint32_t* rec_all = (const uint32_t*)((const char*)ws + y.rec_off_bytes);
  constexpr size_t REGION = (size_t)(8 * HB_PTS * HB_THREADS) * (1 + C);        // dwords per place workgroup
  for (int t0 = wg0; t0 < wg1; t0 += HB2_TILE) {
    // this pass's runs: start and length per place workgroup, inclusive scan of the lengths
    const int wg = t0 + tid;
    int s0 = 0, n = 0;
    if (tid < HB2_TILE && wg < wg1) {
      const int* r = ws + ((size_t)level * y.n_wg + wg) * y.rt_stride + slice;
      s0 = r[0];
      n = r[1] - s0;
    }
    __syncthreads();                       // previous pass done with pre / st (and the zero fill, first pass)
    st[tid] = s0;
    pre[tid + 1] = n;
    if (tid == 0) pre[0] = 0;
    __syncthreads();
    for (int d = 1; d < HB2_TILE; d <<= 1) {
      const int a = (tid >= d) ? pre[tid + 1 - d] : 0;
      __syncthreads();
      pre[tid + 1] += a;
      __syncthreads();
    }
    const int total = pre[HB2_TILE];
    constexpr int U = 4;
    for (int i0 = tid; i0 < total; i0 += U * HB_THREADS) {
      uint32_t e[U];
      float v[U][C];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = min(i0 + u * HB_THREADS, total - 1);       // every load issued (index clamped), adds guarded below
        int j = 0;                                                // largest j with pre[j] <= i
#pragma unroll
        for (int step = HB2_TILE / 2; step > 0; step >>= 1)
          if (pre[j + step] <= i) j += step;
        const uint32_t* r = rec_all + ((size_t)level * y.n_wg + t0 + j) * REGION + (size_t)(st[j] + i - pre[j]) * (1 + C);
        e[u] = r[0];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) v[u][ch] = __uint_as_float(r[1 + ch]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i0 + u * HB_THREADS < total) lds_add_entry<C>(acc, e[u], v[u]);
    }
  }
  __syncthreads();
  const uint32_t e0 = (uint32_t)slice * epb;
  const uint32_t nf = min(epb, hsize - e0) * C;
  float* table = grad_grid + ((size_t)(uint32_t)offsets[level] + e0) * C;
  if (!shared_slice) {
    // this workgroup owns the slice: plain read-modify-write, one entry (C floats) per lane and step.  Level offsets
    // are arbitrary entry counts (12,167 ...), so a table row is aligned to one entry, not to 16 bytes.
    typedef float vcf __attribute__((ext_vector_type(C)));
    vcf* tc = (vcf*)table;
    const vcf* ac = (const vcf*)acc;
    const uint32_t ne = nf / C;
    if (overwrite) {                       // the table gradient is an output, not an accumulator: nothing to read
      for (uint32_t i = tid; i < ne; i += HB_THREADS) tc[i] = ac[i];
    } else {
      constexpr int UF = 8;
      for (uint32_t i0 = tid; i0 < ne; i0 += UF * HB_THREADS) {
        vcf t[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) t[u] = tc[min(i0 + u * HB_THREADS, ne - 1)];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
          const uint32_t i = i0 + u * HB_THREADS;
          if (i < ne) tc[i] = t[u] + ac[i];
        }
      }
    }
  } else {
    for (uint32_t i = tid; i < nf; i += HB_THREADS) {
      const float v = acc[i];
      if (v != 0.f) unsafeAtomicAdd(table + i, v);  // neighbouring lanes, neighbouring floats: the fast atomic shape
    }
  }
}

// MSDF_HASH_SCATTER=1 keeps the first form (count / scan / place / accumulate) for comparison runs
static bool hb_force_first_form() {
  static const int v = [] { const char* e = getenv("MSDF_HASH_SCATTER"); return (e && e[0] == '1') ? 1 : 0; }();
  return v != 0;
}

template <int C, int MODE>
static int hb_run(const float* grad, const float* grad2, const float* inputs, const int* offsets,
                  const float* gg_inputs, float* grad_grid, const uint32_t B, const uint32_t L, const float S,
                  const uint32_t H, const uint64_t n_entries, void* workspace, const size_t workspace_bytes,
                  hipStream_t st, const bool overwrite = false, const uint32_t pitch = 0) {
  if (n_entries * C >= (1ull << 31) || (uint64_t)B * L * 8 >= (1ull << 31)) return MSDF_ERR_UNSUPPORTED;
  int* ws = (int*)workspace;
  const Hb2Layout y2 = hb2_layout(B, C, L, n_entries);
  if (y2.ns_bound <= HB2_NS_MAX && !hb_force_first_form()) {
    if (workspace == nullptr || workspace_bytes < y2.total_bytes || ((uintptr_t)workspace & 15)) return MSDF_ERR_ARG;
    const size_t lds = (size_t)(y2.ns_bound + 1 + HB_THREADS) * sizeof(int);
    hb2_place_k<C, MODE><<<dim3((unsigned)y2.n_wg, L), HB_THREADS, lds, st>>>(grad, grad2, inputs, offsets, gg_inputs, ws,
                                                                             y2, B, S, H, overwrite ? grad_grid : nullptr, pitch);
    hb2_accumulate_k<C><<<(unsigned)y2.work_max, HB_THREADS, 0, st>>>(ws, y2, offsets, L, grad_grid, overwrite ? 1 : 0);
    return MSDF_OK;
  }
  if (pitch != 0) return MSDF_ERR_UNSUPPORTED;      // the first form reads level-major gradients only
  if (overwrite && hipMemsetAsync(grad_grid, 0, (size_t)n_entries * C * sizeof(float), st) != hipSuccess) return MSDF_ERR_LAUNCH;
  const HbLayout y = hb_layout(B, C, L, n_entries);
  if (workspace == nullptr || workspace_bytes < y.total_bytes || ((uintptr_t)workspace & 15)) return MSDF_ERR_ARG;
  const dim3 grid_pl((B + HB_PTS * HB_THREADS - 1) / (HB_PTS * HB_THREADS), L);
  hb_setup_k<<<1, HB_THREADS, 0, st>>>(ws, y, offsets, L, C);
  hb_count_k<C><<<grid_pl, HB_THREADS, 0, st>>>(inputs, offsets, ws, y, B, S, H);
  hb_scan_k<<<1, 1024, 0, st>>>(ws, y, offsets, L, C, S, H);
  hb_place_k<C, MODE><<<grid_pl, HB_THREADS, 0, st>>>(grad, grad2, inputs, offsets, gg_inputs, ws, y, B, S, H);
  hb_accumulate_k<C><<<(unsigned)y.work_max, HB_THREADS, 0, st>>>(ws, y, grad_grid);
  return MSDF_OK;
}

// grad_inputs[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]
template <int C>
__global__ void __launch_bounds__(HG_THREADS)
hg_backward_input_kernel(const float* __restrict__ grad, const float* __restrict__ dy_dx,
                         float* __restrict__ grad_inputs, const uint32_t B, const uint32_t L, const int level_major) {
  const uint32_t b = blockIdx.x * HG_THREADS + threadIdx.x;
  if (b >= B) return;
  float r0 = 0.f, r1 = 0.f, r2 = 0.f;
  for (uint32_t l = 0; l < L; ++l) {
    const float* dy = level_major ? dy_dx + ((size_t)l * B + b) * 3 * C : dy_dx + (size_t)b * L * 3 * C + l * 3 * C;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      const float g = grad[((size_t)l * B + b) * C + ch];
      r0 += g * dy[0 * C + ch];
      r1 += g * dy[1 * C + ch];
      r2 += g * dy[2 * C + ch];
    }
  }
  grad_inputs[(size_t)b * 3 + 0] = r0;
  grad_inputs[(size_t)b * 3 + 1] = r1;
  grad_inputs[(size_t)b * 3 + 2] = r2;
}

// grad_grad[l,b,c] = sum_d gg_inputs[b,d] * dy_dx[b,l,d,c]
template <int C>
__global__ void __launch_bounds__(HG_THREADS)
hg_second_backward_grad_kernel(const float* __restrict__ gg_inputs, const float* __restrict__ dy_dx,
                               float* __restrict__ grad_grad, const uint32_t B, const uint32_t L,
                               const int level_major) {
  const uint32_t b = blockIdx.x * HG_THREADS + threadIdx.x;
  if (b >= B) return;
  const uint32_t level = blockIdx.y;
  const float g0 = gg_inputs[(size_t)b * 3 + 0], g1 = gg_inputs[(size_t)b * 3 + 1], g2 = gg_inputs[(size_t)b * 3 + 2];
  const float* dy = level_major ? dy_dx + ((size_t)level * B + b) * 3 * C
                                : dy_dx + (size_t)b * L * 3 * C + (size_t)level * 3 * C;
#pragma unroll
  for (int ch = 0; ch < C; ++ch)
    grad_grad[((size_t)level * B + b) * C + ch] = g0 * dy[0 * C + ch] + g1 * dy[1 * C + ch] + g2 * dy[2 * C + ch];
}

// d/d embeddings of (gg_inputs . d enc/dx): +-(w * grad * gg[d] * smoothstep'(d) * scale) on the corner pairs
// (same channel-per-lane mapping as the grid backward)
// ---------------------------------------------------------------------------
// C-ABI (mirrors hash_encode_forward / _backward / _second_backward of
// code/hashencoder/src/hashencoder.h:13-15, same argument order, raw device pointers)
// ---------------------------------------------------------------------------
#define HG_DISPATCH_C(C, ...)                                    \
  switch (C) {                                                   \
    case 1: { constexpr int CC = 1; __VA_ARGS__; } break;        \
    case 2: { constexpr int CC = 2; __VA_ARGS__; } break;        \
    case 4: { constexpr int CC = 4; __VA_ARGS__; } break;        \
    case 8: { constexpr int CC = 8; __VA_ARGS__; } break;        \
    default: return MSDF_ERR_UNSUPPORTED;                        \
  }

extern "C" int msdf_hash_encode_forward(const float* inputs, const float* embeddings, const int* offsets,
                                        float* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                        uint32_t H, int calc_grad_inputs, float* dy_dx, void* stream) {
  if (D != 3) return MSDF_ERR_UNSUPPORTED;   // the reference also accepts D=2; this path only uses 3
  if (B == 0) return MSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((B + HG_THREADS - 1) / HG_THREADS, L);
  HG_DISPATCH_C(C, (hg_forward_kernel<CC><<<grid, HG_THREADS, 0, st>>>(inputs, embeddings, offsets, outputs, B, L, S,
                                                                       H, calc_grad_inputs, dy_dx)));
  return msdf_check_launch();
}

extern "C" int msdf_hash_encode_backward(const float* grad, const float* inputs, const float* embeddings,
                                         const int* offsets, float* grad_embeddings, uint32_t B, uint32_t D,
                                         uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                                         const float* dy_dx, float* grad_inputs, void* stream) {
  (void)embeddings;
  if (D != 3) return MSDF_ERR_UNSUPPORTED;
  if (B == 0) return MSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((B + HG_THREADS - 1) / HG_THREADS, L);
  HG_DISPATCH_C(C, {
    if (grad_embeddings != nullptr) {
      const int rc = hg_launch_scatter<CC, false>(grad, inputs, offsets, nullptr, grad_embeddings, B, L, S, H, st);
      if (rc != MSDF_OK) return rc;
    }
    if (calc_grad_inputs)
      hg_backward_input_kernel<CC><<<grid.x, HG_THREADS, 0, st>>>(grad, dy_dx, grad_inputs, B, L, calc_grad_inputs == 2);
  });
  return msdf_check_launch();
}

extern "C" int64_t msdf_hash_scatter_workspace_bytes(uint32_t B, uint32_t C, uint32_t L, uint64_t n_entries) {
  const size_t a = hb_layout(B, C, L, n_entries).total_bytes, b = hb2_layout(B, C, L, n_entries).total_bytes;
  return (int64_t)(a > b ? a : b);
}

extern "C" int msdf_hash_encode_backward_ws(const float* grad, const float* inputs, const float* embeddings,
                                            const int* offsets, float* grad_embeddings, uint32_t B, uint32_t D,
                                            uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                                            const float* dy_dx, float* grad_inputs, uint64_t n_entries,
                                            void* workspace, uint64_t workspace_bytes, void* stream) {
  (void)embeddings;
  if (D != 3) return MSDF_ERR_UNSUPPORTED;
  if (B == 0) return MSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  HG_DISPATCH_C(C, {
    if (grad_embeddings != nullptr) {
      const int rc = hb_run<CC, 0>(grad, nullptr, inputs, offsets, nullptr, grad_embeddings, B, L, S, H, n_entries,
                                   workspace, workspace_bytes, st);
      if (rc != MSDF_OK) return rc;
    }
    if (calc_grad_inputs)
      hg_backward_input_kernel<CC><<<(B + HG_THREADS - 1) / HG_THREADS, HG_THREADS, 0, st>>>(grad, dy_dx, grad_inputs, B, L,
                                                                                             calc_grad_inputs == 2);
  });
  return msdf_check_launch();
}

extern "C" int msdf_hash_encode_second_backward_ws(const float* grad, const float* inputs, const float* embeddings,
                                                   const int* offsets, uint32_t B, uint32_t D, uint32_t C,
                                                   uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                                                   const float* dy_dx, const float* grad_grad_inputs,
                                                   float* grad_grad, float* grad2_embeddings, uint64_t n_entries,
                                                   void* workspace, uint64_t workspace_bytes, void* stream) {
  (void)embeddings;
  if (D != 3) return MSDF_ERR_UNSUPPORTED;
  if (C == 1) return MSDF_ERR_UNSUPPORTED;
  if (B == 0) return MSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((B + HG_THREADS - 1) / HG_THREADS, L);
  HG_DISPATCH_C(C, {
    if (grad_grad != nullptr)
      hg_second_backward_grad_kernel<CC><<<grid, HG_THREADS, 0, st>>>(grad_grad_inputs, dy_dx, grad_grad, B, L,
                                                                    calc_grad_inputs == 2);
    if (grad2_embeddings != nullptr) {
      const int rc = hb_run<CC, 1>(grad, nullptr, inputs, offsets, grad_grad_inputs, grad2_embeddings, B, L, S, H,
                                   n_entries, workspace, workspace_bytes, st);
      if (rc != MSDF_OK) return rc;
    }
  });
  return msdf_check_launch();
}

// both embedding gradients of one training step in ONE scatter:
//   grad_embeddings += sum_k [ w_k * grad_first[l,b,c] + coef_k(grad_grad_inputs[b]) * grad_second[l,b,c] ]
// (the sum of what msdf_hash_encode_backward and msdf_hash_encode_second_backward add for the same points)
extern "C" int msdf_hash_encode_backward_fused(const float* grad_first, const float* grad_second, const float* inputs,
                                               const int* offsets, float* grad_embeddings, uint32_t B, uint32_t D,
                                               uint32_t C, uint32_t L, float S, uint32_t H,
                                               const float* grad_grad_inputs, uint64_t n_entries, void* workspace,
                                               uint64_t workspace_bytes, void* stream) {
  if (D != 3 || C == 1) return MSDF_ERR_UNSUPPORTED;
  if (B == 0) return MSDF_OK;
  if (grad_embeddings == nullptr || grad_first == nullptr || grad_second == nullptr || grad_grad_inputs == nullptr)
    return MSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  HG_DISPATCH_C(C, {
    const int rc = hb_run<CC, 2>(grad_first, grad_second, inputs, offsets, grad_grad_inputs, grad_embeddings, B, L, S,
                                 H, n_entries, workspace, workspace_bytes, st);
    if (rc != MSDF_OK) return rc;
  });
  return msdf_check_launch();
}

// the same with "=" instead of "+=": grad_embeddings need not be initialised (no 48.8 MB fill before the call, no
// read of the table inside it)
extern "C" int msdf_hash_encode_backward_fused_out(const float* grad_first, const float* grad_second,
                                                   const float* inputs, const int* offsets, float* grad_embeddings,
                                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                                   const float* grad_grad_inputs, uint64_t n_entries, void* workspace,
                                                   uint64_t workspace_bytes, void* stream) {
  if (D != 3 || C == 1) return MSDF_ERR_UNSUPPORTED;
  if (grad_embeddings == nullptr || grad_first == nullptr || grad_second == nullptr || grad_grad_inputs == nullptr)
    return MSDF_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (B == 0)
    return hipMemsetAsync(grad_embeddings, 0, (size_t)n_entries * C * sizeof(float), st) == hipSuccess ? MSDF_OK
                                                                                                      : MSDF_ERR_LAUNCH;
  HG_DISPATCH_C(C, {
    const int rc = hb_run<CC, 2>(grad_first, grad_second, inputs, offsets, grad_grad_inputs, grad_embeddings, B, L, S,
                                 H, n_entries, workspace, workspace_bytes, st, true);
    if (rc != MSDF_OK) return rc;
  });
  return msdf_check_launch();
}

// [L, B, C] <-> [B, pitch] (level l, channel c of a point at column l C + c; columns >= L C zero) through an LDS tile
// of 64 points: both the level-major side (64 C contiguous floats per level) and the point-major side (whole rows)
// move as contiguous runs.  Up to two tensors of the same shape per launch.
#define HT_PTS 64
template <bool TO_PM>
__global__ void __launch_bounds__(256)
hg_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ src2,
                    float* __restrict__ dst2, const uint32_t L, const uint32_t B, const uint32_t C,
                    const uint32_t pitch) {
  extern __shared__ float ht_tile[];                 // [HT_PTS][LC + 1]
  const uint32_t LC = L * C, ld = LC + 1;
  const uint32_t b0 = blockIdx.x * HT_PTS;
  const float* in = blockIdx.y ? src2 : src;
  float* out = blockIdx.y ? dst2 : dst;
  const uint32_t n_pts = min((uint32_t)HT_PTS, B - b0);
  if (TO_PM) {
    for (uint32_t i = threadIdx.x; i < L * n_pts * C; i += 256) {          // level-major reads: runs of n_pts * C floats
      const uint32_t l = i / (n_pts * C), r = i - l * (n_pts * C);
      ht_tile[(r / C) * ld + l * C + (r % C)] = in[((size_t)l * B + b0) * C + r];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_pts * pitch; i += 256) {          // rows out
      const uint32_t p = i / pitch, k = i - p * pitch;
      out[(size_t)(b0 + p) * pitch + k] = (k < LC) ? ht_tile[p * ld + k] : 0.f;
    }
  } else {
    for (uint32_t i = threadIdx.x; i < n_pts * LC; i += 256) {             // rows in
      const uint32_t p = i / LC, k = i - p * LC;
      ht_tile[p * ld + k] = in[(size_t)(b0 + p) * pitch + k];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < L * n_pts * C; i += 256) {
      const uint32_t l = i / (n_pts * C), r = i - l * (n_pts * C);
      out[((size_t)l * B + b0) * C + r] = ht_tile[(r / C) * ld + l * C + (r % C)];
    }
  }
}

extern "C" int msdf_hash_transpose(const float* src, float* dst, const float* src2, float* dst2, uint32_t L,
                                   uint32_t B, uint32_t C, uint32_t pitch, int to_point_major, void* stream) {
  if (B == 0) return MSDF_OK;          // an empty point set has NULL tensors (empty CUDA tensors have no storage)
  if (src == nullptr || dst == nullptr || (src2 == nullptr) != (dst2 == nullptr) || pitch < L * C || L * C == 0)
    return MSDF_ERR_ARG;
  const size_t lds = (size_t)HT_PTS * (L * C + 1) * sizeof(float);
  if (lds > 64 * 1024) return MSDF_ERR_UNSUPPORTED;
  const dim3 grid((B + HT_PTS - 1) / HT_PTS, src2 != nullptr ? 2 : 1);
  if (to_point_major) hg_transpose_kernel<true><<<grid, 256, lds, (hipStream_t)stream>>>(src, dst, src2, dst2, L, B, C, pitch);
  else hg_transpose_kernel<false><<<grid, 256, lds, (hipStream_t)stream>>>(src, dst, src2, dst2, L, B, C, pitch);
  return msdf_check_launch();
}

// ---- node forms (see hg_node_forward_kernel) ----
extern "C" int msdf_hash_node_forward(const float* x, double divide_factor, float* x01_out, const float* embeddings,
                                      const int* offsets, float* feat, uint32_t pitch, uint32_t B, uint32_t C,
                                      uint32_t L, float S, uint32_t H, float* dy_dx, void* stream) {
  if (B == 0) return MSDF_OK;          // as msdf_hash_encode_forward: nothing to do, NULL tensors allowed
  if ((pitch != 0 && pitch < L * C) || x == nullptr || feat == nullptr) return MSDF_ERR_ARG;
  // what the module's `x / divide_factor` multiplies by on the device: the reciprocal formed in double, rounded to
  // fp32 (scripts/dbg/x01_rounding.py: bit-identical on 196,608 values; 1.0f / 1.1f differs in 41 % of them)
  const float inv = (float)(1.0 / divide_factor);
  const dim3 grid((B + HG_THREADS - 1) / HG_THREADS, L);
  HG_DISPATCH_C(C, (hg_node_forward_kernel<CC><<<grid, HG_THREADS, 0, (hipStream_t)stream>>>(
                       x, inv, x01_out, embeddings, offsets, feat, pitch, B, L, S, H, dy_dx)));
  return msdf_check_launch();
}

extern "C" int msdf_hash_node_input_gradient(const float* g, uint32_t pitch, const float* dy_dx, uint32_t B,
                                             uint32_t C, uint32_t L, float scale, float* inout, void* stream) {
  if (B == 0) return MSDF_OK;
  if ((pitch != 0 && pitch < L * C) || g == nullptr || dy_dx == nullptr || inout == nullptr) return MSDF_ERR_ARG;
  if ((uint64_t)B * 3 >= (1ull << 32)) return MSDF_ERR_UNSUPPORTED;
  HG_DISPATCH_C(C, (hg_node_input_gradient_kernel<CC><<<(3 * B + HG_THREADS - 1) / HG_THREADS, HG_THREADS, 0,
                                                        (hipStream_t)stream>>>(g, pitch, dy_dx, B, L, scale, inout)));
  return msdf_check_launch();
}

extern "C" int msdf_hash_node_second_grad(const float* g_a, const float* g_b, uint32_t n_split, float scale,
                                          float* gg_out, const float* dy_dx, float* grad_grad, uint32_t pitch,
                                          uint32_t B, uint32_t C, uint32_t L, void* stream) {
  if (B == 0) return MSDF_OK;
  if ((pitch != 0 && pitch < L * C) || gg_out == nullptr || dy_dx == nullptr || grad_grad == nullptr || n_split > B)
    return MSDF_ERR_ARG;
  const dim3 grid((B + HG_THREADS - 1) / HG_THREADS, L);
  HG_DISPATCH_C(C, (hg_node_second_grad_kernel<CC><<<grid, HG_THREADS, 0, (hipStream_t)stream>>>(
                       g_a, g_b, n_split, scale, gg_out, dy_dx, grad_grad, pitch, B, L)));
  return msdf_check_launch();
}

// msdf_hash_encode_backward_fused_out with grad_first / grad_second as point-major rows of `pitch` floats
extern "C" int msdf_hash_node_scatter(const float* grad_first, const float* grad_second, uint32_t pitch,
                                      const float* inputs, const int* offsets, float* grad_embeddings, uint32_t B,
                                      uint32_t C, uint32_t L, float S, uint32_t H, const float* grad_grad_inputs,
                                      uint64_t n_entries, void* workspace, uint64_t workspace_bytes, void* stream) {
  if (C == 1) return MSDF_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {                        // no points: the table gradient is all zeros (the other tensors may be NULL)
    if (grad_embeddings == nullptr) return MSDF_ERR_ARG;
    return hipMemsetAsync(grad_embeddings, 0, (size_t)n_entries * C * sizeof(float), st) == hipSuccess ? MSDF_OK
                                                                                                      : MSDF_ERR_LAUNCH;
  }
  if ((pitch != 0 && pitch < L * C) || grad_embeddings == nullptr || grad_first == nullptr || grad_second == nullptr ||
      grad_grad_inputs == nullptr)
    return MSDF_ERR_ARG;
  HG_DISPATCH_C(C, {
    const int rc = hb_run<CC, 2>(grad_first, grad_second, inputs, offsets, grad_grad_inputs, grad_embeddings, B, L, S,
                                 H, n_entries, workspace, workspace_bytes, st, true, pitch);
    if (rc != MSDF_OK) return rc;
  });
  return msdf_check_launch();
}

extern "C" int msdf_hash_encode_second_backward(const float* grad, const float* inputs, const float* embeddings,
                                                const int* offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                                float S, uint32_t H, int calc_grad_inputs, const float* dy_dx,
                                                const float* grad_grad_inputs, float* grad_grad,
                                                float* grad2_embeddings, void* stream) {
  (void)embeddings;
  if (D != 3) return MSDF_ERR_UNSUPPORTED;
  if (C == 1) return MSDF_ERR_UNSUPPORTED;   // the reference has no C=1 second backward either (cu:678-684)
  if (B == 0) return MSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((B + HG_THREADS - 1) / HG_THREADS, L);
  HG_DISPATCH_C(C, {
    hg_second_backward_grad_kernel<CC><<<grid, HG_THREADS, 0, st>>>(grad_grad_inputs, dy_dx, grad_grad, B, L,
                                                                    calc_grad_inputs == 2);
    const int rc = hg_launch_scatter<CC, true>(grad, inputs, offsets, grad_grad_inputs, grad2_embeddings, B, L, S, H, st);
    if (rc != MSDF_OK) return rc;
  });
  return msdf_check_launch();
}
