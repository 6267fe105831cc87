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

// per-level constants (scale, resolution, table size): uniform over a workgroup of the level-major launches
struct HgLevel {
  float scale;
  uint32_t res, hsize;
};
__device__ __forceinline__ HgLevel hg_level(const int* __restrict__ offsets, const uint32_t level, const float S,
                                            const uint32_t H) {
  HgLevel v;
  v.hsize = (uint32_t)(offsets[level + 1] - offsets[level]);
  // exp2f(level * S) as the correctly rounded float (device exp2f is only ~1 ulp; one ulp of scale moves a
  // fine-level sample by 1e-4 of a cell)
  v.scale = (float)exp2((double)((float)level * S)) * (float)H - 1.0f;
  v.res = (uint32_t)ceilf(v.scale) + 1u;
  return v;
}

__device__ __forceinline__ HgCell hg_locate(const float* __restrict__ inputs, const HgLevel& lv, const uint32_t b) {
  HgCell c;
  const float x = inputs[(size_t)b * 3 + 0], y = inputs[(size_t)b * 3 + 1], z = inputs[(size_t)b * 3 + 2];
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

__device__ __forceinline__ HgCell hg_locate(const float* __restrict__ inputs, const int* __restrict__ offsets,
                                            const uint32_t b, const uint32_t level, const float S,
                                            const uint32_t H) {
  return hg_locate(inputs, hg_level(offsets, level, S, H), b);
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(HG_THREADS)
hg_forward_kernel(const float* __restrict__ inputs, const float* __restrict__ grid, const int* __restrict__ offsets,
                  float* __restrict__ outputs, const uint32_t B, const uint32_t L, const float S, const uint32_t H,
                  const int calc_grad_inputs, float* __restrict__ dy_dx) {
  const uint32_t b = blockIdx.x * HG_THREADS + threadIdx.x;
  if (b >= B) return;
  const uint32_t level = blockIdx.y;
  float* out = outputs + ((size_t)level * B + b) * C;
  float* dy = dy_dx + (size_t)b * 3 * L * C + (size_t)level * 3 * C;
  const HgCell c = hg_locate(inputs, offsets, b, level, S, H);
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
    const uint32_t idx = hg_index(c.gx + (k & 1), c.gy + ((k >> 1) & 1), c.gz + ((k >> 2) & 1), c.hsize, c.res);
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

// grad_inputs[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]
template <int C>
__global__ void __launch_bounds__(HG_THREADS)
hg_backward_input_kernel(const float* __restrict__ grad, const float* __restrict__ dy_dx,
                         float* __restrict__ grad_inputs, const uint32_t B, const uint32_t L) {
  const uint32_t b = blockIdx.x * HG_THREADS + threadIdx.x;
  if (b >= B) return;
  float r0 = 0.f, r1 = 0.f, r2 = 0.f;
  const float* dy = dy_dx + (size_t)b * L * 3 * C;
  for (uint32_t l = 0; l < L; ++l) {
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      const float g = grad[((size_t)l * B + b) * C + ch];
      r0 += g * dy[l * 3 * C + 0 * C + ch];
      r1 += g * dy[l * 3 * C + 1 * C + ch];
      r2 += g * dy[l * 3 * C + 2 * C + ch];
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
                               float* __restrict__ grad_grad, const uint32_t B, const uint32_t L) {
  const uint32_t b = blockIdx.x * HG_THREADS + threadIdx.x;
  if (b >= B) return;
  const uint32_t level = blockIdx.y;
  const float g0 = gg_inputs[(size_t)b * 3 + 0], g1 = gg_inputs[(size_t)b * 3 + 1], g2 = gg_inputs[(size_t)b * 3 + 2];
  const float* dy = dy_dx + (size_t)b * L * 3 * C + (size_t)level * 3 * C;
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
      hg_backward_input_kernel<CC><<<grid.x, HG_THREADS, 0, st>>>(grad, dy_dx, grad_inputs, B, L);
  });
  return msdf_check_launch();
}

extern "C" int msdf_hash_encode_second_backward(const float* grad, const float* inputs, const float* embeddings,
                                                const int* offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                                float S, uint32_t H, int calc_grad_inputs, const float* dy_dx,
                                                const float* grad_grad_inputs, float* grad_grad,
                                                float* grad2_embeddings, void* stream) {
  (void)embeddings; (void)calc_grad_inputs;
  if (D != 3) return MSDF_ERR_UNSUPPORTED;
  if (C == 1) return MSDF_ERR_UNSUPPORTED;   // the reference has no C=1 second backward either (cu:678-684)
  if (B == 0) return MSDF_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((B + HG_THREADS - 1) / HG_THREADS, L);
  HG_DISPATCH_C(C, {
    hg_second_backward_grad_kernel<CC><<<grid, HG_THREADS, 0, st>>>(grad_grad_inputs, dy_dx, grad_grad, B, L);
    const int rc = hg_launch_scatter<CC, true>(grad, inputs, offsets, grad_grad_inputs, grad2_embeddings, B, L, S, H, st);
    if (rc != MSDF_OK) return rc;
  });
  return msdf_check_launch();
}
