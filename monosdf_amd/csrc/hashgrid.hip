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
  // calc_grad_inputs == 2: dy_dx level-major [L, B, 3 C] (a wave writes 64 x 3 C contiguous floats) instead of the
  // reference's [B, L, 3 C] (3 C floats every L 3 C: partial lines written by 16 different launches' blocks)
  float* dy = (calc_grad_inputs == 2) ? dy_dx + ((size_t)level * B + b) * 3 * C
                                      : dy_dx + (size_t)b * 3 * L * C + (size_t)level * 3 * C;
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
#define HB_SLICE_FLOATS 8192
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

template <int C, int MODE>
static int hb_run(const float* grad, const float* grad2, const float* inputs, const int* offsets,
                  const float* gg_inputs, float* grad_grid, const uint32_t B, const uint32_t L, const float S,
                  const uint32_t H, const uint64_t n_entries, void* workspace, const size_t workspace_bytes,
                  hipStream_t st) {
  const HbLayout y = hb_layout(B, C, L, n_entries);
  if (workspace == nullptr || workspace_bytes < y.total_bytes || ((uintptr_t)workspace & 15)) return MSDF_ERR_ARG;
  if (n_entries * C >= (1ull << 31)) return MSDF_ERR_UNSUPPORTED;
  int* ws = (int*)workspace;
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
  return (int64_t)hb_layout(B, C, L, n_entries).total_bytes;
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
