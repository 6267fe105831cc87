// Measured ceiling for the hash-grid forward kernel's access shape on this GPU (diagnostic, not part of the library):
// per (point, level) eight 8-byte reads at pseudo-random entries of the level's 4 MB table (2^19 entries x 2 floats),
// level-major launch (grid.y = level, 256 points per workgroup) like hg_node_forward_kernel, the same coalesced
// stores (8 B output + 24 B dy_dx per (point, level)) and the 12 B point read -- and nothing else: no cell arithmetic,
// no smoothstep, one multiply-add per read so that the loads cannot be dropped.
//
//   hipcc -O3 --offload-arch=gfx950 scripts/dbg/gather_ceiling.hip -o /tmp/gather_ceiling && /tmp/gather_ceiling
//
// Modes: "random"  eight independent random entries (no reuse at all between the corners or between points);
//        "cell"    the eight corners of a random cell through the reference's xor hash (x, x+1 share ... nothing:
//                  a hashed level scatters them), i.e. the same index arithmetic as the hashed levels;
//        "ray"     cells along rays: consecutive points advance by a fraction of a cell, as the sampler's points do
//                  (neighbouring lanes often share a cell -> L1 / L2 hits the real kernel also gets).
// Prints one JSON line per mode: algorithmic bytes per launch (64 + 8 + 24 + 12 per point and level),
// microseconds per launch (HIP events, 20 launches), GB/s.
// Mode "replay" is the measured ceiling bench.py quotes (`frac_of_gather_ceiling`): the library's own level geometry,
// cell location and index function (this file includes csrc/hashgrid.hip for them) on points distributed like a
// training batch (1,024 rays x 98 samples from origins in [-0.2, 0.2]^3 + 4,096 uniform points), the real 16-level
// table (16 -> 2048, 2^19 entries per level at most), the real loads and stores -- and no interpolation arithmetic.
#include "../../monosdf_amd/csrc/hashgrid.hip"
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE, bool DY>
__global__ void __launch_bounds__(256)
gather_k(const float* __restrict__ pts, const v2f* __restrict__ tables, v2f* __restrict__ out, float* __restrict__ dy,
         const uint32_t B, const uint32_t entries_log2, const uint32_t seed) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const uint32_t level = blockIdx.y;
  const uint32_t mask = (1u << entries_log2) - 1u;
  const v2f* t = tables + ((size_t)level << entries_log2);
  const float x = pts[(size_t)b * 3], y = pts[(size_t)b * 3 + 1], z = pts[(size_t)b * 3 + 2];
  uint32_t gx, gy, gz;
  if (MODE == 2) {            // along rays of 128 samples: 1/8 of a cell per sample at this level
    const uint32_t ray = b >> 7, s = b & 127;
    gx = (mix(ray * 3 + seed) & 1023) + (s >> 3);
    gy = (mix(ray * 3 + 1 + seed) & 1023) + (s >> 4);
    gz = (mix(ray * 3 + 2 + seed) & 1023) + (s >> 5);
  } else {
    gx = mix(b * 3 + level * 77 + seed) & 2047; gy = mix(b * 3 + 1 + level * 77 + seed) & 2047;
    gz = mix(b * 3 + 2 + level * 77 + seed) & 2047;
  }
  v2f acc = {x, y};
  float a2 = z;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    uint32_t idx;
    if (MODE == 0) idx = mix(b * 8 + k + level * 0x9e3779b9u + seed) & mask;
    else idx = ((gx + (k & 1)) ^ ((gy + ((k >> 1) & 1)) * 2654435761u) ^ ((gz + ((k >> 2) & 1)) * 805459861u)) & mask;
    const v2f v = t[idx];
    acc += v * (float)(k + 1);
    a2 += v.x - v.y;
  }
  out[(size_t)level * B + b] = acc;
  if (DY) {
    float* d = dy + ((size_t)level * B + b) * 6;
    d[0] = acc.x; d[1] = acc.y; d[2] = a2; d[3] = acc.x + a2; d[4] = acc.y + a2; d[5] = a2 * 2.f;
  }
}

template <int MODE, bool DY>
static void run(const char* name, const float* pts, const v2f* tables, v2f* out, float* dy, uint32_t B, uint32_t L,
                uint32_t elog) {
  const dim3 grid((B + 255) / 256, L);
  for (int i = 0; i < 3; ++i) gather_k<MODE, DY><<<grid, 256>>>(pts, tables, out, dy, B, elog, 17u * i);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 20;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) gather_k<MODE, DY><<<grid, 256>>>(pts, tables, out, dy, B, elog, 1000u + i);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / iters;
  const double bytes = (double)B * L * (64.0 + 8.0 + (DY ? 24.0 : 0.0)) + (double)B * 12.0 * L;
  printf("{\"mode\": \"%s\", \"dy_dx\": %s, \"points\": %u, \"levels\": %u, \"table_MB_per_level\": %.1f, "
         "\"algorithmic_bytes\": %.0f, \"us_per_launch\": %.2f, \"GBps\": %.1f}\n",
         name, DY ? "true" : "false", B, L, 8.0 * (1u << elog) / 1e6, bytes, us, bytes / us / 1e3);
}

template <bool DY>
__global__ void __launch_bounds__(HG_THREADS)
replay_k(const float* __restrict__ x, const float inv_divide, const v2f* __restrict__ grid, const int* __restrict__ offsets,
         v2f* __restrict__ out, float* __restrict__ dy, const uint32_t B, const float S, const uint32_t H) {
  const uint32_t b = blockIdx.x * HG_THREADS + threadIdx.x;
  if (b >= B) return;
  const uint32_t level = blockIdx.y;
  const HgLevel lv = hg_level(offsets, level, S, H);
  const float u0 = (x[(size_t)b * 3 + 0] * inv_divide + 1.0f) * 0.5f;
  const float u1 = (x[(size_t)b * 3 + 1] * inv_divide + 1.0f) * 0.5f;
  const float u2 = (x[(size_t)b * 3 + 2] * inv_divide + 1.0f) * 0.5f;
  const HgCell c = hg_locate_xyz(u0, u1, u2, lv);
  v2f acc = {0.f, 0.f};
  float a2 = 0.f;
  if (!c.oob) {
    const v2f* t = grid + (size_t)(uint32_t)offsets[level];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const v2f v = t[hg_index_lv(lv, c.gx + (k & 1), c.gy + ((k >> 1) & 1), c.gz + ((k >> 2) & 1))];
      acc += v;
      a2 += v.x - v.y;
    }
  }
  out[(size_t)level * B + b] = acc;
  if (DY) {
    float* d = dy + ((size_t)level * B + b) * 6;
    d[0] = acc.x; d[1] = acc.y; d[2] = a2; d[3] = acc.x; d[4] = acc.y; d[5] = a2;
  }
}

static double lcg(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; }

static void replay(const bool with_dy) {
  const uint32_t L = 16, H = 16, n_rays = 1024, spr = 98, n_eik = 4096, B = n_rays * spr + n_eik;
  const float S = (float)(std::log2(2048.0 / 16.0) / 15.0);
  std::vector<int> off(L + 1);
  size_t total = 0;
  for (uint32_t l = 0; l < L; ++l) {
    const double res = std::ceil(16.0 * std::pow(std::exp2(std::log2(2048.0 / 16.0) / 15.0), (double)l));
    off[l] = (int)total;
    const double n = res * res * res;
    total += (size_t)(n < 524288.0 ? n : 524288.0);
  }
  off[L] = (int)total;
  std::vector<float> pts((size_t)B * 3);
  uint64_t s = 12345;
  for (uint32_t r = 0; r < n_rays; ++r) {
    double o[3], d[3], nn = 0;
    for (int k = 0; k < 3; ++k) { o[k] = -0.2 + 0.4 * lcg(s); d[k] = 2 * lcg(s) - 1; nn += d[k] * d[k]; }
    nn = std::sqrt(nn) + 1e-9;
    for (uint32_t i = 0; i < spr; ++i) {
      const double z = 1.3 * (i + lcg(s)) / spr;         // to the bounding sphere (radius 1.1) and a little beyond
      for (int k = 0; k < 3; ++k) pts[((size_t)r * spr + i) * 3 + k] = (float)(o[k] + z * d[k] / nn);
    }
  }
  for (uint32_t i = 0; i < n_eik; ++i)
    for (int k = 0; k < 3; ++k) pts[((size_t)n_rays * spr + i) * 3 + k] = (float)(-1.1 + 2.2 * lcg(s));
  float* x; v2f* grid; int* offs; v2f* out; float* dy;
  CHECK(hipMalloc(&x, pts.size() * 4)); CHECK(hipMalloc(&grid, total * 8)); CHECK(hipMalloc(&offs, (L + 1) * 4));
  CHECK(hipMalloc(&out, (size_t)L * B * 8)); CHECK(hipMalloc(&dy, (size_t)L * B * 24));
  CHECK(hipMemcpy(x, pts.data(), pts.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(offs, off.data(), (L + 1) * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(grid, 0, total * 8));
  const dim3 g((B + HG_THREADS - 1) / HG_THREADS, L);
  const float inv = (float)(1.0 / 1.1);
  auto launch = [&]() {
    if (with_dy) replay_k<true><<<g, HG_THREADS>>>(x, inv, grid, offs, out, dy, B, S, H);
    else replay_k<false><<<g, HG_THREADS>>>(x, inv, grid, offs, out, dy, B, S, H);
  };
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 20;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / iters;
  // SURVEY 8(d) bytes per point: 1,548 with dy_dx (16 levels x (64 + 8 + 24) + 12), 1,164 without
  const double bytes = (double)B * (with_dy ? 1548.0 : 1164.0);
  printf("{\"mode\": \"replay\", \"dy_dx\": %s, \"points\": %u, \"levels\": %u, \"table_MB\": %.1f, "
         "\"algorithmic_bytes\": %.0f, \"us_per_launch\": %.2f, \"GBps\": %.1f}\n",
         with_dy ? "true" : "false", B, L, 8.0 * total / 1e6, bytes, us, bytes / us / 1e3);
  hipFree(x); hipFree(grid); hipFree(offs); hipFree(out); hipFree(dy);
}

int main(int argc, char** argv) {
  replay(true);
  replay(false);
  const uint32_t B = argc > 1 ? (uint32_t)atoi(argv[1]) : 104448u, L = 16, elog = 19;
  float* pts; v2f* tables; v2f* out; float* dy;
  CHECK(hipMalloc(&pts, (size_t)B * 12));
  CHECK(hipMalloc(&tables, ((size_t)L << elog) * 8));
  CHECK(hipMalloc(&out, (size_t)L * B * 8));
  CHECK(hipMalloc(&dy, (size_t)L * B * 24));
  CHECK(hipMemset(pts, 0, (size_t)B * 12));
  CHECK(hipMemset(tables, 0, ((size_t)L << elog) * 8));
  run<0, true>("random", pts, tables, out, dy, B, L, elog);
  run<1, true>("cell", pts, tables, out, dy, B, L, elog);
  run<2, true>("ray", pts, tables, out, dy, B, L, elog);
  run<0, false>("random", pts, tables, out, dy, B, L, elog);
  run<2, false>("ray", pts, tables, out, dy, B, L, elog);
  return 0;
}
