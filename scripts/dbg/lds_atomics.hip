// LDS atomic throughput on gfx950: wave-instructions per cycle for ds_add_f32 / ds_add_u32 (+rtn) with random
// addresses in a 32 KB array, 256-thread workgroups, 4 per CU.   hipcc --offload-arch=gfx950 -O3 lds_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(256) k(const uint32_t* __restrict__ idx, float* out, int iters) {
  __shared__ float acc[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) acc[i] = 0.f;
  __syncthreads();
  uint32_t e[8];
  for (int u = 0; u < 8; ++u) e[u] = idx[(blockIdx.x * 8 + u) * 256 + threadIdx.x] & 8191;
  uint32_t sum = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t a = (e[u] + it * 97) & 8191;
      if (MODE == 0) atomicAdd(&acc[a], 1.0f);
      if (MODE == 1) atomicAdd((uint32_t*)&acc[a], 1u);
      if (MODE == 2) sum += atomicAdd((uint32_t*)&acc[a], 1u);
      if (MODE == 3) { float v = acc[a]; acc[a] = v + 1.0f; }        // plain (racy) read-modify-write
      if (MODE == 4) sum += __float_as_uint(atomicAdd(&acc[a], 1.0f));
      if (MODE == 5 || MODE == 6) {                                    // float add as a compare-and-swap loop
        const uint32_t aa = (MODE == 6) ? (a & ~63u) + ((threadIdx.x >> 3) & 7) : a;   // 6: 8 lanes per address
        uint32_t* p = (uint32_t*)&acc[aa];
        uint32_t old = *p;
        while (true) {
          const uint32_t want = __float_as_uint(__uint_as_float(old) + 1.0f);
          const uint32_t got = atomicCAS(p, old, want);
          if (got == old) break;
          old = got;
        }
      }
      if (MODE == 7) { const uint32_t aa = (a & ~63u) + ((threadIdx.x >> 3) & 7); atomicAdd(&acc[aa], 1.0f); }
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x] + (float)sum;
}
int main() {
  const int blocks = 1024, iters = 200;
  std::vector<uint32_t> h(blocks * 8 * 256);
  uint32_t s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 8; }
  uint32_t* d; float* o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, blocks * 256 * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const char* names[] = {"ds_add_f32", "ds_add_u32", "ds_add_rtn_u32", "plain rmw", "ds_add_rtn_f32", "cas loop",
                         "cas loop 8/addr", "ds_add_f32 8/addr"};
  for (int m = 0; m < 8; ++m) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (m == 0) k<0><<<blocks, 256>>>(d, o, iters);
      if (m == 1) k<1><<<blocks, 256>>>(d, o, iters);
      if (m == 2) k<2><<<blocks, 256>>>(d, o, iters);
      if (m == 3) k<3><<<blocks, 256>>>(d, o, iters);
      if (m == 4) k<4><<<blocks, 256>>>(d, o, iters);
      if (m == 5) k<5><<<blocks, 256>>>(d, o, iters);
      if (m == 6) k<6><<<blocks, 256>>>(d, o, iters);
      if (m == 7) k<7><<<blocks, 256>>>(d, o, iters);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double winstr = (double)blocks * 4 * iters * 8;      // wave-instructions
    printf("%-16s %.3f ms  %.1f ns per wave-instruction per CU (256 CUs) = %.0f cycles at 2.1 GHz\n", names[m], ms,
           ms * 1e6 / (winstr / 256), ms * 1e6 / (winstr / 256) * 2.1);
  }
  return 0;
}
