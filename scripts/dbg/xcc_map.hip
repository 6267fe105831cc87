// Which XCD does workgroup i of a 1-D launch run on?  (gfx950: HW_REG_XCC_ID, bits 3:0 = XCC id)
//   hipcc --offload-arch=gfx950 -O2 xcc_map.hip -o xcc_map && ./xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
  const int n = 4096;
  unsigned* d;
  hipMalloc(&d, n * 4);
  for (int threads : {64, 256, 1024}) {
    k<<<n, threads>>>(d);
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    int match = 0, hist[16] = {0};
    for (int i = 0; i < n; ++i) { match += ((h[i] & 15) == (unsigned)(i & 7)); hist[h[i] & 15]++; }
    printf("threads %4d: %d of %d workgroups on XCD (i mod 8); first 24:", threads, match, n);
    for (int i = 0; i < 24; ++i) printf(" %u", h[i] & 15);
    printf("  per-XCD counts:");
    for (int x = 0; x < 8; ++x) printf(" %d", hist[x]);
    printf("\n");
  }
  return 0;
}
