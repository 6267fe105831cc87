// Which fp32 MFMA shape does the chip run fastest under its power limit?  Bare loops of every fp32 MFMA shape on
// random operands (registers only), all CUs, 2 waves per SIMD.  Prints TFLOP/s, cycles per MFMA and the shader clock.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_shapes scripts/dbg/mfma_shapes.hip && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v32f __attribute__((ext_vector_type(32)));

// SHAPE: 0 = 16x16x4, 1 = 32x32x2, 2 = 16x16x1 (4 blocks), 3 = 32x32x1 (2 blocks), 4 = 4x4x1 (16 blocks)
template <int SHAPE>
__global__ void __launch_bounds__(256) loop_k(const float* __restrict__ in, float* __restrict__ out, int iters,
                                              unsigned long long* cyc) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = in[(t * 16 + i) & 65535]; b[i] = in[(t * 16 + 8 + i) & 65535]; }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  float res = 0.f;
  if (SHAPE == 0 || SHAPE == 4) {
    v4f c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        c[i] = (SHAPE == 0) ? __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], c[i], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_4x4x1f32(a[i], b[i], c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) res += c[i].x + c[i].y + c[i].z + c[i].w;
  } else if (SHAPE == 1 || SHAPE == 2) {
    v16f c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        c[i] = (SHAPE == 1) ? __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], c[i], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_16x16x1f32(a[i], b[i], c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) res += c[i][j];
  } else if (SHAPE == 5 || SHAPE == 6) {
    // one (5) or two (6) dependent chains of 16x16x1 4B per wave
    v16f c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int w = (SHAPE == 5) ? 0 : (i & 1);
        c[w] = __builtin_amdgcn_mfma_f32_16x16x1f32(a[i], b[i], c[w], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) res += c[i][j];
  } else if (SHAPE == 7) {
    // 16x16x4 with the A operand re-read from LDS: one ds_read_b128 per 4 MFMAs, two accumulators (the MLP kernels' mix)
    __shared__ v4f wbuf[64 * 33];
    for (int i = threadIdx.x; i < 64 * 33; i += 256) wbuf[i] = (v4f){a[0], a[1], a[2], a[3]} * (float)(i & 7);
    __syncthreads();
    v4f c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
    const v4f* w = wbuf + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const v4f f0 = w[(2 * k) * 64 + (it & 15) * 64], f1 = w[(2 * k + 1) * 64 + (it & 15) * 64];
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f0.x, b[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.x, b[0], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f0.y, b[1], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.y, b[1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f0.z, b[2], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.z, b[2], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f0.w, b[3], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.w, b[3], c1, 0, 0, 0);
      }
    }
    res = c0.x + c0.y + c0.z + c0.w + c1.x + c1.y + c1.z + c1.w;
  } else if (SHAPE == 8) {
    // 32x32x2 with the A operand re-read from LDS: one ds_read_b128 per 4 MFMAs (half the LDS bytes per product)
    __shared__ v4f wbuf2[64 * 33];
    for (int i = threadIdx.x; i < 64 * 33; i += 256) wbuf2[i] = (v4f){a[0], a[1], a[2], a[3]} * (float)(i & 7);
    __syncthreads();
    v16f c0, c1;
#pragma unroll
    for (int j = 0; j < 16; ++j) { c0[j] = 0.f; c1[j] = 0.f; }
    const v4f* w = wbuf2 + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
      const v4f f0 = w[(it & 15) * 64], f1 = w[64 + (it & 15) * 64];
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.x, b[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f1.x, b[0], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.y, b[1], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f1.y, b[1], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.z, b[2], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f1.z, b[2], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.w, b[3], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f1.w, b[3], c1, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) res += c0[j] + c1[j];
  } else {
    v32f c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 32; ++j) c[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 2; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x1f32(a[i], b[i], c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 32; ++j) res += c[i][j];
  }
  out[t] = res;
  if (t == 0) *cyc = __builtin_amdgcn_s_memtime() - c0;
}

template <int SHAPE>
void run(const char* name, const float* in, float* out, int blocks, int iters, unsigned long long* cyc,
         double flop_per_mfma, int mfma_per_iter) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  loop_k<SHAPE><<<blocks, 256>>>(in, out, iters / 10, cyc);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  loop_k<SHAPE><<<blocks, 256>>>(in, out, iters, cyc);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long hc = 0;
  (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  const double n_mfma = (double)mfma_per_iter * iters;
  printf("%-24s %7.1f TFLOP/s  %6.1f cycles per MFMA of a wave (2 waves per SIMD)  shader clock %.2f GHz\n", name,
         flop_per_mfma * n_mfma * blocks * 4 / (ms * 1e-3) / 1e12, hc / n_mfma, hc / (ms * 1e6));
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 2;       // 2 waves per SIMD
  std::vector<float> h(65536);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  float *in, *out;
  unsigned long long* cyc;
  (void)hipMalloc(&cyc, 8);
  (void)hipMalloc(&in, 65536 * 4); (void)hipMalloc(&out, blocks * 256 * 4);
  (void)hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  const int iters = 300000;     // >= 0.1 s per launch: long enough for the power limit to act
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("16x16x4", in, out, blocks, iters, cyc, 2.0 * 16 * 16 * 4, 8);
    run<1>("32x32x2", in, out, blocks, iters, cyc, 2.0 * 32 * 32 * 2, 4);
    run<2>("16x16x1 4B", in, out, blocks, iters, cyc, 2.0 * 16 * 16 * 1 * 4, 4);
    run<3>("32x32x1 2B", in, out, blocks, iters, cyc, 2.0 * 32 * 32 * 1 * 2, 2);
    run<4>("4x4x1 16B", in, out, blocks, iters * 2, cyc, 2.0 * 4 * 4 * 1 * 16, 8);
    run<5>("16x16x1 4B, 1 chain", in, out, blocks, iters, cyc, 2.0 * 16 * 16 * 1 * 4, 4);
    run<6>("16x16x1 4B, 2 chains", in, out, blocks, iters, cyc, 2.0 * 16 * 16 * 1 * 4, 4);
    run<5>("1 chain, 1 wave/SIMD", in, out, blocks / 2, iters, cyc, 2.0 * 16 * 16 * 1 * 4, 4);
    run<6>("2 chains, 1 wave/SIMD", in, out, blocks / 2, iters, cyc, 2.0 * 16 * 16 * 1 * 4, 4);
    run<0>("16x16x4, 1 wave/SIMD", in, out, blocks / 2, iters, cyc, 2.0 * 16 * 16 * 4, 8);
    run<7>("16x16x4 + LDS A reads", in, out, blocks, iters / 2, cyc, 2.0 * 16 * 16 * 4, 16);
    run<8>("32x32x2 + LDS A reads", in, out, blocks, iters / 2, cyc, 2.0 * 32 * 32 * 2, 8);
  }
  return 0;
}
