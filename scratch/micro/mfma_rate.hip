// Issue rate of v_mfma_f32_32x32x16_f16 / 16x16x32_f16 from ONE wave per SIMD (and two), accumulators as hipcc allocates them,
// 1 / 2 / 4 independent accumulator chains. Prints shader cycles per MFMA (s_memtime).   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int BIG>
__global__ void k(const half8* in, float* out, unsigned long long* cyc, int iters) {
  half8 a = in[threadIdx.x], b = in[threadIdx.x + 64];
  f32x16 acc[4]; f32x4 acs[4];
  for (int i = 0; i < 4; ++i) { for (int e = 0; e < 16; ++e) acc[i][e] = 0.f; for (int e = 0; e < 4; ++e) acs[i][e] = 0.f; }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        if (BIG) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
        else acs[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acs[j], 0, 0, 0);
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) { for (int e = 0; e < 16; ++e) s += acc[j][e]; for (int e = 0; e < 4; ++e) s += acs[j][e]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  half8* in; float* out; unsigned long long* cyc;
  hipMalloc(&in, 128 * 16); hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 8);
  hipMemset(in, 0x3c, 128 * 16);
  const int iters = 1000;
  auto run = [&](auto kern, const char* name, int nacc, int threads, int blocks) {
    unsigned long long h = 0;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters); hipDeviceSynchronize(); }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s threads/WG %4d WGs %4d: %.1f cycles per MFMA per wave\n", name, threads, blocks, (double)h / (iters * 8.0 * nacc));
  };
  for (int blocks : {1, 256}) {
    run(k<1, 1>, "32x32x16 1 acc", 1, 256, blocks);  run(k<2, 1>, "32x32x16 2 acc", 2, 256, blocks);  run(k<4, 1>, "32x32x16 4 acc", 4, 256, blocks);
    run(k<1, 0>, "16x16x32 1 acc", 1, 256, blocks);  run(k<2, 0>, "16x16x32 2 acc", 2, 256, blocks);  run(k<4, 0>, "16x16x32 4 acc", 4, 256, blocks);
    run(k<2, 1>, "32x32x16 2 acc, 2 waves/SIMD", 2, 512, blocks); run(k<4, 0>, "16x16x32 4 acc, 2 waves/SIMD", 4, 512, blocks);
  }
  return 0;
}
