// Cost of s_barrier per iteration for 4- and 8-wave workgroups, alone and with MFMA work between barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MF>
__global__ void k(int iters, float* out) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
  const bool consumer = (threadIdx.x >> 6) < 4;
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_s_barrier();
    if (MF > 0 && consumer) {
#pragma unroll
      for (int j = 0; j < MF; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j & 3], 0, 0, 0);
    }
  }
  if (out) out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
template <int MF> void run(int threads, const char* name) {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(k<MF>, dim3(256), dim3(threads), 0, 0, 100, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MF>, dim3(256), dim3(threads), 0, 0, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s threads=%d mfma/iter=%d: %.1f ns per iteration (%.0f cycles @2.4GHz); mfma-only bound %.0f cycles\n", name, threads, MF,
         ms * 1e6 / iters, ms * 1e6 / iters * 2.4, MF * 16.0);
  hipFree(out);
}
int main() {
  run<0>(256, "barrier only"); run<0>(512, "barrier only");
  run<48>(256, "barrier+48 mfma"); run<48>(512, "barrier+48 mfma (4 of 8 waves)");
  run<96>(512, "barrier+96 mfma (4 of 8 waves)");
  return 0;
}
