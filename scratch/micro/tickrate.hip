// s_memtime tick rate: one wave spins for N ticks, wall time by HIP events (light load: what the counter's clock is when the chip is not
// power-limited), then the same with 1024 workgroups hammering MFMAs beside it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void spin(unsigned long long n, unsigned long long* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long t = t0;
  while (t - t0 < n) t = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t - t0;
}
__global__ __launch_bounds__(512, 1) void burn(const half8* in, float* out, int iters, unsigned long long* cyc) {
  half8 a = in[threadIdx.x & 63], b = in[(threadIdx.x & 63) + 64];
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[1] = t1 - t0;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64); half8* in; float* out; hipMalloc(&in, 4096); hipMalloc(&out, 8 << 20);
  _Float16 h[2048]; for (int i = 0; i < 2048; ++i) h[i] = (_Float16)(((i * 2654435761u) >> 20) % 2001 / 1000.0f - 1.0f);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 200000000ull, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); unsigned long long t; hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
    printf("idle chip: %llu ticks in %.3f ms -> %.1f MHz\n", t, ms, t / ms / 1e3);
  }
  for (int rep = 0; rep < 3; ++rep) {
    const int iters = 20000;
    hipEventRecord(e0); hipLaunchKernelGGL(burn, dim3(256), dim3(512), 0, 0, in, out, iters, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); unsigned long long t; hipMemcpy(&t, d + 1, 8, hipMemcpyDeviceToHost);
    printf("all CUs MFMA 32x32x16 f16 (8 waves per CU): %llu ticks in %.3f ms -> %.1f MHz; %.1f ticks per MFMA per wave; %.0f TFLOP/s\n", t, ms, t / ms / 1e3,
           (double)t / (iters * 8.0), 256.0 * 8 * iters * 8.0 * 32768.0 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
