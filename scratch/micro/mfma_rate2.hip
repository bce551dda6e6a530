// What slows a wave's v_mfma_f32_32x32x16_f16 stream below one per 32 cycles? Variants: operand registers rotating (as when fragments
// come from LDS), two interleaved accumulator chains, a partner wave on the same SIMD running a VALU loop (exp / cvt) or idling.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// MODE bit0: rotate A operands over 4 register sets; bit1: waves 4-7 run a VALU loop instead of MFMAs; bit2: waves 4-7 idle at once
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(const half8* in, float* out, unsigned long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  half8 a[4], b = in[threadIdx.x & 63];
  for (int i = 0; i < 4; ++i) a[i] = in[(threadIdx.x & 63) + 64 * (i + 1)];
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float v[8]; for (int i = 0; i < 8; ++i) v[i] = (float)threadIdx.x * 0.001f + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4 || !(MODE & 6)) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const half8 x = (MODE & 1) ? a[r] : a[0];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, b, acc[1], 0, 0, 0);
      }
    }
  } else if (MODE & 2) {
    for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.5f + (float)(_Float16)v[i]; }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
int main() {
  half8* in; float* out; unsigned long long* cyc;
  hipMalloc(&in, 64 * 5 * 16); hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 64);
  _Float16 h[64 * 5 * 8]; for (int i = 0; i < 64 * 5 * 8; ++i) h[i] = (_Float16)(((i * 2654435761u) >> 20) % 2001 / 1000.0f - 1.0f);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 500;
  auto run = [&](auto kern, const char* name) {
    unsigned long long c[8];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, in, out, cyc, iters); hipDeviceSynchronize(); }
    hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-58s wave0 %.1f  wave4 %.1f cycles per MFMA (of wave 0's count)\n", name, (double)c[0] / (iters * 8.0), (double)c[4] / (iters * 8.0));
  };
  run(k<0>, "8 waves all MFMA, constant operands");
  run(k<1>, "8 waves all MFMA, rotating A operands");
  run(k<2>, "waves 0-3 MFMA, waves 4-7 VALU loop (exp, cvt)");
  run(k<3>, "waves 0-3 MFMA rotating A, waves 4-7 VALU loop");
  run(k<4>, "waves 0-3 MFMA, waves 4-7 exit");
  run(k<5>, "waves 0-3 MFMA rotating A, waves 4-7 exit");
  return 0;
}
