// Does memory traffic of the partner wave slow a wave's MFMA stream? Waves 0-3 issue v_mfma_f32_32x32x16_f16 (optionally fetching their A
// operand from LDS with ds_read_b128 every second MFMA, as the attention kernel does); waves 4-7 (SIMD partners) run: nothing, an LDS-DMA
// stream (global_load_lds_dwordx4, 16 KiB per round, drained with vmcnt(0)), a ds_read_b128 stream, or plain global loads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { P_NONE, P_DMA, P_DSREAD, P_GLOAD };
template <int P, bool FRAG>
__global__ __launch_bounds__(512, 1) void k(const half8* in, const char* big, float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  half8 a = in[lane], b = in[lane + 64];
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int i = threadIdx.x; i < 98304 / 16; i += 512) ((f32x4*)smem)[i] = (f32x4){1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  f32x4 sink = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (FRAG) a = *(const half8*)(smem + ((it * 4 + r) & 63) * 1024 + lane * 16);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0);
      }
    }
  } else {
    const char* src = big + (size_t)blockIdx.x * (1 << 20) + (size_t)(wave - 4) * (256 << 10);
    for (int it = 0; it < iters; ++it) {
      if (P == P_DMA) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((it * 4 + j) & 255) * 1024 + lane * 16),
                                           (__attribute__((address_space(3))) void*)(smem + 65536 + (wave - 4) * 4096 + j * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (P == P_DSREAD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const f32x4 x = *(const f32x4*)(smem + ((it * 8 + j) & 63) * 1024 + lane * 16); sink += x; }
      } else if (P == P_GLOAD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const f32x4 x = *(const f32x4*)(src + ((it * 4 + j) & 255) * 1024 + lane * 16); sink += x; }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = sink[0] + sink[1] + sink[2] + sink[3];
  for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)a[0];
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
int main() {
  half8* in; char* big; float* out; unsigned long long* cyc;
  hipMalloc(&in, 64 * 2 * 16); hipMalloc(&big, (size_t)256 << 20); hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 64);
  hipMemset(big, 0, (size_t)256 << 20);
  _Float16 h[64 * 2 * 8]; for (int i = 0; i < 64 * 2 * 8; ++i) h[i] = (_Float16)(((i * 2654435761u) >> 20) % 2001 / 1000.0f - 1.0f);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 400;
  auto run = [&](auto kern, const char* name) {
    unsigned long long c[8];
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(512), 98304, 0, in, big, out, cyc, iters); hipDeviceSynchronize(); }
    hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-64s MFMA wave %.1f cycles per MFMA; partner %.0f cycles per round\n", name, (double)c[0] / (iters * 8.0), (double)c[4] / iters);
  };
  run(k<P_NONE, false>, "operands in registers, partner idle");
  run(k<P_NONE, true>, "A fragment from LDS every 2nd MFMA, partner idle");
  run(k<P_DMA, false>, "operands in registers, partner LDS-DMA (4 KiB per round)");
  run(k<P_DMA, true>, "A fragment from LDS, partner LDS-DMA");
  run(k<P_DSREAD, false>, "operands in registers, partner ds_read_b128 x 8 per round");
  run(k<P_DSREAD, true>, "A fragment from LDS, partner ds_read_b128 x 8 per round");
  run(k<P_GLOAD, false>, "operands in registers, partner global_load_dwordx4 x 4");
  return 0;
}
