// LDS-DMA (global_load_lds_dwordx4) throughput per CU vs the row granularity of one wave-instruction:
//   64-B pieces  (16 rows x 64 B,  the x3 BK=32 operand layout: hi and lo in separate arrays)
//   128-B pieces ( 8 rows x 128 B, what an interleaved [hi 32 | lo 32] layout would give)
//   256-B pieces ( 4 rows x 256 B)
// 4 loader waves per workgroup, one workgroup per CU, each wave streams through a 128-row x K panel that stays L2 resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int ROWB>   // bytes per row per piece
__global__ __launch_bounds__(256) void k(const char* base, long row_stride, int ksteps, int reps, float* sink) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = ROWB / 16;                 // lanes per row
  constexpr int RPI = 64 / LPR;                  // rows per instruction
  const char* panel = base + (size_t)(blockIdx.x % 64) * 128 * row_stride;      // 64 distinct panels, shared by 4 CUs each
  const long lane_off = (long)(lane / LPR) * row_stride + (lane % LPR) * 16;
  for (int r = 0; r < reps; ++r)
    for (int ks = 0; ks < ksteps; ++ks) {
      // per k-step each wave moves 8 KiB (8 instructions), the workgroup 32 KiB, like the x3 GEMM
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int piece = wave * 8 + t;                        // 32 pieces of 1 KiB
        const int row0 = (piece * RPI) % 128;
        const int col = ((piece * RPI) / 128) * ROWB;
        glds16(panel + (long)row0 * row_stride + lane_off + (long)ks * (32768 / 128) + col, smem + (ks & 3) * 32768 + piece * 1024);
      }
      if ((ks & 1) == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = ((float*)smem)[lane];
}
template <int ROWB> void run(const char* name) {
  const int K = 4096; const long row_stride = (long)K * 4;      // bytes per row (hi+lo of 4096 halves)
  char* buf; hipMalloc(&buf, (size_t)64 * 128 * row_stride);
  hipMemset(buf, 1, (size_t)64 * 128 * row_stride);
  float* sink; hipMalloc(&sink, 1024);
  const int ksteps = 64, reps = 8;
  hipFuncSetAttribute((const void*)k<ROWB>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipLaunchKernelGGL(k<ROWB>, dim3(256), dim3(256), 131072, 0, buf, row_stride, ksteps, 1, sink);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<ROWB>, dim3(256), dim3(256), 131072, 0, buf, row_stride, ksteps, reps, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * ksteps * reps * 32768;
  printf("%s: %.3f ms, %.1f GB/s per CU, %.2f TB/s chip, %.2f us per 32 KiB k-step\n", name, ms, bytes / 256 / ms / 1e6, bytes / ms / 1e9,
         ms * 1e3 / (ksteps * reps));
  hipFree(buf); hipFree(sink);
}
int main() {
  run<64>("64-B pieces (16 rows x 64 B)");
  run<128>("128-B pieces (8 rows x 128 B)");
  run<256>("256-B pieces (4 rows x 256 B)");
  return 0;
}
