// Which VALU instruction class of a partner wave on the same SIMD slows a wave's v_mfma_f32_32x32x16_f16 stream (32 cycles each alone)?
// Waves 0-3 (one per SIMD) issue MFMAs; waves 4-7 (their SIMD partners) run ONE class of instruction in a loop. Printed: cycles per MFMA of
// wave 0 and cycles per partner instruction of wave 4 (against the partner loop alone).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
enum { P_NONE, P_FMA, P_PKADD, P_PKMUL, P_MAX, P_EXP, P_CVTPK, P_CVT16, P_DPP, P_BPERM, P_MUL, P_LDSREAD };
template <int P, bool MF>
__global__ __launch_bounds__(512, 1) void k(const half8* in, float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  half8 a = in[lane], b = in[lane + 64];
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float v[8]; for (int i = 0; i < 8; ++i) v[i] = (float)threadIdx.x * 0.001f + i;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = (float)i;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (MF) for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0);
      }
    }
  } else if (P != P_NONE) {
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        if (P == P_FMA) { v[i] = v[i] * 1.0001f + 0.5f; v[i + 1] = v[i + 1] * 0.9999f + 0.25f; }
        if (P == P_MUL) { v[i] *= 1.0001f; v[i + 1] *= 0.9999f; }
        if (P == P_PKADD) { f32x2 d = {v[i], v[i + 1]}; const f32x2 c = {0.5f, 0.25f}; d += c; v[i] = d[0]; v[i + 1] = d[1]; }
        if (P == P_PKMUL) { f32x2 d = {v[i], v[i + 1]}; const f32x2 c = {1.0001f, 0.9999f}; d *= c; v[i] = d[0]; v[i + 1] = d[1]; }
        if (P == P_MAX) { v[i] = fmaxf(v[i], v[i + 1] + 1.f); v[i + 1] = fmaxf(v[i + 1], v[i]); }
        if (P == P_EXP) { v[i] = __builtin_amdgcn_exp2f(v[i]); v[i + 1] = __builtin_amdgcn_exp2f(v[i + 1]); }
        if (P == P_CVTPK) { half2v h = {(_Float16)v[i], (_Float16)v[i + 1]}; asm volatile("" : "+v"(h)); v[i] += (float)h[0]; v[i + 1] += (float)h[1]; }
        if (P == P_CVT16) { _Float16 h0 = (_Float16)v[i]; asm volatile("" : "+v"(h0)); v[i] = (float)h0 + 1.f; _Float16 h1 = (_Float16)v[i + 1]; asm volatile("" : "+v"(h1)); v[i + 1] = (float)h1 + 1.f; }
        if (P == P_DPP) { v[i] += __shfl_xor(v[i], 1, 64); v[i + 1] += __shfl_xor(v[i + 1], 16, 64); }
        if (P == P_BPERM) { v[i] += __shfl_xor(v[i], 32, 64); v[i + 1] += __shfl_xor(v[i + 1], 32, 64); }
        if (P == P_LDSREAD) { v[i] += lds[(lane * 4 + i + it) & 4095]; v[i + 1] += lds[(lane * 4 + i + 1 + it) & 4095]; }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
int main() {
  half8* in; float* out; unsigned long long* cyc;
  hipMalloc(&in, 64 * 2 * 16); hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 64);
  _Float16 h[64 * 2 * 8]; for (int i = 0; i < 64 * 2 * 8; ++i) h[i] = (_Float16)(((i * 2654435761u) >> 20) % 2001 / 1000.0f - 1.0f);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 400;
  auto run = [&](auto kern, auto alone, const char* name) {
    unsigned long long c[8], c2[8];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, in, out, cyc, iters); hipDeviceSynchronize(); }
    hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(alone, dim3(256), dim3(512), 0, 0, in, out, cyc, iters); hipDeviceSynchronize(); }
    hipMemcpy(c2, cyc, 64, hipMemcpyDeviceToHost);
    printf("partner %-28s MFMA wave %.1f cycles per MFMA; partner %.1f cycles per 8 ops (alone %.1f)\n", name, (double)c[0] / (iters * 8.0),
           (double)c[4] / (iters * 4.0), (double)c2[4] / (iters * 4.0));
  };
#define RUN(P, NAME) run(k<P, true>, k<P, false>, NAME)
  RUN(P_NONE, "none"); RUN(P_FMA, "v_fma_f32"); RUN(P_MUL, "v_mul_f32"); RUN(P_PKADD, "v_pk_add_f32"); RUN(P_PKMUL, "v_pk_mul_f32");
  RUN(P_MAX, "v_max_f32"); RUN(P_EXP, "v_exp_f32"); RUN(P_CVTPK, "cvt pk f16 + back"); RUN(P_CVT16, "cvt f16 scalar + back");
  RUN(P_DPP, "shfl xor 1 / 16"); RUN(P_BPERM, "shfl xor 32"); RUN(P_LDSREAD, "ds_read_b32");
  return 0;
}
