// Shared device helpers for the gfx950 (CDNA4) kernels of the OVMono3D-LIFT path.
// Wave = 64 lanes everywhere; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ovm {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Split-precision operand: x ~= hi + lo, hi = fp16(x), lo = fp16(x - hi) (22 significant bits; lo may be an
// fp16 subnormal - v_mfma_*_f16 honours fp16 denormal inputs on gfx950, verified on hardware).
// One-pass mode (precision "f16") uses hi only; three-pass mode ("f16x3") accumulates
//   acc += Ah*Wh + Al*Wh + Ah*Wl          (the Al*Wl term, ~2^-22 relative, is dropped)
// into ONE fp32 accumulator, so the split costs MFMA issue slots but no extra registers.
// Q is stored pre-multiplied by dh^-0.5 * log2(e) (dh = 64) so that attention probabilities are exp2(s - m).
constexpr float kQScale = 0.125f * 1.44269504088896340736f;
constexpr float kLoScale = 1.0f;
constexpr float kLoInv = 1.0f;

// f32 -> f16, round-to-nearest-even, fp16 subnormal results preserved. Written as the scalar
// v_cvt_f16_f32: when hipcc pairs two conversions into gfx950's v_cvt_pk_f16_f32 the subnormal results
// come back as zero (measured: the split's lo parts and tiny attention outputs were lost), which would
// silently cut the f16x3 mode back to ~fp16 accuracy for |x| < 2^-3.
// The leading s_nop covers the trans->VALU wait state (v_exp_f32 etc. feeding the conversion): hipcc pads
// hazards for its own instructions only, not for ones inside an asm string.
__device__ __forceinline__ half_t cvt_f16_rn(float x) {
  uint32_t r;
  asm volatile("s_nop 1\n\tv_cvt_f16_f32 %0, %1" : "=v"(r) : "v"(x));
  return __builtin_bit_cast(half_t, (uint16_t)(r & 0xFFFFu));
}

// Same conversion for operands that do NOT come straight out of a transcendental instruction (loads, adds, FMAs): no
// hazard nop needed, and not volatile, so the scheduler may interleave it with MFMAs and memory instructions.
__device__ __forceinline__ half_t cvt_f16_rn_nt(float x) {
  uint32_t r;
  asm("v_cvt_f16_f32 %0, %1" : "=v"(r) : "v"(x));
  return __builtin_bit_cast(half_t, (uint16_t)(r & 0xFFFFu));
}
__device__ __forceinline__ void split_f16_nt(float x, half_t& hi, half_t& lo) {
  hi = cvt_f16_rn_nt(x);
  lo = cvt_f16_rn_nt(x - (float)hi);
}

__device__ __forceinline__ void split_f16(float x, half_t& hi, half_t& lo) {
  hi = cvt_f16_rn(x);
  lo = cvt_f16_rn(x - (float)hi);
}

// 16-byte async global -> LDS copy (global_load_lds_dwordx4). LDS destination is
// wave-uniform base + lane*16; the per-lane part lives in the global source address.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Column of logical element k inside a row of an interleaved split image [k/32][hi 32 | lo 32] (the lo array starts 32 halves
// after the hi array, so the same offset addresses both parts); row stride = 2 K.
__host__ __device__ __forceinline__ int il_col(int k) { return ((k >> 5) << 6) | (k & 31); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware remap of a linear workgroup id: blocks that share an XCD (id % 8 under the
// observed round-robin placement) get a contiguous chunk of the tile space. Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

}  // namespace ovm

#ifndef OVM_OK
#define OVM_OK 0
#define OVM_ERR_INVALID (-1)
#define OVM_ERR_HIP (-2)
#define OVM_ERR_MISSING_WEIGHT (-3)
#define OVM_ERR_SHAPE (-4)
#define OVM_ERR_CAPACITY (-5)
#endif
