// Data feeding on the device ("next" row 2 of SURVEY.md 8f): the uint8 bilinear resize behind detectron2's
// `T.ResizeShortestEdge` (reference demo/demo.py:79-83, cubercnn/data/dataset_mapper.py:62-72), which for uint8 images is
// Pillow's `Image.resize(size, BILINEAR)`. Pillow's algorithm (libImaging/Resample.c, restated from the published source) is
// a separable convolution in fixed point: per output coordinate a window [xmin, xmin+n) of triangle-filter weights (support
// widened by the scale factor when shrinking = antialiasing), normalised in double, rounded to 22-bit integers; a horizontal
// pass to a uint8 intermediate, then a vertical pass; each output = clip8((2^21 + sum pixel*coef) >> 22). The coefficient
// tables are built on the host in double exactly as Pillow does, the two passes run here in integer arithmetic, so the
// result is bit-identical to Pillow's.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <vector>

#include "kernels.hpp"
#include "common.hpp"
#include "../../include/ovm3d.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v) {
  v >>= kPrecisionBits;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// out[y][xx][c] = clip8(2^21 + sum_x in[y][xmin+x][c] * k[xx][x]);  one thread per output byte
__global__ void resize_h_kernel(const uint8_t* __restrict__ in, long sy, long sx, long sc, int H, int C, int outW, const int* __restrict__ bounds,
                                const int* __restrict__ kk, int ksize, uint8_t* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)H * outW * C) return;
  const int c = (int)(i % C); const int xx = (int)((i / C) % outW); const int y = (int)(i / ((long)C * outW));
  const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
  const int* k = kk + (size_t)xx * ksize;
  int ss = 1 << (kPrecisionBits - 1);
  const uint8_t* p = in + (size_t)y * sy + (size_t)xmin * sx + (size_t)c * sc;
  for (int x = 0; x < n; ++x) ss += (int)p[(size_t)x * sx] * k[x];
  out[i] = clip8(ss);
}

// out[yy][x][c] = clip8(2^21 + sum_y tmp[ymin+y][x][c] * k[yy][y]);  tmp and out are dense [rows][W][C]
__global__ void resize_v_kernel(const uint8_t* __restrict__ tmp, int W, int C, int outH, const int* __restrict__ bounds, const int* __restrict__ kk,
                                int ksize, uint8_t* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row = (long)W * C;
  if (i >= (long)outH * row) return;
  const int yy = (int)(i / row); const long xc = i - (long)yy * row;
  const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
  const int* k = kk + (size_t)yy * ksize;
  int ss = 1 << (kPrecisionBits - 1);
  const uint8_t* p = tmp + (size_t)ymin * row + xc;
  for (int y = 0; y < n; ++y) ss += (int)p[(size_t)y * row] * k[y];
  out[i] = clip8(ss);
}

inline double bilinear_filter(double x) {
  if (x < 0.0) x = -x;
  return x < 1.0 ? 1.0 - x : 0.0;
}

}  // namespace

extern "C" {

// Pillow precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter over the full input range.
// bounds: 2*out_size ints (first input index, count); coefs: out_size * ksize ints; returns ksize, or <0 on error.
int ovm_host_pil_bilinear_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* coefs, int32_t coefs_capacity) {
  if (in_size <= 0 || out_size <= 0) return OVM_ERR_INVALID;
  const double support0 = 1.0;
  const double scale = (double)in_size / out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = support0 * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  if (!bounds || !coefs) return ksize;                              // size query
  if ((long)out_size * ksize > coefs_capacity) return OVM_ERR_INVALID;
  std::vector<double> k((size_t)ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int x;
    for (x = 0; x < xmax; ++x) {
      const double w = bilinear_filter((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    for (; x < ksize; ++x) k[x] = 0.0;
    bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
    for (x = 0; x < ksize; ++x)
      coefs[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << kPrecisionBits)) : (int)(0.5 + k[x] * (1 << kPrecisionBits));
  }
  return ksize;
}

// src: uint8, element strides (sy, sx, sc) over [H][W][C]; tmp: dense [H][outW][C]; dst: dense [outH][outW][C].
// A pass whose size does not change is skipped as Pillow does (tmp may then be null if neither or only one pass runs... it is
// only needed when both run).
int ovm_resize_bilinear_u8(const uint8_t* src, int32_t H, int32_t W, int32_t C, int64_t sy, int64_t sx, int64_t sc, int32_t outH, int32_t outW,
                           const int32_t* xbounds, const int32_t* xcoefs, int32_t xksize, const int32_t* ybounds, const int32_t* ycoefs,
                           int32_t yksize, uint8_t* tmp, uint8_t* dst, ovm_stream_t stream) {
  if (!src || !dst || H <= 0 || W <= 0 || C <= 0 || outH <= 0 || outW <= 0) return OVM_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  const bool need_h = outW != W, need_v = outH != H;
  if (need_h && (!xbounds || !xcoefs)) return OVM_ERR_INVALID;
  if (need_v && (!ybounds || !ycoefs)) return OVM_ERR_INVALID;
  if (need_h && need_v && !tmp) return OVM_ERR_INVALID;
  const int bs = 256;
  if (need_h) {
    uint8_t* o = need_v ? tmp : dst;
    const long n = (long)H * outW * C;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, s, src, (long)sy, (long)sx, (long)sc, H, C, outW, xbounds, xcoefs,
                       xksize, o);
  }
  if (need_v) {
    const long n = (long)outH * outW * C;
    if (need_h) {
      hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, s, tmp, outW, C, outH, ybounds, ycoefs, yksize, dst);
    } else {
      if (!(sc == 1 && sx == C && sy == (int64_t)W * C)) return OVM_ERR_INVALID;       // vertical-only pass reads a dense image
      hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, s, src, W, C, outH, ybounds, ycoefs, yksize, dst);
    }
  }
  if (!need_h && !need_v) {
    if (!(sc == 1 && sx == C && sy == (int64_t)W * C)) return OVM_ERR_INVALID;
    if (hipMemcpyAsync(dst, src, (size_t)H * W * C, hipMemcpyDeviceToDevice, s) != hipSuccess) return OVM_ERR_HIP;
  }
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int ovm_resize_bilinear_f32(const float* src, int32_t B, int32_t H, int32_t W, int32_t outH, int32_t outW, float* dst, ovm_stream_t stream) {
  if (!src || !dst) return OVM_ERR_INVALID;
  return ovm::launch_resize_bilinear_f32(src, B, H, W, outH, outW, dst, (hipStream_t)stream);
}

}  // extern "C"
