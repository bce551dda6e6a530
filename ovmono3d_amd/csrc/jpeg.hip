// Data feeding on the device ("next" row 2 of SURVEY.md 8f): JPEG decode split at the only serial step.
//
// The reference reads its images with cv2.imread (demo/demo.py:52) / detectron2's read_image = Pillow
// (cubercnn/data/dataset_mapper.py:38), i.e. with libjpeg-turbo at its defaults: JDCT_ISLOW, fancy upsampling, YCbCr -> RGB through
// the integer tables of jdcolor.c. A baseline JPEG is (1) a Huffman-coded stream of quantised DCT coefficients - inherently serial,
// decoded here on the host (`ovm_host_jpeg_entropy_decode`, jpeg_host.hpp; progressive files add their scans' bands and bit planes to the same coefficient planes) - and (2) dequantisation, an 8 x 8 inverse DCT per block, chroma
// upsampling and a colour transform per pixel - data parallel, all integer arithmetic, done here on the device
// (`ovm_jpeg_reconstruct`) so that the image is born in HBM as the [H][W][3] uint8 tensor the resize kernel (resize.hip) and the
// patch gather consume; what crosses PCIe is the coefficient planes (int16, mostly zero).
//
// The arithmetic is restated from the published algorithms of libjpeg-turbo 3.x (source absent from the container; the library
// itself is present inside Pillow, which is the live oracle of the tests: results are bit-identical):
//   jidctint.c  jpeg_idct_islow  - LL&M 13-bit fixed-point IDCT, two passes, descale by 11 / 18 bits, range limit around 128
//   jdsample.c  h2v1_fancy / h2v2_fancy upsampling - triangle filter 3/4 : 1/4 with the alternating rounding bias,
//               edge columns replicated, rows above the first / below the last real row replicated (jdmainct.c context rows);
//               plain replication when the downsampled width is <= 2
//   jdcolor.c   build_ycc_rgb_table / ycc_rgb_convert - R = Y + (91881 Cr' + 2^15 >> 16), B = Y + (116130 Cb' + 2^15 >> 16),
//               G = Y + ((-22554 Cb' - 46802 Cr' + 2^15) >> 16), Cb' = Cb - 128, clamped to 0..255
//   jdapimin.c  default_decompress_parms - colour space from the JFIF / Adobe markers or the component ids
// Scope: 8-bit Huffman streams - baseline, extended-sequential and progressive (SOF0 / SOF1 / SOF2; a progression must be complete:
// libjpeg smooths an incomplete one across blocks, which is not reproduced) -, 1 component or 3 components with chroma at 1 x 1 and
// luma at 1x1, 2x1 or 2x2, interleaved or one scan per component, restart intervals. Arithmetic-coded, 12-bit,
// 4-component and odd sampling layouts return OVM_ERR_UNSUPPORTED from ovm_host_jpeg_info (the caller's other decoder handles them,
// as it handles PNG); a truncated or corrupt stream returns OVM_ERR_INVALID.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>

#include "../../include/ovm3d.h"
#include "jpeg_host.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------------------
struct JpegDev {
  int width, height, ncomp, colorspace;
  int hs, vs;                       // luma sampling relative to chroma (1 or 2 each)
  int bw[3], bh[3], cw[3], ch[3];
  int boff[3];                      // first block of each component
  int qidx[3];
  uint16_t qt[4][64];
};

#define OVM_FIX_0_298631336 2446
#define OVM_FIX_0_390180644 3196
#define OVM_FIX_0_541196100 4433
#define OVM_FIX_0_765366865 6270
#define OVM_FIX_0_899976223 7373
#define OVM_FIX_1_175875602 9633
#define OVM_FIX_1_501321110 12299
#define OVM_FIX_1_847759065 15137
#define OVM_FIX_1_961570560 16069
#define OVM_FIX_2_053119869 16819
#define OVM_FIX_2_562915447 20995
#define OVM_FIX_3_072711026 25172

// one 1-D pass of jpeg_idct_islow on (i0..i7); o0..o7 = the eight outputs before the descale
__device__ __forceinline__ void idct_1d(int i0, int i1, int i2, int i3, int i4, int i5, int i6, int i7, int* o) {
  int z1 = (i2 + i6) * OVM_FIX_0_541196100;
  int tmp2 = z1 + i6 * (-OVM_FIX_1_847759065);
  int tmp3 = z1 + i2 * OVM_FIX_0_765366865;
  int tmp0 = (i0 + i4) << 13;
  int tmp1 = (i0 - i4) << 13;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = i7; tmp1 = i5; tmp2 = i3; tmp3 = i1;
  z1 = tmp0 + tmp3;
  int z2 = tmp1 + tmp2, z3 = tmp0 + tmp2, z4 = tmp1 + tmp3;
  const int z5 = (z3 + z4) * OVM_FIX_1_175875602;
  tmp0 *= OVM_FIX_0_298631336; tmp1 *= OVM_FIX_2_053119869; tmp2 *= OVM_FIX_3_072711026; tmp3 *= OVM_FIX_1_501321110;
  z1 *= -OVM_FIX_0_899976223; z2 *= -OVM_FIX_2_562915447; z3 *= -OVM_FIX_1_961570560; z4 *= -OVM_FIX_0_390180644;
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  o[0] = tmp10 + tmp3; o[7] = tmp10 - tmp3;
  o[1] = tmp11 + tmp2; o[6] = tmp11 - tmp2;
  o[2] = tmp12 + tmp1; o[5] = tmp12 - tmp1;
  o[3] = tmp13 + tmp0; o[4] = tmp13 - tmp0;
}

// one thread per 8 x 8 block: dequantise, columns (descale 11), rows (descale 18), + 128, clamp
__global__ __launch_bounds__(64) void jpeg_idct_kernel(const int16_t* __restrict__ coef, JpegDev J, int nblocks, uint8_t* __restrict__ planes) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  int c = 0;
  if (J.ncomp == 3) c = b >= J.boff[2] ? 2 : (b >= J.boff[1] ? 1 : 0);
  const int lb = b - J.boff[c];
  const int by = lb / J.bw[c], bx = lb - by * J.bw[c];
  const uint16_t* q = J.qt[J.qidx[c]];
  const uint4* src = (const uint4*)(coef + (size_t)b * 64);
  int in[64];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint4 v = src[i];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      in[i * 8 + 2 * j] = (int)(int16_t)(w[j] & 0xffff) * (int)q[i * 8 + 2 * j];
      in[i * 8 + 2 * j + 1] = (int)(int16_t)(w[j] >> 16) * (int)q[i * 8 + 2 * j + 1];
    }
  }
  int ws[64];
#pragma unroll
  for (int col = 0; col < 8; ++col) {
    int o[8];
    idct_1d(in[col], in[8 + col], in[16 + col], in[24 + col], in[32 + col], in[40 + col], in[48 + col], in[56 + col], o);
#pragma unroll
    for (int r = 0; r < 8; ++r) ws[r * 8 + col] = (o[r] + (1 << 10)) >> 11;
  }
  const size_t stride = (size_t)J.bw[c] * 8;
  uint8_t* dst = planes + (size_t)J.boff[c] * 64 + (size_t)by * 8 * stride + (size_t)bx * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    int o[8];
    idct_1d(ws[r * 8], ws[r * 8 + 1], ws[r * 8 + 2], ws[r * 8 + 3], ws[r * 8 + 4], ws[r * 8 + 5], ws[r * 8 + 6], ws[r * 8 + 7], o);
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      int v = ((o[x] + (1 << 17)) >> 18) + 128;
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      if (x < 4) lo |= (unsigned)v << (8 * x); else hi |= (unsigned)v << (8 * (x - 4));
    }
    *(uint2*)(dst + (size_t)r * stride) = make_uint2(lo, hi);
  }
}

__device__ __forceinline__ int chroma_at(const uint8_t* __restrict__ p, int stride, int cw, int ch, int hs, int vs, int x, int y) {
  if (hs == 1 && vs == 1) return p[(size_t)y * stride + x];
  if (hs == 2 && vs == 1) {                                        // h2v1
    const uint8_t* r = p + (size_t)y * stride;
    const int i = x >> 1;
    if (cw <= 2) return r[i];
    if (!(x & 1)) return i == 0 ? r[0] : (3 * r[i] + r[i - 1] + 1) >> 2;
    return i == cw - 1 ? r[i] : (3 * r[i] + r[i + 1] + 2) >> 2;
  }
  const int rr = y >> 1;
  int rn = (y & 1) ? rr + 1 : rr - 1;
  rn = rn < 0 ? 0 : (rn > ch - 1 ? ch - 1 : rn);
  const uint8_t* r0 = p + (size_t)rr * stride;
  const uint8_t* r1 = p + (size_t)rn * stride;
  const int i = x >> 1;                                                       // h2v2
  if (cw <= 2) return r0[i];
  const int cs = 3 * r0[i] + r1[i];
  if (!(x & 1)) return i == 0 ? (4 * cs + 8) >> 4 : (3 * cs + 3 * r0[i - 1] + r1[i - 1] + 8) >> 4;
  return i == cw - 1 ? (4 * cs + 7) >> 4 : (3 * cs + 3 * r0[i + 1] + r1[i + 1] + 7) >> 4;
}

__device__ __forceinline__ int clamp8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// one thread per output pixel: upsample the chroma, colour transform, [H][W][3] RGB
__global__ void jpeg_color_kernel(const uint8_t* __restrict__ planes, JpegDev J, uint8_t* __restrict__ rgb) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= J.width) return;
  const int s0 = J.bw[0] * 8;
  const int Y = planes[(size_t)y * s0 + x];
  uint8_t* o = rgb + ((size_t)y * J.width + x) * 3;
  if (J.ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; return; }
  int c1, c2;
  if (J.colorspace == 2 && J.hs == 1 && J.vs == 1) {
    c1 = planes[(size_t)J.boff[1] * 64 + (size_t)y * J.bw[1] * 8 + x]; c2 = planes[(size_t)J.boff[2] * 64 + (size_t)y * J.bw[2] * 8 + x];
  } else {
    c1 = chroma_at(planes + (size_t)J.boff[1] * 64, J.bw[1] * 8, J.cw[1], J.ch[1], J.hs, J.vs, x, y);
    c2 = chroma_at(planes + (size_t)J.boff[2] * 64, J.bw[2] * 8, J.cw[2], J.ch[2], J.hs, J.vs, x, y);
  }
  if (J.colorspace == 2) { o[0] = (uint8_t)Y; o[1] = (uint8_t)c1; o[2] = (uint8_t)c2; return; }
  const int cb = c1 - 128, cr = c2 - 128;
  o[0] = (uint8_t)clamp8(Y + ((91881 * cr + 32768) >> 16));
  o[1] = (uint8_t)clamp8(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
  o[2] = (uint8_t)clamp8(Y + ((116130 * cb + 32768) >> 16));
}

}  // namespace

extern "C" {

int ovm_host_jpeg_info(const uint8_t* data, size_t n, OvmJpegInfo* info) { return ovm_jpeg::host_info(data, n, info); }

int ovm_host_jpeg_entropy_decode(const uint8_t* data, size_t n, int16_t* coef, int64_t coef_capacity, OvmJpegInfo* info) {
  return ovm_jpeg::host_entropy_decode(data, n, coef, coef_capacity, info);
}

int ovm_jpeg_reconstruct(const int16_t* coef, const OvmJpegInfo* info, uint8_t* planes, uint8_t* rgb, ovm_stream_t stream) {
  if (!coef || !info || !planes || !rgb) return OVM_ERR_INVALID;
  const OvmJpegInfo& I = *info;
  if ((I.ncomp != 1 && I.ncomp != 3) || I.width <= 0 || I.height <= 0 || I.coef_blocks <= 0) return OVM_ERR_INVALID;
  JpegDev J; memset(&J, 0, sizeof(J));
  J.width = I.width; J.height = I.height; J.ncomp = I.ncomp; J.colorspace = I.colorspace; J.hs = I.hmax; J.vs = I.vmax;
  if (J.hs < 1 || J.hs > 2 || J.vs < 1 || J.vs > 2 || (J.hs == 1 && J.vs == 2)) return OVM_ERR_INVALID;
  int off = 0;
  for (int c = 0; c < I.ncomp; ++c) {
    if (I.bw[c] <= 0 || I.bh[c] <= 0 || I.qidx[c] < 0 || I.qidx[c] > 3) return OVM_ERR_INVALID;
    if (I.cw[c] > I.bw[c] * 8 || I.ch[c] > I.bh[c] * 8) return OVM_ERR_INVALID;
    J.bw[c] = I.bw[c]; J.bh[c] = I.bh[c]; J.cw[c] = I.cw[c]; J.ch[c] = I.ch[c]; J.qidx[c] = I.qidx[c]; J.boff[c] = off;
    off += I.bw[c] * I.bh[c];
  }
  if (off != I.coef_blocks || I.cw[0] != I.width || I.ch[0] != I.height) return OVM_ERR_INVALID;
  memcpy(J.qt, I.qt, sizeof(J.qt));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((off + 63) / 64), dim3(64), 0, s, coef, J, off, planes);
  hipLaunchKernelGGL(jpeg_color_kernel, dim3((I.width + 255) / 256, I.height), dim3(256), 0, s, planes, J, rgb);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

}  // extern "C"
