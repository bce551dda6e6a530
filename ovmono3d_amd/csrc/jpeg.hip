// Data feeding on the device ("next" row 2 of SURVEY.md 8f): JPEG decode split at the only serial step.
//
// The reference reads its images with cv2.imread (demo/demo.py:52) / detectron2's read_image = Pillow
// (cubercnn/data/dataset_mapper.py:38), i.e. with libjpeg-turbo at its defaults: JDCT_ISLOW, fancy upsampling, YCbCr -> RGB through
// the integer tables of jdcolor.c. A baseline JPEG is (1) a Huffman-coded stream of quantised DCT coefficients - inherently serial,
// decoded here on the host (`ovm_host_jpeg_entropy_decode`) - and (2) dequantisation, an 8 x 8 inverse DCT per block, chroma
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
// Scope: 8-bit baseline / extended-sequential Huffman streams (SOF0 / SOF1), 1 component or 3 components with chroma at 1 x 1 and
// luma at 1x1, 2x1 or 2x2, interleaved or one scan per component, restart intervals. Progressive, arithmetic-coded, 12-bit,
// 4-component and odd sampling layouts return OVM_ERR_UNSUPPORTED from ovm_host_jpeg_info (the caller's other decoder handles them,
// as it handles PNG); a truncated or corrupt stream returns OVM_ERR_INVALID.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>

#include "../../include/ovm3d.h"

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool present = false;
  uint8_t bits[17] = {0}, vals[256] = {0};
  // canonical decoding: codes of length l occupy [mincode[l], maxcode[l]]; fast table for codes of <= 9 bits
  int maxcode[18], valptr[17], mincode[17];
  uint16_t fast[512];              // (length << 8) | symbol, 0 = longer than 9 bits
  int16_t fast_ac[512];            // AC tables: (value << 8) | (run << 4) | (code + magnitude bits) when both fit in the 9-bit window, else 0
  void build_fast_ac() {
    for (int i = 0; i < 512; ++i) {
      fast_ac[i] = 0;
      const uint16_t f = fast[i];
      if (!f) continue;
      const int len = f >> 8, rs = f & 255, run = rs >> 4, mag = rs & 15;
      if (mag == 0 || len + mag > 9) continue;
      int k = ((i << len) & 511) >> (9 - mag);                     // the magnitude bits behind the code
      if (k < (1 << (mag - 1))) k += (int)((~0u) << mag) + 1;      // extend()
      if (k >= -128 && k <= 127) fast_ac[i] = (int16_t)(k * 256 + run * 16 + len + mag);
    }
  }
  bool build() {
    int code = 0, k = 0;
    memset(fast, 0, sizeof(fast));
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k; mincode[l] = code;
      for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
        if (k >= 256) return false;
        if (l <= 9) {
          const int lo = code << (9 - l), n = 1 << (9 - l);
          if (lo + n > 512) return false;
          for (int j = 0; j < n; ++j) fast[lo + j] = (uint16_t)((l << 8) | vals[k]);
        }
      }
      maxcode[l] = bits[l] ? code - 1 : -1;
      if (code > (1 << l)) return false;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    return true;
  }
};

struct BitReader {
  const uint8_t* p; const uint8_t* end;
  uint64_t acc = 0; int n = 0;         // n valid bits at the bottom of acc
  bool hit_marker = false;
  void fill() {
    if (!hit_marker && p + 8 <= end) {                             // whole bytes at once while no 0xFF is near
      uint64_t v; memcpy(&v, p, 8);
      const uint64_t nv = ~v;
      if (!((nv - 0x0101010101010101ull) & v & 0x8080808080808080ull)) {     // no byte of v is 0xFF
        const int k = (64 - n) >> 3;
        if (k > 0) {
          const uint64_t be = __builtin_bswap64(v);
          acc = (k == 8) ? be : ((acc << (8 * k)) | (be >> (64 - 8 * k)));
          p += k; n += 8 * k;
        }
        return;
      }
    }
    while (n <= 56) {
      uint8_t b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) p += 2;            // stuffed byte
          else { hit_marker = true; b = 0; }                   // a marker ends the segment: zeros from here on (as libjpeg)
        } else ++p;
      } else hit_marker = hit_marker || p >= end;
      acc = (acc << 8) | b; n += 8;
    }
  }
  inline int peek(int k) { if (n < k) fill(); return (int)((acc >> (n - k)) & ((1u << k) - 1)); }
  inline void skip(int k) { n -= k; }
  inline int get(int k) { const int v = peek(k); n -= k; return v; }
  void reset() { acc = 0; n = 0; }
};

inline int decode_symbol(BitReader& br, const Huff& h) {
  const int look = br.peek(9);
  const uint16_t f = h.fast[look];
  if (f) { br.skip(f >> 8); return f & 255; }
  int code = br.peek(16), l;
  for (l = 10; l <= 16; ++l)
    if ((code >> (16 - l)) <= h.maxcode[l] && h.bits[l]) break;
  if (l > 16) return -1;
  const int c = code >> (16 - l);
  if (c < h.mincode[l]) return -1;
  br.skip(l);
  return h.vals[h.valptr[l] + c - h.mincode[l]];
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Frame {
  OvmJpegInfo info;
  int cid[3] = {0, 0, 0};
  size_t plane_off[3] = {0, 0, 0};      // in blocks
  int restart = 0;
  Huff dc[4], ac[4];
  bool have_sof = false, jfif = false, adobe = false; int adobe_transform = -1;
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// walks the marker segments up to (not into) the first SOS when `scan_cb` is null, else decodes every scan
int parse(const uint8_t* d, size_t n, Frame& f, int16_t* coef) {
  if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return OVM_ERR_INVALID;
  memset(&f.info, 0, sizeof(f.info));
  size_t pos = 2;
  int dcpred[3];
  while (pos + 4 <= n) {
    if (d[pos] != 0xFF) return OVM_ERR_INVALID;
    while (pos < n && d[pos] == 0xFF) ++pos;              // fill bytes
    if (pos >= n) return OVM_ERR_INVALID;
    const int m = d[pos++];
    if (m == 0xD9) break;                                 // EOI
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;  // standalone
    if (pos + 2 > n) return OVM_ERR_INVALID;
    const int len = be16(d + pos);
    if (len < 2 || pos + len > n) return OVM_ERR_INVALID;
    const uint8_t* s = d + pos + 2; const int sl = len - 2;
    if (m == 0xC0 || m == 0xC1) {
      if (f.have_sof || sl < 6) return OVM_ERR_INVALID;
      if (s[0] != 8) return OVM_ERR_UNSUPPORTED;
      OvmJpegInfo& I = f.info;
      I.height = be16(s + 1); I.width = be16(s + 3); I.ncomp = s[5];
      if (I.height <= 0 || I.width <= 0) return OVM_ERR_UNSUPPORTED;      // (height 0 = DNL-defined: not handled)
      if (I.ncomp != 1 && I.ncomp != 3) return OVM_ERR_UNSUPPORTED;
      if (sl < 6 + 3 * I.ncomp) return OVM_ERR_INVALID;
      for (int c = 0; c < I.ncomp; ++c) {
        f.cid[c] = s[6 + 3 * c]; I.h[c] = s[7 + 3 * c] >> 4; I.v[c] = s[7 + 3 * c] & 15; I.qidx[c] = s[8 + 3 * c];
        if (I.h[c] < 1 || I.h[c] > 4 || I.v[c] < 1 || I.v[c] > 4 || I.qidx[c] > 3) return OVM_ERR_INVALID;
      }
      if (I.ncomp == 1) { I.h[0] = I.v[0] = 1; }          // a single component is never subsampled (libjpeg: MCU = one block)
      else {
        if (I.h[1] != 1 || I.v[1] != 1 || I.h[2] != 1 || I.v[2] != 1 || I.h[0] > 2 || I.v[0] > 2) return OVM_ERR_UNSUPPORTED;
        if (I.h[0] == 1 && I.v[0] == 2) return OVM_ERR_UNSUPPORTED;       // 4:4:0 (h1v2): Pillow cannot write it, so it could not be pinned
      }
      I.hmax = I.h[0]; I.vmax = I.v[0];
      const int mcux = (I.width + 8 * I.hmax - 1) / (8 * I.hmax), mcuy = (I.height + 8 * I.vmax - 1) / (8 * I.vmax);
      size_t off = 0;
      for (int c = 0; c < I.ncomp; ++c) {
        I.bw[c] = mcux * I.h[c]; I.bh[c] = mcuy * I.v[c];
        I.cw[c] = (I.width * I.h[c] + I.hmax - 1) / I.hmax; I.ch[c] = (I.height * I.v[c] + I.vmax - 1) / I.vmax;
        f.plane_off[c] = off; off += (size_t)I.bw[c] * I.bh[c];
      }
      if (off > (size_t)0x7fffffff / 64) return OVM_ERR_UNSUPPORTED;
      I.coef_blocks = (int32_t)off;
      f.have_sof = true;
    } else if (m == 0xC4) {
      int o = 0;
      while (o + 17 <= sl) {
        const int tc = s[o] >> 4, th = s[o] & 15;
        if (tc > 1 || th > 3) return OVM_ERR_INVALID;
        Huff& h = tc ? f.ac[th] : f.dc[th];
        int cnt = 0;
        h.bits[0] = 0;
        for (int l = 1; l <= 16; ++l) { h.bits[l] = s[o + l]; cnt += h.bits[l]; }
        if (cnt > 256 || o + 17 + cnt > sl) return OVM_ERR_INVALID;
        memcpy(h.vals, s + o + 17, cnt);
        if (!h.build()) return OVM_ERR_INVALID;
        if (tc) h.build_fast_ac();
        h.present = true;
        o += 17 + cnt;
      }
    } else if (m == 0xDB) {
      int o = 0;
      while (o < sl) {
        const int pq = s[o] >> 4, tq = s[o] & 15;
        if (tq > 3 || pq > 1 || o + 1 + 64 * (pq + 1) > sl) return OVM_ERR_INVALID;
        for (int i = 0; i < 64; ++i) f.info.qt[tq][kZigzag[i]] = pq ? (uint16_t)be16(s + o + 1 + 2 * i) : s[o + 1 + i];
        o += 1 + 64 * (pq + 1);
      }
    } else if (m == 0xDD) {
      if (sl < 2) return OVM_ERR_INVALID;
      f.restart = be16(s);
    } else if (m == 0xE0) {
      if (sl >= 5 && !memcmp(s, "JFIF", 5)) f.jfif = true;
    } else if (m == 0xEE) {
      if (sl >= 12 && !memcmp(s, "Adobe", 5)) { f.adobe = true; f.adobe_transform = s[11]; }
    } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
      return OVM_ERR_UNSUPPORTED;                          // progressive, lossless, differential, arithmetic
    } else if (m == 0xCC) {
      return OVM_ERR_UNSUPPORTED;                          // arithmetic conditioning
    } else if (m == 0xDA) {
      if (!f.have_sof) return OVM_ERR_INVALID;
      if (!coef) break;                                    // header walk ends here
      const OvmJpegInfo& I = f.info;
      if (sl < 1) return OVM_ERR_INVALID;
      const int ns = s[0];
      if (ns < 1 || ns > I.ncomp || sl < 1 + 2 * ns + 3) return OVM_ERR_INVALID;
      int sc[3], td[3], ta[3];
      for (int i = 0; i < ns; ++i) {
        int c = -1;
        for (int q = 0; q < I.ncomp; ++q) if (f.cid[q] == s[1 + 2 * i]) c = q;
        if (c < 0) return OVM_ERR_INVALID;
        sc[i] = c; td[i] = s[2 + 2 * i] >> 4; ta[i] = s[2 + 2 * i] & 15;
        if (td[i] > 3 || ta[i] > 3 || !f.dc[td[i]].present || !f.ac[ta[i]].present) return OVM_ERR_INVALID;
      }
      if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return OVM_ERR_UNSUPPORTED;   // spectral selection = progressive
      BitReader br; br.p = d + pos + len; br.end = d + n;
      int mx, my;                                          // MCU grid of this scan
      if (ns == 1) { mx = (I.cw[sc[0]] + 7) / 8; my = (I.ch[sc[0]] + 7) / 8; }
      else { mx = I.bw[0] / I.h[0]; my = I.bh[0] / I.v[0]; }
      dcpred[0] = dcpred[1] = dcpred[2] = 0;
      int until_restart = f.restart ? f.restart : -1, next_rst = 0;
      for (int yy = 0; yy < my; ++yy)
        for (int xx = 0; xx < mx; ++xx) {
          if (until_restart == 0) {
            // byte-align, expect RSTn
            br.reset();
            const uint8_t* q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
              if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) break;
              ++q;
            }
            if (q + 1 >= br.end || !(q[0] == 0xFF && q[1] == 0xD0 + next_rst)) return OVM_ERR_INVALID;
            br.p = q + 2; br.hit_marker = false;
            next_rst = (next_rst + 1) & 7;
            dcpred[0] = dcpred[1] = dcpred[2] = 0;
            until_restart = f.restart;
          }
          for (int i = 0; i < ns; ++i) {
            const int c = sc[i];
            const int nh = ns == 1 ? 1 : I.h[c], nv = ns == 1 ? 1 : I.v[c];
            const Huff& hd = f.dc[td[i]]; const Huff& ha = f.ac[ta[i]];
            for (int v = 0; v < nv; ++v)
              for (int h = 0; h < nh; ++h) {
                const int by = yy * nv + v, bx = xx * nh + h;
                int16_t* blk = coef + (f.plane_off[c] + (size_t)by * I.bw[c] + bx) * 64;
                int sym = decode_symbol(br, hd);
                if (sym < 0 || sym > 11) return OVM_ERR_INVALID;
                if (sym) dcpred[c] += extend(br.get(sym), sym);
                blk[0] = (int16_t)dcpred[c];
                for (int k = 1; k < 64;) {
                  const int fa = ha.fast_ac[br.peek(9)];
                  if (fa) {                                            // code and magnitude in one lookup
                    k += (fa >> 4) & 15;
                    if (k > 63) return OVM_ERR_INVALID;
                    br.skip(fa & 15);
                    blk[kZigzag[k++]] = (int16_t)(fa >> 8);
                    continue;
                  }
                  sym = decode_symbol(br, ha);
                  if (sym < 0) return OVM_ERR_INVALID;
                  const int r = sym >> 4, sz = sym & 15;
                  if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                  k += r;
                  if (k > 63) return OVM_ERR_INVALID;
                  blk[kZigzag[k]] = (int16_t)extend(br.get(sz), sz);
                  ++k;
                }
              }
          }
          if (until_restart > 0) --until_restart;
        }
      // resume the marker walk behind the entropy-coded segment
      const uint8_t* q = br.p;
      if (!br.hit_marker) {
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) ++q;
      }
      pos = (size_t)(q - d);
      continue;
    }
    pos += len;
  }
  if (!f.have_sof) return OVM_ERR_INVALID;
  OvmJpegInfo& I = f.info;
  if (I.ncomp == 1) I.colorspace = 0;
  else if (f.jfif) I.colorspace = 1;
  else if (f.adobe) I.colorspace = f.adobe_transform == 0 ? 2 : 1;
  else I.colorspace = (f.cid[0] == 'R' && f.cid[1] == 'G' && f.cid[2] == 'B') ? 2 : 1;
  for (int c = 0; c < I.ncomp; ++c) {
    bool any = false;
    for (int i = 0; i < 64; ++i) any = any || I.qt[I.qidx[c]][i] != 0;
    if (!any) return OVM_ERR_INVALID;                      // quantisation table never defined
  }
  return OVM_OK;
}

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

int ovm_host_jpeg_info(const uint8_t* data, size_t n, OvmJpegInfo* info) {
  if (!data || !info) return OVM_ERR_INVALID;
  Frame f;
  const int r = parse(data, n, f, nullptr);
  if (r) return r;
  *info = f.info;
  return OVM_OK;
}

int ovm_host_jpeg_entropy_decode(const uint8_t* data, size_t n, int16_t* coef, int64_t coef_capacity, OvmJpegInfo* info) {
  if (!data || !coef || !info) return OVM_ERR_INVALID;
  Frame f;
  int r = parse(data, n, f, nullptr);
  if (r) return r;
  if (coef_capacity < (int64_t)f.info.coef_blocks * 64) return OVM_ERR_CAPACITY;
  memset(coef, 0, sizeof(int16_t) * 64 * (size_t)f.info.coef_blocks);
  Frame g;
  r = parse(data, n, g, coef);
  if (r) return r;
  *info = g.info;
  return OVM_OK;
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
