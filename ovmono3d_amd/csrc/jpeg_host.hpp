// Host half of the JPEG decode (see jpeg.hip): marker walk and Huffman decode of baseline / extended-sequential streams into
// coefficient planes. Plain C++ (no HIP) so that the CPU build can be fuzzed under AddressSanitizer (scratch/fuzz_jpeg.cpp).
#pragma once
#include <cstdint>
#include <cstring>

#include "../../include/ovm3d.h"

namespace ovm_jpeg {

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
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
  int pad = 0;                         // how many of them (the lowest) are phantom zeros appended behind a marker / the end of the data
  bool hit_marker = false;
  // The look-ahead of peek() may run into the phantom bits; CONSUMING one means the entropy-coded data ended before the scan did:
  // a truncated or cut file. libjpeg pads such a stream with zeros under a warning, Pillow (the reference's reader) raises on it -
  // so does this decoder (checked once per block by the scan loop).
  inline bool exhausted() const { return n < pad; }
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
      bool real = false;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) { p += 2; real = true; }   // stuffed byte
          else { hit_marker = true; b = 0; }                           // a marker ends the segment: zeros from here on (as libjpeg)
        } else { ++p; real = true; }
      } else hit_marker = hit_marker || p >= end;
      if (!real) pad += 8;
      acc = (acc << 8) | b; n += 8;
    }
  }
  inline int peek(int k) { if (n < k) fill(); return (int)((acc >> (n - k)) & ((1u << k) - 1)); }
  inline void skip(int k) { n -= k; }
  inline int get(int k) { const int v = peek(k); n -= k; return v; }
  void reset() { acc = 0; n = 0; pad = 0; }
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

// ---- progressive mode (SOF2; ITU T.81 Annex G as libjpeg's jdphuff.c implements it): every scan adds a band of coefficients
// (spectral selection Ss..Se) at a bit position (successive approximation Ah -> Al) to the same coefficient planes ----
inline int prog_dc_first(BitReader& br, const Huff& hd, int16_t* blk, int& pred, int Al) {
  const int sym = decode_symbol(br, hd);
  if (sym < 0 || sym > 11) return OVM_ERR_INVALID;
  if (sym) pred += extend(br.get(sym), sym);
  blk[0] = (int16_t)(pred * (1 << Al));
  return OVM_OK;
}

inline int prog_ac_first(BitReader& br, const Huff& ha, int16_t* blk, int Ss, int Se, int Al, int& eobrun) {
  if (eobrun > 0) { --eobrun; return OVM_OK; }            // inside a run of blocks whose band is all zero
  for (int k = Ss; k <= Se; ++k) {
    const int rs = decode_symbol(br, ha);
    if (rs < 0) return OVM_ERR_INVALID;
    const int r = rs >> 4, sz = rs & 15;
    if (sz) {
      k += r;
      if (k > Se) return OVM_ERR_INVALID;
      blk[kZigzag[k]] = (int16_t)(extend(br.get(sz), sz) * (1 << Al));
    } else if (r == 15) {
      k += 15;                                             // ZRL
    } else {
      eobrun = 1 << r;                                     // EOBr: this block and eobrun - 1 more end here
      if (r) eobrun += br.get(r);
      --eobrun;
      break;
    }
  }
  return OVM_OK;
}

inline void prog_refine_nonzero(BitReader& br, int16_t* c, int p1) {      // one correction bit for an already-nonzero coefficient
  if (br.get(1) && (*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c - p1);
}

inline int prog_ac_refine(BitReader& br, const Huff& ha, int16_t* blk, int Ss, int Se, int Al, int& eobrun) {
  const int p1 = 1 << Al;
  int k = Ss;
  if (eobrun == 0) {
    for (; k <= Se; ++k) {
      const int rs = decode_symbol(br, ha);
      if (rs < 0) return OVM_ERR_INVALID;
      int r = rs >> 4, sz = rs & 15, val = 0;
      if (sz) {
        if (sz != 1) return OVM_ERR_INVALID;               // a newly nonzero coefficient is +-1 at this bit position
        val = br.get(1) ? p1 : -p1;
      } else if (r != 15) {
        eobrun = 1 << r;
        if (r) eobrun += br.get(r);
        break;                                             // the rest of the band: correction bits only (below)
      }
      // walk over r still-zero coefficients, refining the nonzero ones passed on the way
      for (; k <= Se; ++k) {
        int16_t* c = blk + kZigzag[k];
        if (*c != 0) prog_refine_nonzero(br, c, p1);
        else if (--r < 0) break;
      }
      if (val) {
        if (k > Se) return OVM_ERR_INVALID;
        blk[kZigzag[k]] = (int16_t)val;
      }
    }
  }
  if (eobrun > 0) {
    for (; k <= Se; ++k) {
      int16_t* c = blk + kZigzag[k];
      if (*c != 0) prog_refine_nonzero(br, c, p1);
    }
    --eobrun;
  }
  return OVM_OK;
}

struct Frame {
  OvmJpegInfo info;
  int cid[3] = {0, 0, 0};
  size_t plane_off[3] = {0, 0, 0};      // in blocks
  int restart = 0;
  Huff dc[4], ac[4];
  bool have_sof = false, jfif = false, adobe = false; int adobe_transform = -1;
  bool progressive = false;
  int n_scans = 0;
  int orientation = 1;                  // Exif Orientation of APP1 (1 = as stored)
  signed char cbits[3][64];             // progressive: bit position each coefficient has been received down to (-1: never)
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// walks the marker segments up to (not into) the first SOS when `scan_cb` is null, else decodes every scan
inline int parse(const uint8_t* d, size_t n, Frame& f, int16_t* coef) {
  if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return OVM_ERR_INVALID;
  memset(&f.info, 0, sizeof(f.info));
  size_t pos = 2;
  int dcpred[3];
  bool saw_sos = false, saw_eoi = false;
  while (pos + 2 <= n) {
    if (d[pos] != 0xFF) return OVM_ERR_INVALID;
    while (pos < n && d[pos] == 0xFF) ++pos;              // fill bytes
    if (pos >= n) return OVM_ERR_INVALID;
    const int m = d[pos++];
    if (m == 0xD9) { saw_eoi = true; break; }             // EOI
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;  // standalone
    if (pos + 2 > n) return OVM_ERR_INVALID;
    const int len = be16(d + pos);
    if (len < 2 || pos + len > n) return OVM_ERR_INVALID;
    const uint8_t* s = d + pos + 2; const int sl = len - 2;
    if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
      if (f.have_sof || sl < 6) return OVM_ERR_INVALID;
      f.progressive = (m == 0xC2);
      memset(f.cbits, -1, sizeof(f.cbits));
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
      // Pillow's MAX_IMAGE_PIXELS (the size above which the reference's reader warns of a decompression bomb): beyond it the file
      // goes to the host reader and its policy instead of pinning hundreds of MB for a 100-byte header
      if ((int64_t)I.width * I.height > 89478485 || off > ((size_t)1 << 22)) return OVM_ERR_UNSUPPORTED;
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
    } else if (m == 0xE1) {
      // APP1 "Exif": the Orientation tag (0x0112) of IFD0. Both readers of the reference apply it (cv2.imread, demo/demo.py:52;
      // detectron2 read_image -> _apply_exif_orientation, dataset_mapper.py:38); files that carry one other than 1 are left to the
      // host reader, which transposes (OVM_ERR_UNSUPPORTED below).
      if (sl >= 14 && !memcmp(s, "Exif\0\0", 6)) {
        const uint8_t* t = s + 6; const int tl = sl - 6;
        const bool le = t[0] == 'I' && t[1] == 'I', be = t[0] == 'M' && t[1] == 'M';
        auto r16 = [&](int o) { return le ? (t[o] | (t[o + 1] << 8)) : ((t[o] << 8) | t[o + 1]); };
        auto r32 = [&](int o) { return le ? (uint32_t)(t[o] | (t[o + 1] << 8) | (t[o + 2] << 16)) | ((uint32_t)t[o + 3] << 24)
                                          : ((uint32_t)t[o] << 24) | (uint32_t)((t[o + 1] << 16) | (t[o + 2] << 8) | t[o + 3]); };
        if ((le || be) && r16(2) == 42) {
          const uint32_t ifd = r32(4);
          if (ifd >= 8 && ifd < (uint32_t)tl && (uint32_t)tl - ifd >= 2) {          // (compared without forming ifd + x: ifd is the file's)
            const int cnt = r16((int)ifd);
            const uint32_t room = (uint32_t)tl - ifd - 2;
            for (int e = 0; e < cnt && 12u * (uint32_t)(e + 1) <= room; ++e) {
              const int o = (int)ifd + 2 + 12 * e;
              if (r16(o) == 0x0112 && r16(o + 2) == 3) { f.orientation = r16(o + 8); break; }
            }
          }
        }
      }
    } else if (m == 0xEE) {
      if (sl >= 12 && !memcmp(s, "Adobe", 5)) { f.adobe = true; f.adobe_transform = s[11]; }
    } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
      return OVM_ERR_UNSUPPORTED;                          // lossless, differential, arithmetic
    } else if (m == 0xCC) {
      return OVM_ERR_UNSUPPORTED;                          // arithmetic conditioning
    } else if (m == 0xDA) {
      if (!f.have_sof) return OVM_ERR_INVALID;
      saw_sos = true;
      if (!coef) break;                                    // header walk ends here
      const OvmJpegInfo& I = f.info;
      // every scan walks all of its blocks even when it carries no data: bound the number of scans (libjpeg-turbo's fuzz limit is of
      // the same order; Pillow's standard progression has 10) so that a few KB of empty SOS headers cannot cost minutes
      if (++f.n_scans > 64 * I.ncomp) return OVM_ERR_INVALID;
      if (sl < 1) return OVM_ERR_INVALID;
      const int ns = s[0];
      if (ns < 1 || ns > I.ncomp || sl < 1 + 2 * ns + 3) return OVM_ERR_INVALID;
      int sc[3], td[3], ta[3];
      for (int i = 0; i < ns; ++i) {
        int c = -1;
        for (int q = 0; q < I.ncomp; ++q) if (f.cid[q] == s[1 + 2 * i]) c = q;
        if (c < 0) return OVM_ERR_INVALID;
        sc[i] = c; td[i] = s[2 + 2 * i] >> 4; ta[i] = s[2 + 2 * i] & 15;
        if (td[i] > 3 || ta[i] > 3) return OVM_ERR_INVALID;
      }
      const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
      if (!f.progressive) {
        if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) return OVM_ERR_INVALID;
      } else {
        if (Ss > Se || Se > 63 || Al > 13 || (Ah != 0 && Ah != Al + 1)) return OVM_ERR_INVALID;
        if (Ss == 0 ? Se != 0 : ns != 1) return OVM_ERR_INVALID;       // DC scans carry only DC; AC scans one component
        for (int i = 0; i < ns; ++i)
          for (int k = Ss; k <= Se; ++k) f.cbits[sc[i]][k] = (signed char)Al;
      }
      for (int i = 0; i < ns; ++i) {
        const bool need_dc = !f.progressive || (Ss == 0 && Ah == 0), need_ac = !f.progressive || Ss > 0;
        if ((need_dc && !f.dc[td[i]].present) || (need_ac && !f.ac[ta[i]].present)) return OVM_ERR_INVALID;
      }
      int eobrun = 0;
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
            eobrun = 0;
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
                if (f.progressive) {
                  int rc = OVM_OK;
                  if (Ss == 0) {
                    if (Ah == 0) rc = prog_dc_first(br, hd, blk, dcpred[c], Al);
                    else if (br.get(1)) blk[0] = (int16_t)(blk[0] | (1 << Al));
                  } else {
                    rc = Ah == 0 ? prog_ac_first(br, ha, blk, Ss, Se, Al, eobrun) : prog_ac_refine(br, ha, blk, Ss, Se, Al, eobrun);
                  }
                  if (rc) return rc;
                  if (br.exhausted()) return OVM_ERR_INVALID;
                  if (dcpred[c] < -65536 || dcpred[c] > 65535) return OVM_ERR_INVALID;
                  continue;
                }
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
                if (br.exhausted()) return OVM_ERR_INVALID;               // the data ended inside this block: a truncated file
                // the DC predictor of a valid 8-bit stream stays within 11 + 1 bits; an attacker's stream must not walk it to overflow
                if (dcpred[c] < -65536 || dcpred[c] > 65535) return OVM_ERR_INVALID;
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
  // a file that ends in its headers, or behind its scans without the EOI marker, is a cut file: Pillow (the reference's reader) raises
  // "image file is truncated" on both
  if (!saw_sos || (coef && !saw_eoi)) return OVM_ERR_INVALID;
  if (f.orientation >= 2 && f.orientation <= 8) return OVM_ERR_UNSUPPORTED;      // to be transposed: the host reader's job
  OvmJpegInfo& I = f.info;
  if (f.progressive && coef)                               // every coefficient of every component down to bit 0? (an incomplete
    for (int c = 0; c < I.ncomp; ++c)                      // progression is what libjpeg smooths across blocks: not reproduced here)
      for (int k = 0; k < 64; ++k)
        if (f.cbits[c][k] != 0) return OVM_ERR_UNSUPPORTED;
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

inline int host_info(const uint8_t* data, size_t n, OvmJpegInfo* info) {
  if (!data || !info) return OVM_ERR_INVALID;
  Frame f;
  const int r = parse(data, n, f, nullptr);
  if (r) return r;
  *info = f.info;
  return OVM_OK;
}

inline int host_entropy_decode(const uint8_t* data, size_t n, int16_t* coef, int64_t coef_capacity, OvmJpegInfo* info) {
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

}  // namespace ovm_jpeg
