// Fused device kernels of the native GroundingDINO engine (see gdino.hpp). gfx950, wave = 64.
#include <hip/hip_runtime.h>
#include <cmath>
#include "gdino.hpp"

namespace ovm {

namespace {

inline dim3 g1(long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

// ------------------------------------------------------------------------------------------------------------------------------
// Row operator. One wave per output row; the row (after gather and residual) is held in registers when a LayerNorm needs two
// passes over it, streamed otherwise.
// ------------------------------------------------------------------------------------------------------------------------------
template <int VEC> struct VecT;
template <> struct VecT<4> { typedef f32x4 T; };
template <> struct VecT<1> { typedef float T; };

template <int VEC>
__device__ __forceinline__ void rowop_load(const RowOpParams& p, int row, int e, float* v) {
  // elements e .. e+VEC-1 of the gathered (+ residual) row; e % VEC == 0 and a VEC group never straddles a segment
  int src = row, c = e;
  if (p.idx) { const int j = e / p.seg; c = e - j * p.seg; src = p.idx[(size_t)row * p.nidx + j]; }
  if (src < 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = 0.f;
  } else if (VEC == 4) {
    const f32x4 t = *(const f32x4*)(p.x + (size_t)src * p.ldx + c);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else {
    v[0] = p.x[(size_t)src * p.ldx + c];
  }
  if (p.res) {
    if (VEC == 4) { const f32x4 t = *(const f32x4*)(p.res + (size_t)row * p.ldr + e); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
    else v[0] += p.res[(size_t)row * p.ldr + e];
  }
}

template <int VEC>
__device__ __forceinline__ void rowop_store(const RowOpParams& p, int row, int e, const float* y) {
  if (p.y) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) p.y[(size_t)row * p.ldy + e + i] = y[i];
  }
  if (p.hi) {
    const int c0 = p.il ? il_col(e) : e;                  // VEC consecutive columns stay inside one 32-column group (e % VEC == 0, VEC <= 4)
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      half_t h, l; split_f16_nt(y[i], h, l);
      p.hi[(size_t)row * p.ldh + c0 + i] = h;
      if (p.lo) p.lo[(size_t)row * p.ldh + c0 + i] = l;
    }
  }
  if (p.add) {
    float y2[VEC];
    const float* ar = p.add + (size_t)(row % p.add_rows) * p.ld_add + e;
#pragma unroll
    for (int i = 0; i < VEC; ++i) y2[i] = y[i] + ar[i];
    if (p.y2) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) p.y2[(size_t)row * p.ldy2 + e + i] = y2[i];
    }
    if (p.hi2) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        half_t h, l; split_f16_nt(y2[i], h, l);
        p.hi2[(size_t)row * p.ldh2 + e + i] = h;
        if (p.lo2) p.lo2[(size_t)row * p.ldh2 + e + i] = l;
      }
    }
  }
}

template <int VEC, int MAXV>
__global__ __launch_bounds__(256) void rowop_kernel(const RowOpParams p) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= p.M) return;
  const int D = p.D;
  if (p.gamma) {
    float v[MAXV][VEC];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int e = (lane + 64 * i) * VEC;
      if (e < D) {
        rowop_load<VEC>(p, row, e, v[i]);
#pragma unroll
        for (int r = 0; r < VEC; ++r) sum += v[i][r];
      }
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int e = (lane + 64 * i) * VEC;
      if (e < D) {
#pragma unroll
        for (int r = 0; r < VEC; ++r) { const float d = v[i][r] - mean; sq += d * d; }
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + p.eps);
    const bool masked = p.zero_masked && p.idx && p.idx[(size_t)row * p.nidx] < 0;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int e = (lane + 64 * i) * VEC;
      if (e < D) {
        float y[VEC];
#pragma unroll
        for (int r = 0; r < VEC; ++r) y[r] = masked ? 0.f : (v[i][r] - mean) * rstd * p.gamma[e + r] + p.beta[e + r];
        rowop_store<VEC>(p, row, e, y);
      }
    }
  } else {
    for (int e = lane * VEC; e < D; e += 64 * VEC) {
      float v[VEC];
      rowop_load<VEC>(p, row, e, v);
      rowop_store<VEC>(p, row, e, v);
    }
  }
  // zero fill of the K padding of the split images
  if (p.hi && !p.il && p.ldh > D)
    for (int e = D + lane; e < p.ldh; e += 64) { p.hi[(size_t)row * p.ldh + e] = (half_t)0.f; if (p.lo) p.lo[(size_t)row * p.ldh + e] = (half_t)0.f; }
  if (p.hi2 && p.ldh2 > D)
    for (int e = D + lane; e < p.ldh2; e += 64) { p.hi2[(size_t)row * p.ldh2 + e] = (half_t)0.f; if (p.lo2) p.lo2[(size_t)row * p.ldh2 + e] = (half_t)0.f; }
}

// ------------------------------------------------------------------------------------------------------------------------------
// fp32 attention on the matrix cores. S^T = K Q^T per 16-key tile (A = K rows from LDS, B = the wave's 16 queries in
// registers), so a lane's accumulator registers are 4 keys of ONE query (column = lane & 15): the softmax reduces over
// registers and the 4 lane groups only, and the probabilities are directly the B operand of O^T = V^T P^T - no data movement
// between the two products. v_mfma_f32_16x16x4_f32 is an exact fp32 FMA chain (MI355X_MICROARCH.md, Matrix cores).
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int kKC = 144;                       // keys per LDS chunk (= one Swin window)

template <int DH>
__global__ __launch_bounds__(576) void attn_f32_kernel(const AttnF32Params p) {
  // LDS: K rows [kKC][DH + 4] and V TRANSPOSED [DH][kKC + 4], so that both MFMA A operands are 16-byte fragment reads: a lane's
  // slice of the head dimension is contiguous (k = g * NS + s, the same regrouping on the Q side), and the 4 keys a lane feeds to
  // the 4 PV MFMAs of a key tile are contiguous in the transposed V. Row strides 36 / 68 / 20 and 148 floats keep the 16 lanes
  // of a read phase on distinct banks.
  constexpr int LDK = DH + 4, LDT = kKC + 4, NS = DH / 4, ND = DH / 16, NKT = kKC / 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Ks = (float*)smem_raw;                // [kKC][LDK]
  float* Vt = Ks + kKC * LDK;                  // [DH][LDT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int li = lane & 15, g = lane >> 4;
  const int z = blockIdx.y, b1 = z / p.nb2, b2 = z - b1 * p.nb2;
  const float* qb = p.q + (size_t)b1 * p.sq1 + (size_t)b2 * p.sq2;
  const float* kb = p.k + (size_t)b1 * p.sk1 + (size_t)b2 * p.sk2;
  const float* vb = p.v + (size_t)b1 * p.sv1 + (size_t)b2 * p.sv2;
  const int q0 = (blockIdx.x * (nthr >> 6) + wave) * 16;
  const bool wave_active = q0 < p.Tq;          // idle waves still help staging and must reach the barriers
  const int qi = (q0 + li < p.Tq) ? q0 + li : p.Tq - 1;
  const float kLog2e = 1.44269504088896340736f;
  float qf[NS];
  {
    const float sc = p.scale * kLog2e;
    const float* qr = qb + (size_t)qi * p.ldq + g * NS;
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = qr[s] * sc;
  }
  const float* bh = p.bias_h ? p.bias_h + (size_t)b2 * p.sbh + (size_t)qi * p.ldbh : nullptr;
  const float* bb = p.bias_b ? p.bias_b + (size_t)b1 * p.sbb + (size_t)qi * p.ldbb : nullptr;
  const size_t relrow = (((size_t)b1 * p.Tq + qi) * p.nb2 + b2) * (size_t)p.ldrel;
  const float* rh = p.rel_h ? p.rel_h + relrow : nullptr;
  const float* rw = p.rel_h ? p.rel_w + relrow : nullptr;
  f32x4 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;

  for (int kc = 0; kc < p.Tk; kc += kKC) {
    __syncthreads();
    // stage K / V rows kc .. kc+kKC (zero beyond Tk): DH/4 float4 pieces per row; V goes in transposed
    for (int t = tid; t < kKC * (DH / 4); t += nthr) {
      const int r = t / (DH / 4), c = (t - r * (DH / 4)) * 4;
      f32x4 kv = (f32x4){0.f, 0.f, 0.f, 0.f}, vv = kv;
      if (kc + r < p.Tk) {
        kv = *(const f32x4*)(kb + (size_t)(kc + r) * p.ldk + c);
        vv = *(const f32x4*)(vb + (size_t)(kc + r) * p.ldv + c);
      }
      *(f32x4*)(Ks + r * LDK + c) = kv;
      float* vd = Vt + c * LDT + r;
      vd[0] = vv[0]; vd[LDT] = vv[1]; vd[2 * LDT] = vv[2]; vd[3 * LDT] = vv[3];
    }
    __syncthreads();
    if (!wave_active) continue;
    const int nk = (p.Tk - kc < kKC) ? p.Tk - kc : kKC;
    const int nkt = (nk + 15) >> 4;
    f32x4 st[NKT];
    float cmax = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {                          // wave-uniform
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* kr = Ks + (kt * 16 + li) * LDK + g * NS;
        f32x4 kq[NS / 4];
#pragma unroll
        for (int j = 0; j < NS / 4; ++j) kq[j] = *(const f32x4*)(kr + 4 * j);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kq[s >> 2][s & 3], qf[s], acc, 0, 0, 0);
        const int key0 = kc + kt * 16 + g * 4;
        if (p.bias_vec) {                      // wave-uniform: Tk % 4 == 0 and 16-byte aligned bias rows -> 4 keys per load
          if (key0 < p.Tk) {
            if (bh) { const f32x4 b = *(const f32x4*)(bh + key0); acc[0] += b[0] * kLog2e; acc[1] += b[1] * kLog2e; acc[2] += b[2] * kLog2e; acc[3] += b[3] * kLog2e; }
            if (bb) { const f32x4 b = *(const f32x4*)(bb + key0); acc[0] += b[0] * kLog2e; acc[1] += b[1] * kLog2e; acc[2] += b[2] * kLog2e; acc[3] += b[3] * kLog2e; }
          } else {
            acc = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) cmax = fmaxf(cmax, acc[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = key0 + r;
            float x = acc[r];
            if (key < p.Tk) {
              if (bh) x += bh[key] * kLog2e;
              if (bb) x += bb[key] * kLog2e;
              if (rh) { const int kh = key / p.rel_gw; x += (rh[kh] + rw[key - kh * p.rel_gw]) * kLog2e; }
            } else {
              x = -INFINITY;
            }
            acc[r] = x;
            cmax = fmaxf(cmax, x);
          }
        }
        st[kt] = acc;
      } else {
        st[kt] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      }
    }
    cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
    cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
    const float mnew = fmaxf(m, cmax);
    const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
    const float alpha = exp2f(m - msafe);      // m = -inf -> 0
    l *= alpha;
#pragma unroll
    for (int d = 0; d < ND; ++d) { o[d][0] *= alpha; o[d][1] *= alpha; o[d][2] *= alpha; o[d][3] *= alpha; }
    m = mnew;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
        f32x4 pr;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pr[r] = exp2f(st[kt][r] - msafe); l += pr[r]; }
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          const f32x4 vv = *(const f32x4*)(Vt + (d * 16 + li) * LDT + kt * 16 + g * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[r], pr[r], o[d], 0, 0, 0);
        }
      }
    }
  }
  if (!wave_active) return;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = (l > 0.f) ? 1.0f / l : 0.f;
  if (q0 + li < p.Tq) {
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const f32x4 y = (f32x4){o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv};
      const int col = d * 16 + g * 4;
      if (p.o) *(f32x4*)(p.o + (size_t)b1 * p.so1 + (size_t)b2 * p.so2 + (size_t)(q0 + li) * p.ldo + col) = y;
      if (p.ohi) {
        half4 h, lo4;
#pragma unroll
        for (int r = 0; r < 4; ++r) { half_t hh, ll; split_f16_nt(y[r], hh, ll); h[r] = hh; lo4[r] = ll; }
        const size_t oo = (size_t)b1 * p.soh1 + (size_t)b2 * p.soh2 + (size_t)(q0 + li) * p.ldoh + col;
        *(half4*)(p.ohi + oo) = h;
        if (p.olo) *(half4*)(p.olo + oo) = lo4;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Swin window attention with the qkv projection inside. Phase 1: the 9 waves form a 3 x 3 grid over (48 window rows) x (q | k | v of
// the head): per k-step a lane loads 6 activation fragments (3 row tiles x hi / lo, 16 B each, straight from the split rows in HBM)
// and 4 weight fragments (2 column tiles x hi / lo, 1 KiB contiguous per instruction in the fragment-ordered image), the next step's
// loads are issued before this step's 18 MFMAs. Same products in the same order as the 128-tile GEMM (lo . hi, hi . lo, hi . hi per
// k-step of 32). The accumulators go to LDS as Q rows [144][36], K rows [144][36] and V transposed [32][148] - what attn_f32_kernel
// stages from HBM. Phase 2: that kernel's loop for one chunk of 144 keys (wave = 16 queries).
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(576) void swin_qkv_attn_kernel(const SwinQkvAttnParams p) {
  constexpr int DH = 32, WS2 = 144, LDK = DH + 4, LDT = WS2 + 4, NS = DH / 4, ND = DH / 16, NKT = WS2 / 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Qs = (float*)smem_raw;                // [144][LDK]
  float* Ks = Qs + WS2 * LDK;                  // [144][LDK]
  float* Vt = Ks + WS2 * LDK;                  // [DH][LDT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int w = (int)blockIdx.x / p.nh, h = (int)blockIdx.x - w * p.nh;
  // ---------------- phase 1: [q | k | v] = x W^T + b ----------------
  {
    const int mg = wave / 3, ng = wave - mg * 3;              // rows 48 mg .. 48 mg + 47; part ng: 0 = q, 1 = k, 2 = v
    const int tile0 = (ng * p.C + h * DH) >> 4;               // first of the two 16-row weight tiles of this part and head
    const half_t* xh = p.xhi + (size_t)(w * WS2 + mg * 48 + fr) * p.ldx + fq * 8;
    const half_t* xl = p.xlo + (size_t)(w * WS2 + mg * 48 + fr) * p.ldx + fq * 8;
    const half_t* wp = p.wfrag + (size_t)tile0 * p.KS * 1024 + lane * 8;       // tile stride = KS * (hi 512 + lo 512) halves
    const size_t wt = (size_t)p.KS * 1024;
    f32x4 acc[2][3];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 3; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // The weights come from HBM (cold: the detector streams ~0.9 GB of them per forward beside the ViT's 1.2 GB), the activation rows
    // from L2: weight fragments are kept THREE k-steps ahead (4 register slots), activation fragments one step ahead (2 slots).
    half8 xa[2][6], wa[4][4];
    auto loadx = [&](int ks, half8* xd) {
      const int k = (ks < p.KS) ? ks : p.KS - 1;              // past the end: re-load the last step (never used)
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        xd[2 * m] = *(const half8*)(xh + (size_t)m * 16 * p.ldx + k * 32);
        xd[2 * m + 1] = *(const half8*)(xl + (size_t)m * 16 * p.ldx + k * 32);
      }
    };
    auto loadw = [&](int ks, half8* wd) {
      const int k = (ks < p.KS) ? ks : p.KS - 1;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        wd[2 * n] = *(const half8*)(wp + n * wt + (size_t)k * 1024);
        wd[2 * n + 1] = *(const half8*)(wp + n * wt + (size_t)k * 1024 + 512);
      }
    };
    auto mfmak = [&](const half8* xd, const half8* wd) {
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wd[2 * n + 1], xd[2 * m], acc[n][m], 0, 0, 0);
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wd[2 * n], xd[2 * m + 1], acc[n][m], 0, 0, 0);
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wd[2 * n], xd[2 * m], acc[n][m], 0, 0, 0);
        }
    };
    loadw(0, wa[0]); loadw(1, wa[1]); loadw(2, wa[2]); loadx(0, xa[0]);
    for (int ks = 0; ks < p.KS; ks += 4) {                    // KS % 4 == 0 (C % 128 == 0, launcher)
      // sched_barrier(0): left alone the scheduler sinks the loads next to their uses (80 VGPRs, nothing in flight under the MFMAs)
#define OVM_SWIN_STEP(KW, WS, KX, XS, XC, WC)                                                   \
      loadw(KW, wa[WS]); loadx(KX, xa[XS]); __builtin_amdgcn_sched_barrier(0);                  \
      mfmak(xa[XC], wa[WC]); __builtin_amdgcn_sched_barrier(0);
      OVM_SWIN_STEP(ks + 3, 3, ks + 1, 1, 0, 0)
      OVM_SWIN_STEP(ks + 4, 0, ks + 2, 0, 1, 1)
      OVM_SWIN_STEP(ks + 5, 1, ks + 3, 1, 0, 2)
      OVM_SWIN_STEP(ks + 6, 2, ks + 4, 0, 1, 3)
#undef OVM_SWIN_STEP
    }
    // lane (fr, fq) holds row m = fr of a row tile, columns 4 fq .. 4 fq + 3 of a column tile (dec_chain.hip's convention)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int d0 = n * 16 + fq * 4;
      const f32x4 b = *(const f32x4*)(p.bias + ng * p.C + h * DH + d0);
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int row = mg * 48 + m * 16 + fr;
        const f32x4 y = (f32x4){acc[n][m][0] + b[0], acc[n][m][1] + b[1], acc[n][m][2] + b[2], acc[n][m][3] + b[3]};
        if (ng == 0) *(f32x4*)(Qs + row * LDK + d0) = y;
        else if (ng == 1) *(f32x4*)(Ks + row * LDK + d0) = y;
        else { float* vd = Vt + d0 * LDT + row; vd[0] = y[0]; vd[LDT] = y[1]; vd[2 * LDT] = y[2]; vd[3 * LDT] = y[3]; }
      }
    }
  }
  __syncthreads();
  // ---------------- phase 2: attn_f32_kernel<32>'s loop over one chunk of 144 keys ----------------
  const int li = lane & 15, g = lane >> 4;
  const int q0 = wave * 16, qi = q0 + li;
  const float kLog2e = 1.44269504088896340736f;
  float qf[NS];
  {
    const float sc = p.scale * kLog2e;
    const float* qr = Qs + qi * LDK + g * NS;
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = qr[s] * sc;
  }
  const float* bh = p.relbias + ((size_t)h * WS2 + qi) * WS2;
  const float* bb = p.mask ? p.mask + ((size_t)w * WS2 + qi) * WS2 : nullptr;
  f32x4 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  f32x4 st[NKT];
  float cmax = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* kr = Ks + (kt * 16 + li) * LDK + g * NS;
    f32x4 kq[NS / 4];
#pragma unroll
    for (int j = 0; j < NS / 4; ++j) kq[j] = *(const f32x4*)(kr + 4 * j);
#pragma unroll
    for (int s = 0; s < NS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kq[s >> 2][s & 3], qf[s], acc, 0, 0, 0);
    const int key0 = kt * 16 + g * 4;
    { const f32x4 b = *(const f32x4*)(bh + key0); acc[0] += b[0] * kLog2e; acc[1] += b[1] * kLog2e; acc[2] += b[2] * kLog2e; acc[3] += b[3] * kLog2e; }
    if (bb) { const f32x4 b = *(const f32x4*)(bb + key0); acc[0] += b[0] * kLog2e; acc[1] += b[1] * kLog2e; acc[2] += b[2] * kLog2e; acc[3] += b[3] * kLog2e; }
#pragma unroll
    for (int r = 0; r < 4; ++r) cmax = fmaxf(cmax, acc[r]);
    st[kt] = acc;
  }
  cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
  cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
  const float mnew = fmaxf(m, cmax);
  const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
  const float alpha = exp2f(m - msafe);
  l *= alpha;
#pragma unroll
  for (int d = 0; d < ND; ++d) { o[d][0] *= alpha; o[d][1] *= alpha; o[d][2] *= alpha; o[d][3] *= alpha; }
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    f32x4 pr;
#pragma unroll
    for (int r = 0; r < 4; ++r) { pr[r] = exp2f(st[kt][r] - msafe); l += pr[r]; }
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const f32x4 vv = *(const f32x4*)(Vt + (d * 16 + li) * LDT + kt * 16 + g * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[r], pr[r], o[d], 0, 0, 0);
    }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = (l > 0.f) ? 1.0f / l : 0.f;
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    const f32x4 y = (f32x4){o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv};
    const int col = d * 16 + g * 4;
    half4 hh, lo4;
#pragma unroll
    for (int r = 0; r < 4; ++r) { half_t a, b; split_f16_nt(y[r], a, b); hh[r] = a; lo4[r] = b; }
    const size_t oo = (size_t)(w * WS2 + qi) * p.ldo + h * DH + col;
    *(half4*)(p.ohi + oo) = hh;
    *(half4*)(p.olo + oo) = lo4;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Fusion-layer bi-attention.
// ------------------------------------------------------------------------------------------------------------------------------
// image side: one wave per (image token, head); softmax over the text tokens; also leaves the raw scores for the text side.
// The text tokens are taken 16 at a time: 16 independent dot products (their wave reductions overlap), one softmax step over the
// block, 16 independent value rows. The first version walked the tokens one by one with the online-softmax update in between - a
// serial chain of a wave reduction, two exponentials and a rescale per token: 56.8 us per launch at 6,015 tokens x 4 heads x 16 text
// tokens, for 50 MB of traffic.
template <int MAXE>                              // dh <= 256 * MAXE
__global__ __launch_bounds__(256) void biattn_img_kernel(const BiAttnParams p) {
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wid >= (long)p.S * p.H) return;
  const int s = (int)(wid / p.H), h = (int)(wid - (long)s * p.H);
  const int dh = p.dh;
  constexpr int TB = 16;
  f32x4 qv[MAXE], acc[MAXE];
#pragma unroll
  for (int i = 0; i < MAXE; ++i) {
    const int e = (lane + 64 * i) * 4;
    qv[i] = (e < dh) ? *(const f32x4*)(p.qv + (size_t)s * p.ldq + h * dh + e) : (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float m = -INFINITY, l = 0.f;
  for (int t0 = 0; t0 < p.T; t0 += TB) {
    float d[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int t = (t0 + j < p.T) ? t0 + j : p.T - 1;          // clamped: the value is discarded below
      float x = 0.f;
#pragma unroll
      for (int i = 0; i < MAXE; ++i) {
        const int e = (lane + 64 * i) * 4;
        if (e < dh) {
          const f32x4 kv = *(const f32x4*)(p.kt + (size_t)t * p.ldk + h * dh + e);
          x += qv[i][0] * kv[0] + qv[i][1] * kv[1] + qv[i][2] * kv[2] + qv[i][3] * kv[3];
        }
      }
      d[j] = x;
    }
    float mine = 0.f, mb = -INFINITY;
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      d[j] = (t0 + j < p.T) ? wave_sum(d[j]) * p.scale : -INFINITY;
      if (lane == j) mine = d[j];
      mb = fmaxf(mb, d[j]);
    }
    if (lane < TB && t0 + lane < p.T) p.sc[((size_t)h * p.T + t0 + lane) * p.S + s] = mine;
    const float mn = fmaxf(m, mb);
    const float alpha = expf(m - mn);
    l *= alpha;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) { acc[i][0] *= alpha; acc[i][1] *= alpha; acc[i][2] *= alpha; acc[i][3] *= alpha; }
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const float pr = expf(d[j] - mn);                         // -inf beyond T: 0
      l += pr;
      const int t = (t0 + j < p.T) ? t0 + j : p.T - 1;
#pragma unroll
      for (int i = 0; i < MAXE; ++i) {
        const int e = (lane + 64 * i) * 4;
        if (e < dh) {
          const f32x4 tv = *(const f32x4*)(p.vt + (size_t)t * p.ldvt + h * dh + e);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][r] += pr * tv[r];
        }
      }
    }
    m = mn;
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int i = 0; i < MAXE; ++i) {
    const int e = (lane + 64 * i) * 4;
    if (e < dh) {
      const f32x4 y = (f32x4){acc[i][0] * inv, acc[i][1] * inv, acc[i][2] * inv, acc[i][3] * inv};
      if (p.cv) *(f32x4*)(p.cv + (size_t)s * p.ldcv + h * dh + e) = y;
      if (p.cv_hi) {
        half4 hh, ll;
#pragma unroll
        for (int r = 0; r < 4; ++r) { half_t a, b; split_f16_nt(y[r], a, b); hh[r] = a; ll[r] = b; }
        *(half4*)(p.cv_hi + (size_t)s * p.ldcv + h * dh + e) = hh;
        if (p.cv_lo) *(half4*)(p.cv_lo + (size_t)s * p.ldcv + h * dh + e) = ll;
      }
    }
  }
}

// text side, softmax statistics over the image tokens: one workgroup per (head, text token)
__global__ __launch_bounds__(256) void biattn_stats_kernel(const BiAttnParams p) {
  __shared__ float red[8];
  const int ht = blockIdx.x;
  const float* row = p.sc + (size_t)ht * p.S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float m = -INFINITY;
  for (int i = tid; i < p.S; i += 256) m = fmaxf(m, row[i]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int i = tid; i < p.S; i += 256) s += expf(row[i] - m);
  s = wave_sum(s);
  if (lane == 0) red[4 + wave] = s;
  __syncthreads();
  if (tid == 0) { p.stat[2 * ht] = m; p.stat[2 * ht + 1] = (red[4] + red[5]) + (red[6] + red[7]); }
}

// text side, partial context of one chunk of image tokens: part[c][t][h*dh + d] = sum_{s in chunk} exp(sc - max) vv[s][h*dh + d]
__global__ __launch_bounds__(256) void biattn_txt_partial_kernel(const BiAttnParams p) {
  constexpr int TB = 16, CH = 128;
  __shared__ float pr[TB][CH];
  const int c = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int s0 = c * p.chunk;
  const int ns = (p.S - s0 < p.chunk) ? p.S - s0 : p.chunk;   // chunk <= CH
  const int E = p.H * p.dh;
  for (int t0 = 0; t0 < p.T; t0 += TB) {
    const int nt = (p.T - t0 < TB) ? p.T - t0 : TB;
    __syncthreads();
    for (int i = tid; i < TB * CH; i += 256) {
      const int t = i / CH, si = i - t * CH;
      float v = 0.f;
      if (t < nt && si < ns) v = expf(p.sc[((size_t)h * p.T + t0 + t) * p.S + s0 + si] - p.stat[2 * (h * p.T + t0 + t)]);
      pr[t][si] = v;
    }
    __syncthreads();
    for (int d = tid; d < p.dh; d += 256) {
      float acc[TB];
#pragma unroll
      for (int t = 0; t < TB; ++t) acc[t] = 0.f;
      const float* vcol = p.vv + (size_t)s0 * p.ldvv + h * p.dh + d;
      // eight value rows in flight per thread (the one-row-at-a-time loop exposed a memory latency per image token: 41 us per launch
      // for 25 MB); rows past the chunk's end re-read its last row against pr = 0, the sums keep their order
      for (int si = 0; si < ns; si += 8) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int r = (si + j < ns) ? si + j : ns - 1; x[j] = vcol[(size_t)r * p.ldvv]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (si + j < CH) {
#pragma unroll
            for (int t = 0; t < TB; ++t) acc[t] = fmaf(pr[t][si + j], x[j], acc[t]);
          }
        }
      }
      for (int t = 0; t < nt; ++t) p.part[((size_t)c * p.T + t0 + t) * E + h * p.dh + d] = acc[t];
    }
  }
}

__global__ void biattn_txt_combine_kernel(const BiAttnParams p) {
  const int E = p.H * p.dh;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)p.T * E) return;
  const int t = (int)(i / E), e = (int)(i - (long)t * E), h = e / p.dh;
  float acc = 0.f;
  for (int c = 0; c < p.nchunk; ++c) acc += p.part[((size_t)c * p.T + t) * E + e];
  p.ct[(size_t)t * E + e] = acc / p.stat[2 * (h * p.T + t) + 1];
}

// ------------------------------------------------------------------------------------------------------------------------------
// Multi-scale deformable attention (Deformable-DETR sampling = F.grid_sample bilinear, zeros padding, align_corners=False), one
// thread per (query, head, channel); the softmax over the L*P logits and the sampling locations are computed inline.
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void msdeform_fused_kernel(const MsDeformParams p) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)p.Q * p.H * p.dh;
  if (t >= total) return;
  const int d = (int)(t % p.dh); long r = t / p.dh;
  const int hh = (int)(r % p.H); const int q = (int)(r / p.H);
  const int LP = p.L * p.P;
  const float* offp = p.ow + (size_t)q * p.ldow + (size_t)hh * LP * 2;
  const float* lgp = p.ow + (size_t)q * p.ldow + (size_t)p.H * LP * 2 + (size_t)hh * LP;
  float mx = -INFINITY;
  for (int i = 0; i < LP; ++i) mx = fmaxf(mx, lgp[i]);
  float den = 0.f;
  for (int i = 0; i < LP; ++i) den += expf(lgp[i] - mx);
  const float inv = 1.0f / den;
  const float* rf = p.ref + (size_t)q * p.ldref;
  const int HD = p.H * p.dh;
  float acc = 0.f;
  for (int l = 0; l < p.L; ++l) {
    const int Hh = p.lh[l], Ww = p.lw[l];
    const float* vb = p.value + (size_t)p.lstart[l] * p.ldv + (size_t)hh * p.dh + d;
    float mulx, muly;
    if (p.mode == 0) { mulx = 1.0f / (float)Ww; muly = 1.0f / (float)Hh; }
    else { mulx = rf[2] * (0.5f / (float)p.P); muly = rf[3] * (0.5f / (float)p.P); }
    for (int pt = 0; pt < p.P; ++pt) {
      const float lx = __fadd_rn(__fmul_rn(offp[(l * p.P + pt) * 2], mulx), rf[0]);
      const float ly = __fadd_rn(__fmul_rn(offp[(l * p.P + pt) * 2 + 1], muly), rf[1]);
      const float gx = 2.f * lx - 1.f, gy = 2.f * ly - 1.f;
      const float ix = ((gx + 1.f) * (float)Ww - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)Hh - 1.f) * 0.5f;
      const float fx = floorf(ix), fy = floorf(iy);
      const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
      const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
      float v = 0.f;
      if (y0 >= 0 && y0 < Hh) {
        if (x0 >= 0 && x0 < Ww) v += wy0 * wx0 * vb[(size_t)(y0 * Ww + x0) * p.ldv];
        if (x1 >= 0 && x1 < Ww) v += wy0 * wx1 * vb[(size_t)(y0 * Ww + x1) * p.ldv];
      }
      if (y1 >= 0 && y1 < Hh) {
        if (x0 >= 0 && x0 < Ww) v += wy1 * wx0 * vb[(size_t)(y1 * Ww + x0) * p.ldv];
        if (x1 >= 0 && x1 < Ww) v += wy1 * wx1 * vb[(size_t)(y1 * Ww + x1) * p.ldv];
      }
      acc += v * (expf(lgp[l * p.P + pt] - mx) * inv);
    }
  }
  (void)HD;
  const int col = hh * p.dh + d;
  if (p.out) p.out[(size_t)q * p.ldo + col] = acc;
  if (p.ohi) {
    half_t h, lo; split_f16(acc, h, lo);
    p.ohi[(size_t)q * p.ldoh + col] = h;
    if (p.olo) p.olo[(size_t)q * p.ldoh + col] = lo;
  }
}

// The common geometry (L * P = 16 samples per head, dh % 4 == 0): one thread per (query, head, FOUR channels). The softmax over the
// 16 logits is evaluated once per thread into registers (the per-channel kernel above spends 48 expf per output element), the four
// taps of a sample are float4 loads (8 lanes cover a head's 128-byte row), and every output element sees exactly the operations of
// the per-channel kernel in the same order - the two kernels agree bit for bit.
template <int LP>
__global__ __launch_bounds__(256) void msdeform_fused4_kernel(const MsDeformParams p) {
  const int dq = p.dh >> 2;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)p.Q * p.H * dq;
  if (t >= total) return;
  const int d = (int)(t % dq) * 4; long r = t / dq;
  const int hh = (int)(r % p.H); const int q = (int)(r / p.H);
  const float* offp = p.ow + (size_t)q * p.ldow + (size_t)hh * LP * 2;
  const float* lgp = p.ow + (size_t)q * p.ldow + (size_t)p.H * LP * 2 + (size_t)hh * LP;
  float lg[LP], off[2 * LP];
#pragma unroll
  for (int i = 0; i < LP; i += 4) *(f32x4*)(lg + i) = *(const f32x4*)(lgp + i);
#pragma unroll
  for (int i = 0; i < 2 * LP; i += 4) *(f32x4*)(off + i) = *(const f32x4*)(offp + i);
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < LP; ++i) mx = fmaxf(mx, lg[i]);
  float den = 0.f;
#pragma unroll
  for (int i = 0; i < LP; ++i) { lg[i] = expf(lg[i] - mx); den += lg[i]; }
  const float inv = 1.0f / den;
  const float* rf = p.ref + (size_t)q * p.ldref;
  const float r0 = rf[0], r1 = rf[1];
  float mulx_b = 0.f, muly_b = 0.f;
  if (p.mode != 0) { mulx_b = rf[2] * (0.5f / (float)p.P); muly_b = rf[3] * (0.5f / (float)p.P); }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int i = 0;
  for (int l = 0; l < p.L; ++l) {
    const int Hh = p.lh[l], Ww = p.lw[l];
    const float* vb = p.value + (size_t)p.lstart[l] * p.ldv + (size_t)hh * p.dh + d;
    float mulx, muly;
    if (p.mode == 0) { mulx = 1.0f / (float)Ww; muly = 1.0f / (float)Hh; }
    else { mulx = mulx_b; muly = muly_b; }
    for (int pt = 0; pt < p.P; ++pt, ++i) {
      const float lx = __fadd_rn(__fmul_rn(off[2 * i], mulx), r0);
      const float ly = __fadd_rn(__fmul_rn(off[2 * i + 1], muly), r1);
      const float gx = 2.f * lx - 1.f, gy = 2.f * ly - 1.f;
      const float ix = ((gx + 1.f) * (float)Ww - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)Hh - 1.f) * 0.5f;
      const float fx = floorf(ix), fy = floorf(iy);
      const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
      const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
      // branch-free taps: out-of-range corners read a clamped (valid) address with weight 0 - v + 0 * a leaves v exactly as the
      // skipped tap did - so the four loads of a sample (and those of the next samples) are issued together instead of one per branch
      const bool vx0 = x0 >= 0 && x0 < Ww, vx1 = x1 >= 0 && x1 < Ww, vy0 = y0 >= 0 && y0 < Hh, vy1 = y1 >= 0 && y1 < Hh;
      const int cx0 = min(max(x0, 0), Ww - 1), cx1 = min(max(x1, 0), Ww - 1), cy0 = min(max(y0, 0), Hh - 1), cy1 = min(max(y1, 0), Hh - 1);
      const f32x4 a00 = *(const f32x4*)(vb + (size_t)(cy0 * Ww + cx0) * p.ldv), a01 = *(const f32x4*)(vb + (size_t)(cy0 * Ww + cx1) * p.ldv);
      const f32x4 a10 = *(const f32x4*)(vb + (size_t)(cy1 * Ww + cx0) * p.ldv), a11 = *(const f32x4*)(vb + (size_t)(cy1 * Ww + cx1) * p.ldv);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      { const float w_ = (vy0 && vx0) ? wy0 * wx0 : 0.f; v += w_ * a00; }
      { const float w_ = (vy0 && vx1) ? wy0 * wx1 : 0.f; v += w_ * a01; }
      { const float w_ = (vy1 && vx0) ? wy1 * wx0 : 0.f; v += w_ * a10; }
      { const float w_ = (vy1 && vx1) ? wy1 * wx1 : 0.f; v += w_ * a11; }
      const float aw = lg[i] * inv;
      acc += v * aw;
    }
  }
  const int col = hh * p.dh + d;
  if (p.out) *(f32x4*)(p.out + (size_t)q * p.ldo + col) = acc;
  if (p.ohi) {
    half4 h4, l4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { half_t h, lo; split_f16(acc[e], h, lo); h4[e] = h; l4[e] = lo; }
    *(half4*)(p.ohi + (size_t)q * p.ldoh + col) = h4;
    if (p.olo) *(half4*)(p.olo + (size_t)q * p.ldoh + col) = l4;
  }
}

__global__ void box_refine_kernel(const float* __restrict__ delta, int ldd, const float* __restrict__ ref, float eps, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 4) return;
  const int q = i >> 2, c = i & 3;
  const float xc = fminf(fmaxf(ref[i], eps), 1.f - eps);
  const float x = delta[(size_t)q * ldd + c] + logf(xc / (1.f - xc));          // torch.special.logit(x, eps)
  out[i] = 1.0f / (1.0f + expf(-x));
}

__global__ void pad_logits_kernel(const float* __restrict__ x, int ldx, int Q, int T, float* __restrict__ out, int ld) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)Q * ld) return;
  const int q = (int)(i / ld), c = (int)(i - (long)q * ld);
  out[i] = (c < T) ? x[(size_t)q * ldx + c] : -INFINITY;
}

__global__ __launch_bounds__(256) void bert_embed_kernel(const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ typ,
                                                         const int* __restrict__ ids, const int* __restrict__ pids, int T, int D,
                                                         const float* __restrict__ g, const float* __restrict__ b, float eps, float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= T) return;
  const float* w = word + (size_t)ids[row] * D; const float* pp = pos + (size_t)pids[row] * D;
  // same association as the generic chain: (word + pos) + type
  float s = 0.f;
  for (int j = lane; j < D; j += 64) s += (w[j] + pp[j]) + typ[j];
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int j = lane; j < D; j += 64) { const float d = (w[j] + pp[j]) + typ[j] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  for (int j = lane; j < D; j += 64) out[(size_t)row * D + j] = ((w[j] + pp[j]) + typ[j] - mean) * rstd * g[j] + b[j];
}

__global__ void select_ref_kernel(const float* __restrict__ coord, int ldc, const float* __restrict__ prop, const int* __restrict__ idx, int n,
                                  float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 4) return;
  const int q = i >> 2, c = i & 3;
  const int s = idx[q];
  const float x = coord[(size_t)s * ldc + c] + prop[(size_t)s * 4 + c];
  out[i] = 1.0f / (1.0f + expf(-x));
}

}  // namespace

int launch_rowop(const RowOpParams& p, hipStream_t s) {
  if (p.M <= 0) return OVM_OK;
  if (p.il && (!p.hi || p.lo != p.hi + 32 || p.D % 32 || p.ldh != 2 * p.D)) return OVM_ERR_INVALID;     // one interleaved image, no K padding
  const int seg = p.idx ? p.seg : p.D;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  const bool v4 = (p.D % 4 == 0) && (seg % 4 == 0) && (p.ldx % 4 == 0) && al16(p.x) && (!p.res || (p.ldr % 4 == 0 && al16(p.res)));
  const dim3 grid((p.M + 3) / 4), block(256);
  if (p.gamma) {
    if (v4) {
      if (p.D <= 1024) hipLaunchKernelGGL((rowop_kernel<4, 4>), grid, block, 0, s, p);
      else if (p.D <= 4096) hipLaunchKernelGGL((rowop_kernel<4, 16>), grid, block, 0, s, p);
      else return OVM_ERR_SHAPE;
    } else {
      if (p.D <= 1024) hipLaunchKernelGGL((rowop_kernel<1, 16>), grid, block, 0, s, p);
      else return OVM_ERR_SHAPE;
    }
  } else {
    if (v4) hipLaunchKernelGGL((rowop_kernel<4, 1>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((rowop_kernel<1, 1>), grid, block, 0, s, p);
  }
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

bool swin_qkv_attn_supported(int C, int nh, int ws, int npass) {
  return npass == 3 && ws == 12 && nh > 0 && C == nh * 32 && C % 128 == 0;
}

int launch_swin_qkv_attn(const SwinQkvAttnParams& p, hipStream_t s) {
  if (p.nW <= 0) return OVM_OK;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if (p.C != p.nh * 32 || p.C % 128 || p.KS * 32 < p.C || p.KS % 4 || p.ldx % 8 || p.ldo % 4 || !p.xhi || !p.xlo || !p.wfrag || !p.bias || !p.relbias ||
      !p.ohi || !p.olo || !al16(p.xhi) || !al16(p.xlo) || !al16(p.wfrag) || !al16(p.bias) || !al16(p.relbias) || (p.mask && !al16(p.mask)))
    return OVM_ERR_SHAPE;
  if ((long)p.nW * p.nh > 0x7fffffffL) return OVM_ERR_CAPACITY;
  const int smem = (2 * 144 * 36 + 32 * 148) * 4;
  hipLaunchKernelGGL(swin_qkv_attn_kernel, dim3((unsigned)(p.nW * p.nh)), dim3(576), smem, s, p);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int launch_attn_f32(const AttnF32Params& p, hipStream_t s) {
  if (p.Tq <= 0 || p.Tk <= 0 || p.nb1 <= 0 || p.nb2 <= 0) return OVM_OK;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if (p.ldk % 4 || p.ldv % 4 || !al16(p.k) || !al16(p.v) || p.sk1 % 4 || p.sk2 % 4 || p.sv1 % 4 || p.sv2 % 4) return OVM_ERR_SHAPE;
  if (p.o && (p.ldo % 4 || !al16(p.o) || p.so1 % 4 || p.so2 % 4)) return OVM_ERR_SHAPE;
  if (p.ohi && (p.ldoh % 4 || p.soh1 % 4 || p.soh2 % 4)) return OVM_ERR_SHAPE;
  // waves per workgroup: a whole Swin window (144 queries) in one workgroup, else 4 waves (64 queries)
  int nw = (p.Tq + 15) / 16;
  if (nw > 9) nw = 4;
  const int qb = 16 * nw;
  const dim3 grid((p.Tq + qb - 1) / qb, (unsigned)(p.nb1 * p.nb2)), block(64 * nw);
  if ((long)p.nb1 * p.nb2 > 65535) return OVM_ERR_CAPACITY;
  AttnF32Params pv = p;                                   // bias rows readable as float4 (4 keys per lane and key tile)?
  pv.bias_vec = (p.Tk % 4 == 0) && !p.rel_h && (p.bias_h || p.bias_b) &&
                (!p.bias_h || (al16(p.bias_h) && p.ldbh % 4 == 0 && p.sbh % 4 == 0)) &&
                (!p.bias_b || (al16(p.bias_b) && p.ldbb % 4 == 0 && p.sbb % 4 == 0));
#define OVM_ATTN_F32(DH_)                                                                              \
  {                                                                                                    \
    const int smem = kKC * ((DH_) + 4) * 4 + (DH_) * (kKC + 4) * 4;                                    \
    static bool set = false;                                                                           \
    if (!set && smem > 65536) { (void)hipFuncSetAttribute((const void*)attn_f32_kernel<DH_>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); set = true; } \
    hipLaunchKernelGGL((attn_f32_kernel<DH_>), grid, block, smem, s, pv);                               \
  }
  if (p.DH == 16) OVM_ATTN_F32(16)
  else if (p.DH == 32) OVM_ATTN_F32(32)
  else if (p.DH == 64) OVM_ATTN_F32(64)
  else return OVM_ERR_SHAPE;
#undef OVM_ATTN_F32
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

__global__ void relpos_tables_kernel(const float* __restrict__ q, int ldq, int M, int H, int DH, int gh, int gw, const float* __restrict__ Rh,
                                     const float* __restrict__ Rw, float* __restrict__ rel_h, float* __restrict__ rel_w, int ldrel) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per = gh + gw;
  const long total = (long)M * H * per;
  if (idx >= total) return;
  const int j = (int)(idx % per);
  const long mh = idx / per;
  const int h = (int)(mh % H);
  const long m = mh / H;
  const int t = (int)(m % ((long)gh * gw));
  const int qh = t / gw, qw = t - qh * gw;
  const float* qr = q + (size_t)m * ldq + (size_t)h * DH;
  const bool is_h = j < gh;
  const int kk = is_h ? j : j - gh;
  const float* r = is_h ? Rh + (size_t)(qh - kk + gh - 1) * DH : Rw + (size_t)(qw - kk + gw - 1) * DH;
  float acc = 0.f;
  for (int c = 0; c < DH; c += 4) {
    const f32x4 a = *(const f32x4*)(qr + c);
    const f32x4 b = *(const f32x4*)(r + c);
    acc += a[0] * b[0]; acc += a[1] * b[1]; acc += a[2] * b[2]; acc += a[3] * b[3];
  }
  (is_h ? rel_h : rel_w)[(size_t)mh * ldrel + kk] = acc;
}

int launch_relpos_tables(const float* q, int ldq, int M, int H, int DH, int gh, int gw, const float* Rh, const float* Rw, float* rel_h,
                         float* rel_w, int ldrel, hipStream_t s) {
  if (DH % 4 || ldq % 4 || gh > ldrel || gw > ldrel || M <= 0) return M <= 0 ? OVM_OK : OVM_ERR_SHAPE;
  hipLaunchKernelGGL(relpos_tables_kernel, g1((long)M * H * (gh + gw)), dim3(256), 0, s, q, ldq, M, H, DH, gh, gw, Rh, Rw, rel_h, rel_w, ldrel);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int launch_biattn(const BiAttnParams& p, hipStream_t s) {
  if (p.dh % 4 || p.dh > 512 || p.chunk > 128 || p.chunk <= 0) return OVM_ERR_SHAPE;
  if (p.dh <= 256) hipLaunchKernelGGL(biattn_img_kernel<1>, dim3((unsigned)(((long)p.S * p.H + 3) / 4)), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(biattn_img_kernel<2>, dim3((unsigned)(((long)p.S * p.H + 3) / 4)), dim3(256), 0, s, p);
  hipLaunchKernelGGL(biattn_stats_kernel, dim3(p.H * p.T), dim3(256), 0, s, p);
  hipLaunchKernelGGL(biattn_txt_partial_kernel, dim3(p.nchunk, p.H), dim3(256), 0, s, p);
  hipLaunchKernelGGL(biattn_txt_combine_kernel, g1((long)p.T * p.H * p.dh), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

static int g_msdeform_vec = 1;          // ovm_tune_set("msdeform_vec", 0): the one-thread-per-channel kernel
void msdeform_set_vec(int v) { g_msdeform_vec = v; }

int launch_msdeform_fused(const MsDeformParams& p, hipStream_t s) {
  if (p.L > 8 || p.Q <= 0) return p.Q <= 0 ? OVM_OK : OVM_ERR_CAPACITY;
  const bool al16 = !(((uintptr_t)p.value | (uintptr_t)p.ow | (uintptr_t)p.out | (uintptr_t)p.ohi | (uintptr_t)p.olo) & 15);
  if (g_msdeform_vec && p.L * p.P == 16 && p.dh % 4 == 0 && p.ldv % 4 == 0 && p.ldow % 4 == 0 && p.ldo % 4 == 0 && p.ldoh % 4 == 0 &&
      (p.H * p.L * p.P * 2) % 4 == 0 && al16) {
    hipLaunchKernelGGL(msdeform_fused4_kernel<16>, g1((long)p.Q * p.H * (p.dh / 4)), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
  }
  hipLaunchKernelGGL(msdeform_fused_kernel, g1((long)p.Q * p.H * p.dh), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int launch_box_refine(const float* delta, int ldd, const float* ref, float eps, float* out, int n, hipStream_t s) {
  hipLaunchKernelGGL(box_refine_kernel, g1((long)n * 4), dim3(256), 0, s, delta, ldd, ref, eps, out, n);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}
int launch_pad_logits(const float* x, int ldx, int Q, int T, float* out, int ld, hipStream_t s) {
  hipLaunchKernelGGL(pad_logits_kernel, g1((long)Q * ld), dim3(256), 0, s, x, ldx, Q, T, out, ld);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}
int launch_bert_embed(const float* word, const float* pos, const float* typ, const int* ids, const int* pids, int T, int D, const float* g,
                      const float* b, float eps, float* out, hipStream_t s) {
  hipLaunchKernelGGL(bert_embed_kernel, dim3((T + 3) / 4), dim3(256), 0, s, word, pos, typ, ids, pids, T, D, g, b, eps, out);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}
int launch_select_ref(const float* coord, int ldc, const float* prop_logit, const int* idx, int n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(select_ref_kernel, g1((long)n * 4), dim3(256), 0, s, coord, ldc, prop_logit, idx, n, out);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

}  // namespace ovm
