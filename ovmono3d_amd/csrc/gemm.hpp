// MFMA GEMM for gfx950: C[m][n] = sum_k A[m][k] * W[n][k]   (both operands K-contiguous fp16)
//
//  * BM x 128 block tile (BM = 128 or 256; 4 or 8 waves as (BM/64)(M) x 2(N)), each wave 64x64 = 4x4 MFMA tiles of
//    v_mfma_f32_16x16x32_f16 (W rows on the MFMA "A" side, activation rows on the "B" side, so a
//    lane's 4 accumulator registers run along n: 16-byte epilogue vectors).
//  * Operands staged global -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two LDS
//    stages; swizzle applied on the per-lane SOURCE address and again on the ds_read_b128 address
//    (LDS image itself is lane-linear, as the DMA requires).
//  * NPASS=1: plain fp16 operands, fp32 accumulate.  NPASS=3: split operands (hi + lo), three MFMAs per
//    product into the same accumulator - fp32-class accuracy at 1/3 of the fp16 rate (still ~5x the
//    fp32-MFMA rate).
//  * AMODE selects how a row of A is addressed: dense row-major, or implicit-GEMM 3x3 convolution
//    over a zero-bordered NHWC fp16 image (k = tap*C + c).
//  * EPI selects the fused epilogue (bias/LayerScale/residual, GELU, QKV head split with V^T
//    layout, patch-embed + pos-embed, ConvTranspose 2x2 scatter, generic store).
#pragma once
#include "common.hpp"

namespace ovm {

enum AMode { A_ROWMAJOR = 0, A_CONV3X3 = 1 };
enum Epi { EPI_STORE = 0, EPI_RESID = 1, EPI_GELU = 2, EPI_QKV = 3, EPI_PATCH = 4, EPI_CONVT = 5 };

struct GemmParams {
  const half_t* Ahi; const half_t* Alo; int lda;
  // W: [Npad][K] fp16 (one-pass mode), or - split mode - interleaved [Npad][K/32][hi 32 | lo 32] (Wlo = Whi + 32, row
  // stride 2K): one 128-byte line per row and 32-wide k-step holds both parts, so an LDS-DMA wave-instruction fetches 8 rows x
  // 128 B instead of 16 rows x 64 B (measured 1.56x the LDS-DMA rate, scratch/micro/dma_bench.hip). Npad multiple of 128.
  const half_t* Whi; const half_t* Wlo;
  int ldw;                                        // set by the launcher: W row stride in halves (K or 2K)
  int a_il;                                       // split mode: A is interleaved the same way (Alo = Ahi + 32, lda = 2K)
  // split-K (wave-specialised kernel, set by the launcher for thin grids with a long K): workgroup = (tile, k-slice); slices
  // write raw fp32 partial tiles to `part` [ksplit][M][N] and splitk_epilogue_kernel sums them and applies the epilogue
  int ksplit, kchunk; float* part;
  unsigned long long* stamps;                     // diagnostic builds only (gemm256 stamp variant): s_memtime per barrier, [wave][128]
  float* part_ws; size_t part_cap;                // caller-owned split-K workspace (bytes); null: the launcher's process-global one
                                                  // (which may be re-allocated - not usable under HIP-graph capture / replay)
  int ws_slot;                                    // which split-K workspace the launcher may use: callers that run concurrently on
                                                  // different streams (engine = 0, generic GroundingDINO ops = 1) must not share one
  int M, N, K;
  // A_CONV3X3: A is [B][cH+2][cW+2][cC] fp16 with a zero border, m = (b*cH + y)*cW + x
  int cH, cW, cC;
  // epilogue operands
  const float* bias;                              // [N] or null
  const float* gamma;                             // EPI_RESID: LayerScale [N]
  float* X; int ldx;                              // EPI_RESID: fp32 residual stream, in place (gamma null: plain residual)
  const int* row_map;                             // EPI_RESID: optional, row m updates X[row_map[m]] (negative: dropped) - the
                                                  // reverse window partition / un-shift / crop of a Swin block
  float* C; int ldc;                              // EPI_STORE: fp32 out (or null)
  half_t* Ohi; half_t* Olo; int ldo;              // EPI_STORE / EPI_GELU / EPI_CONVT: fp16 split out (or null)
  int o_il;                                       // EPI_GELU: Ohi / Olo form an interleaved image (Olo = Ohi + 32, ldo = 2 N)
  int relu;                                       // EPI_STORE activation: 0 none, 1 ReLU, 2 GELU(erf); EPI_GELU: 3 = QuickGELU, else erf
  const float* R; int ldr;                        // EPI_STORE: optional fp32 residual added after the activation
  // EPI_STORE with padded-NHWC destination for Ohi/Olo (conv input): if padH>0, row m=(b,y,x) goes to
  // ((b*(padH+2) + y+1)*(padW+2) + x+1)
  int padH, padW;
  // EPI_QKV
  half_t *Qhi, *Qlo, *Khi, *Klo, *Vhi, *Vlo; int T, Tpad, heads; float qscale;
  // EPI_PATCH: X[(b*T + c + p)][n] = acc + bias[n] + pos[(c+p)*N + n],   m = b*G2 + p, c = T - G2 leading (class) tokens (1 or 0)
  const float* pos; int G2;
  // EPI_CONVT: m = (b, i, j) over GxG, n = (a*2 + bb)*Cout + co -> NHWC [B][2G][2G][Cout]
  int G, Cout;
  // set by the launcher: workgroups [0, main_tiles) run MFMA tiles over rows [0, tail_begin); workgroups beyond
  // them compute the leftover rows [tail_begin, M) with the dot-product body (see gemm_tail_body)
  int main_tiles, tail_begin, M_total;
  int tail_waves;                                 // waves per leftover-row workgroup that work (0 = all of them)
};

template <int BK> __device__ __forceinline__ int swz_slot(int row, int chunk);
// 128-byte rows (BK=64): 8 chunks of 16 B. Conflict-free for ds_read_b128 of 16 or 32 distinct
// rows at one chunk column (16x16x32 and 32x32x16 operand reads).
template <> __device__ __forceinline__ int swz_slot<64>(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
// 64-byte rows (BK=32): 4 chunks; g = {0,2,3,1}[(row>>2)&3]
template <> __device__ __forceinline__ int swz_slot<32>(int row, int chunk) {
  return chunk ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3);
}

template <int AMODE>
__device__ __forceinline__ uint32_t a_row_offset(const GemmParams& p, int m) {
  if (AMODE == A_ROWMAJOR) return (uint32_t)m * (uint32_t)p.lda;
  const int x = m % p.cW; const int t = m / p.cW; const int y = t % p.cH; const int b = t / p.cH;
  return (uint32_t)(((b * (p.cH + 2) + y) * (p.cW + 2) + x) * p.cC);
}
template <int AMODE>
__device__ __forceinline__ uint32_t a_k_offset(const GemmParams& p, int k0) {
  if (AMODE == A_ROWMAJOR) return (uint32_t)k0;
  const int tap = k0 / p.cC, c0 = k0 - tap * p.cC;
  const int dy = tap / 3, dx = tap - dy * 3;
  return (uint32_t)((dy * (p.cW + 2) + dx) * p.cC + c0);
}


// Staging geometry of one operand side (A or W) of a k-tile in LDS.
//   one-pass:            rows of BK*2 bytes (BK = 64: 128 B), one part
//   split, plain arrays: rows of 64 B (BK = 32), two parts (hi rows, then lo rows)
//   split, interleaved:  rows of 128 B = [hi 32 | lo 32], one part
template <int NPASS, int BK, bool IL>
struct Side {
  static constexpr int ROWB = IL ? 128 : BK * 2;
  static constexpr int PARTS = (NPASS == 3 && !IL) ? 2 : 1;
  static constexpr int RPI = 1024 / ROWB;                    // rows one 1-KiB wave-instruction covers
  static constexpr int CH = ROWB / 16;                       // 16-byte chunks per row
  static constexpr int KSTEP = IL ? 64 : BK;                 // halves a k-tile advances inside a row
  __host__ __device__ static constexpr int bytes(int rows) { return rows * ROWB * PARTS; }
  __device__ __forceinline__ static int swz(int row, int c) { return ROWB == 128 ? swz_slot<64>(row, c) : swz_slot<32>(row, c); }
  // byte offsets of the hi / lo fragment (8 halves at k = 32 kk + 8 fq) of `row`, relative to the side's region
  __device__ __forceinline__ static int frag_hi(int row, int kk, int fq) { return row * ROWB + swz(row, IL ? fq : kk * 4 + fq) * 16; }
  __device__ __forceinline__ static int frag_lo(int rows, int row, int kk, int fq) {
    return IL ? row * ROWB + swz(row, 4 + fq) * 16 : rows * ROWB + row * ROWB + swz(row, kk * 4 + fq) * 16;
  }
};
template <int AMODE, bool AIL>
__device__ __forceinline__ uint32_t a_row_off(const GemmParams& p, int m) {
  const uint32_t o = a_row_offset<AMODE>(p, m);
  return (AIL && AMODE != A_ROWMAJOR) ? o * 2 : o;           // row-major: the caller's lda already is 2K
}
template <int AMODE, bool AIL>
__device__ __forceinline__ uint32_t a_k_off(const GemmParams& p, int k0) {
  const uint32_t o = a_k_offset<AMODE>(p, k0);
  return AIL ? o * 2 : o;                                    // k0 is a multiple of 32: group g starts at 64 g
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// open_clip QuickGELU (the activation of the OpenAI CLIP weights): x * sigmoid(1.702 x)
__device__ __forceinline__ float gelu_quick(float x) { return x / (1.0f + expf(-1.702f * x)); }

template <int EPI>
__device__ __forceinline__ void epilogue4(const GemmParams& p, int m, int n, f32x4 v) {
  // v[r] belongs to (m, n + r); n is a multiple of 4; caller guarantees m < M; n + r may be >= N.
  if (n >= p.N) return;
  float b4[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) b4[r] = (n + r < p.N) ? p.bias[n + r] : 0.f;
  }
  const bool full = (n + 3 < p.N);
  if (EPI == EPI_RESID) {
    int mr = m;
    if (p.row_map) { mr = p.row_map[m]; if (mr < 0) return; }
    float* x = p.X + (size_t)mr * p.ldx + n;
    if (full) {
      f32x4 xv = *(const f32x4*)x;
      const f32x4 g = p.gamma ? *(const f32x4*)(p.gamma + n) : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) xv[r] += g[r] * (v[r] + b4[r]);
      *(f32x4*)x = xv;
    } else {
      for (int r = 0; r < 4 && n + r < p.N; ++r) x[r] += (p.gamma ? p.gamma[n + r] : 1.f) * (v[r] + b4[r]);
    }
  } else if (EPI == EPI_GELU) {
    half4 h, l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float z = v[r] + b4[r];
      half_t hh, ll; split_f16(p.relu == 3 ? gelu_quick(z) : gelu_erf(z), hh, ll); h[r] = hh; l[r] = ll;
    }
    const size_t o = (size_t)m * p.ldo + (p.o_il ? il_col(n) : n);
    *(half4*)(p.Ohi + o) = h;
    if (p.Olo) *(half4*)(p.Olo + o) = l;
  } else if (EPI == EPI_QKV) {
    const int Dm = p.N / 3;
    const int which = n / Dm;
    const int f = n - which * Dm;
    const int head = f >> 6, d = f & 63;
    const int b = m / p.T, t = m - b * p.T;
    half4 h, l;
    const float sc = (which == 0) ? p.qscale : 1.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { half_t hh, ll; split_f16((v[r] + b4[r]) * sc, hh, ll); h[r] = hh; l[r] = ll; }
    if (which < 2) {
      const size_t o = ((size_t)(b * p.heads + head) * p.T + t) * 64 + d;
      half_t* dh = (which == 0) ? p.Qhi : p.Khi;
      half_t* dl = (which == 0) ? p.Qlo : p.Klo;
      *(half4*)(dh + o) = h;
      if (dl) *(half4*)(dl + o) = l;
    } else {
      // V^T [b][head][d][Tpad], token order permuted inside each group of 16 (bits 2<->3 swapped)
      // so that the PV MFMA's A fragment (k order 4h+{0..3}, 8+4h+{0..3}) is one 16-byte LDS read.
      const int tp = (t & ~15) | (t & 3) | ((t & 4) << 1) | ((t & 8) >> 1);
      const size_t o = ((size_t)(b * p.heads + head) * 64 + d) * p.Tpad + tp;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p.Vhi[o + (size_t)r * p.Tpad] = h[r];
        if (p.Vlo) p.Vlo[o + (size_t)r * p.Tpad] = l[r];
      }
    }
  } else if (EPI == EPI_PATCH) {
    const int b = m / p.G2, pp = m - b * p.G2, c0 = p.T - p.G2;
    float* x = p.X + ((size_t)b * p.T + c0 + pp) * p.ldx + n;
    const float* ps = p.pos + (size_t)(c0 + pp) * p.N + n;
    for (int r = 0; r < 4 && n + r < p.N; ++r) x[r] = v[r] + b4[r] + ps[r];
  } else if (EPI == EPI_CONVT) {
    const int q = n / p.Cout, co = n - q * p.Cout;       // q = a*2 + bb; Cout % 4 == 0
    const int a = q >> 1, bb = q & 1;
    const int j = m % p.G; const int t = m / p.G; const int i = t % p.G; const int b = t / p.G;
    const size_t o = (((size_t)b * 2 * p.G + 2 * i + a) * 2 * p.G + 2 * j + bb) * p.Cout + co;
    half4 h, l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float bv = p.bias ? p.bias[co + r] : 0.f;
      half_t hh, ll; split_f16(v[r] + bv, hh, ll); h[r] = hh; l[r] = ll;
    }
    *(half4*)(p.Ohi + o) = h;
    if (p.Olo) *(half4*)(p.Olo + o) = l;
  } else {  // EPI_STORE
    float o4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o4[r] = v[r] + b4[r];
      if (p.relu == 1) o4[r] = fmaxf(o4[r], 0.f); else if (p.relu == 2) o4[r] = gelu_erf(o4[r]);
      if (p.R && n + r < p.N) o4[r] += p.R[(size_t)m * p.ldr + n + r];
    }
    if (p.C) {
      float* c = p.C + (size_t)m * p.ldc + n;
      for (int r = 0; r < 4 && n + r < p.N; ++r) c[r] = o4[r];
    }
    if (p.Ohi) {
      size_t row = (size_t)m;
      if (p.padH > 0) {
        const int x = m % p.padW; const int t = m / p.padW; const int y = t % p.padH; const int b = t / p.padH;
        row = ((size_t)b * (p.padH + 2) + y + 1) * (p.padW + 2) + x + 1;
      }
      const size_t o = row * p.ldo + n;
      for (int r = 0; r < 4 && n + r < p.N; ++r) {
        half_t hh, ll; split_f16(o4[r], hh, ll);
        p.Ohi[o + r] = hh;
        if (p.Olo) p.Olo[o + r] = ll;
      }
    }
  }
}

template <int NPASS, int EPI, int AMODE, bool AIL>
__device__ __forceinline__ void gemm_tail_body(const GemmParams& p, int tb);

template <int NPASS, int BK, int BM, int NSTAGE, int EPI, int AMODE, bool AIL = false>
__global__ __launch_bounds__(BM * 2) void gemm_kernel(const GemmParams p) {
  if ((int)blockIdx.x >= p.main_tiles) { gemm_tail_body<NPASS, EPI, AMODE, AIL>(p, (int)blockIdx.x - p.main_tiles); return; }
  // Block tile BM x 128 (BM = 128: 4 waves, BM = 256: 8 waves); every wave owns a 64 x 64 sub-tile.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = BM / 32;                     // waves per block
  using SA = Side<NPASS, BK, AIL && NPASS == 3>;
  using SW = Side<NPASS, BK, NPASS == 3>;         // split-mode weights are always interleaved
  constexpr int OFF_A = 0, OFF_W = SA::bytes(BM);
  constexpr int STAGE = SA::bytes(BM) + SW::bytes(128);
  constexpr int IA = BM / SA::RPI / NW;           // A DMA instructions per wave per part
  constexpr int NIW = 128 / SW::RPI;              // W DMA instructions per part, whole workgroup
  constexpr int IW = (NIW + NW - 1) / NW;         // ... per wave (may be partial)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = p.main_tiles / tiles_m;
  // XCD-aware remap, then grouped ordering (8 row-tiles per group) so the tiles that run concurrently on one
  // XCD form a compact 2-D patch: each A / W k-slice is fetched into that XCD's L2 once and reused.
  int pid = xcd_remap(blockIdx.x, p.main_tiles);
  constexpr int GROUP_M = 8;
  const int in_group = GROUP_M * tiles_n;
  const int gid = pid / in_group;
  const int first_m = gid * GROUP_M;
  const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
  const int tm = first_m + (pid % in_group) % gsz;
  const int tn = (pid % in_group) / gsz;
  const int m0 = tm * BM, n0 = tn * 128;
  const int wm = wave >> 1, wn = wave & 1;

  uint32_t aoff[IA], woff[IW];
#pragma unroll
  for (int t = 0; t < IA; ++t) {
    const int instr = wave + NW * t;
    const int row = instr * SA::RPI + lane / SA::CH;
    const int chunk = SA::swz(row, lane % SA::CH);
    int m = m0 + row; if (m > p.M - 1) m = p.M - 1;
    aoff[t] = a_row_off<AMODE, AIL && NPASS == 3>(p, m) + chunk * 8;
  }
#pragma unroll
  for (int t = 0; t < IW; ++t) {
    const int instr = wave + NW * t;
    const int row = (instr % NIW) * SW::RPI + lane / SW::CH;
    const int chunk = SW::swz(row, lane % SW::CH);
    woff[t] = (uint32_t)(n0 + row) * (uint32_t)p.ldw + chunk * 8;
  }

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
    const uint32_t ak = a_k_off<AMODE, AIL && NPASS == 3>(p, kt * BK);
    const uint32_t wk = (uint32_t)(kt * SW::KSTEP);
#pragma unroll
    for (int t = 0; t < IA; ++t) {
      char* dst = base + OFF_A + (wave + NW * t) * 1024;
      glds16(p.Ahi + aoff[t] + ak, dst);
      if (SA::PARTS == 2) glds16(p.Alo + aoff[t] + ak, dst + BM * SA::ROWB);
    }
#pragma unroll
    for (int t = 0; t < IW; ++t) {
      const int instr = wave + NW * t;
      if (instr < NIW) {                           // wave-uniform
        char* dst = base + OFF_W + instr * 1024;
        glds16(p.Whi + woff[t] + wk, dst);
        if (SW::PARTS == 2) glds16(p.Wlo + woff[t] + wk, dst + 128 * SW::ROWB);
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  // glds instructions one wave issues per stage (for the counted vmcnt of the 3-stage ring)
  constexpr int PER_STAGE = IA * SA::PARTS + ((NIW >= NW) ? IW : 0) * SW::PARTS;
  static_assert(NSTAGE == 2 || (NIW % NW == 0 || NIW < NW), "3-stage ring needs a wave-uniform DMA count");
  // s_waitcnt immediate: vmcnt = N, expcnt / lgkmcnt untouched
  constexpr int WAIT_ONE_STAGE = (PER_STAGE & 0xF) | (7 << 4) | (15 << 8) | ((PER_STAGE >> 4) << 14);
  const int fr = lane & 15, fq = lane >> 4;

  auto compute = [&](const char* base) {
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      half8 ah[4], wh[4], al[4], wl[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra = wm * 64 + i * 16 + fr;
        const int rw = wn * 64 + i * 16 + fr;
        ah[i] = *(const half8*)(base + OFF_A + SA::frag_hi(ra, kk, fq));
        wh[i] = *(const half8*)(base + OFF_W + SW::frag_hi(rw, kk, fq));
        if (NPASS == 3) {
          al[i] = *(const half8*)(base + OFF_A + SA::frag_lo(BM, ra, kk, fq));
          wl[i] = *(const half8*)(base + OFF_W + SW::frag_lo(128, rw, kk, fq));
        }
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          if (NPASS == 3) {
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ni], ah[mi], acc[ni][mi], 0, 0, 0);
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ni], al[mi], acc[ni][mi], 0, 0, 0);
          }
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ni], ah[mi], acc[ni][mi], 0, 0, 0);
        }
    }
  };

  if (NSTAGE == 2) {
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
      compute(smem + cur * STAGE);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  } else {
    // 3-slot ring, two k-tiles in flight: at the end of iteration kt only tile kt+1 must have landed, so the
    // wait is a counted vmcnt that leaves tile kt+2's DMA outstanding across a raw s_barrier.
    stage(0, 0);
    if (nk > 1) { stage(1, 1); __builtin_amdgcn_s_waitcnt(WAIT_ONE_STAGE); } else { __builtin_amdgcn_s_waitcnt(0x0F70 & ~0xF); }
    __builtin_amdgcn_s_barrier();
    int cur = 0, nxt2 = 2;
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = (kt + 2 < nk);
      if (more) stage(nxt2, kt + 2);
      compute(smem + cur * STAGE);
      if (more) __builtin_amdgcn_s_waitcnt(WAIT_ONE_STAGE); else __builtin_amdgcn_s_waitcnt(0x0F70 & ~0xF);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = (cur == 2) ? 0 : cur + 1;
      nxt2 = (nxt2 == 2) ? 0 : nxt2 + 1;
    }
  }

  // ---- epilogue: lane holds (m = m0+wm*64+mi*16+fr, n = n0+wn*64+ni*16+fq*4 .. +3)
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + fq * 4;
      epilogue4<EPI>(p, m, n, acc[ni][mi]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised variant: 128 x 128 tile, 8 waves. Waves 4-7 (one per SIMD) only move operands: they keep NS-1
// k-tiles of LDS-DMA in flight through an NS-slot ring and publish a tile with a counted vmcnt wait + one
// s_barrier per k-step. Waves 0-3 (one per SIMD, 64 x 64 each) only read fragments and issue MFMAs; they fetch the
// fragments of tile t+1 into a second register set right after the barrier that publishes it, i.e. under the MFMAs
// of tile t. The MFMA stream of a SIMD is therefore never interrupted by DMA issue, address arithmetic or by
// waiting for global memory, which is what bounded the symmetric kernel above (SQ_WAIT_ANY ~ 47 %).
// One barrier per k-step is enough: barrier(t) says "tile t landed" to the consumers and "tile t-1 consumed"
// (its slot may be refilled with tile t+NS-1) to the producers.
// ---------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

template <int NPASS, int BK, int NS, int EPI, int AMODE, bool AIL = false>
__global__ __launch_bounds__(512) void gemm_ws_kernel(const GemmParams p) {
  if ((int)blockIdx.x >= p.main_tiles) { gemm_tail_body<NPASS, EPI, AMODE, AIL>(p, (int)blockIdx.x - p.main_tiles); return; }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 128, NPW = 4;
  using SA = Side<NPASS, BK, AIL && NPASS == 3>;
  using SW = Side<NPASS, BK, NPASS == 3>;
  constexpr int OFF_A = 0, OFF_W = SA::bytes(BM);
  constexpr int STAGE = SA::bytes(BM) + SW::bytes(128);
  constexpr int IA = BM / SA::RPI / NPW;
  constexpr int IW = 128 / SW::RPI / NPW;
  constexpr int PER_STAGE = IA * SA::PARTS + IW * SW::PARTS;        // DMA pieces one producer wave issues per k-tile
  static_assert(NS >= 3 && NS <= 4 && (NS - 2) * PER_STAGE < 64, "ring depth vs vmcnt range");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  const int tiles = p.main_tiles / ksplit;               // output tiles; main_tiles counts (tile, k-slice) workgroups
  const int tiles_n = tiles / tiles_m;
  const int ks = (int)blockIdx.x / tiles;
  int pid = xcd_remap((int)blockIdx.x - ks * tiles, tiles);
  constexpr int GROUP_M = 8;
  const int in_group = GROUP_M * tiles_n;
  const int gid = pid / in_group;
  const int first_m = gid * GROUP_M;
  const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
  const int tm = first_m + (pid % in_group) % gsz;
  const int tn = (pid % in_group) / gsz;
  const int m0 = tm * BM, n0 = tn * 128;
  const int nk_all = p.K / BK;
  const int kt0 = (ksplit > 1) ? ks * p.kchunk : 0;       // first k-tile of this slice
  const int nk = (ksplit > 1) ? ((nk_all - kt0 < p.kchunk) ? nk_all - kt0 : p.kchunk) : nk_all;

  if (wave >= 4) {
    // ---------------- producers ----------------
    const int pw = wave - 4;
    uint32_t aoff[IA], woff[IW];
#pragma unroll
    for (int t = 0; t < IA; ++t) {
      const int row = (pw + NPW * t) * SA::RPI + lane / SA::CH;
      const int chunk = SA::swz(row, lane % SA::CH);
      int m = m0 + row; if (m > p.M - 1) m = p.M - 1;
      aoff[t] = a_row_off<AMODE, AIL && NPASS == 3>(p, m) + chunk * 8;
    }
#pragma unroll
    for (int t = 0; t < IW; ++t) {
      const int row = (pw + NPW * t) * SW::RPI + lane / SW::CH;
      const int chunk = SW::swz(row, lane % SW::CH);
      woff[t] = (uint32_t)(n0 + row) * (uint32_t)p.ldw + chunk * 8;
    }
    auto stage = [&](int slot, int kt) {
      char* base = smem + slot * STAGE;
      const uint32_t ak = a_k_off<AMODE, AIL && NPASS == 3>(p, (kt0 + kt) * BK);
      const uint32_t wk = (uint32_t)((kt0 + kt) * SW::KSTEP);
#pragma unroll
      for (int t = 0; t < IA; ++t) {
        char* dst = base + OFF_A + (pw + NPW * t) * 1024;
        glds16(p.Ahi + aoff[t] + ak, dst);
        if (SA::PARTS == 2) glds16(p.Alo + aoff[t] + ak, dst + BM * SA::ROWB);
      }
#pragma unroll
      for (int t = 0; t < IW; ++t) {
        char* dst = base + OFF_W + (pw + NPW * t) * 1024;
        glds16(p.Whi + woff[t] + wk, dst);
        if (SW::PARTS == 2) glds16(p.Wlo + woff[t] + wk, dst + 128 * SW::ROWB);
      }
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
      if (t < nk) stage(t, t);
    int slot_next = NS - 1;                                   // slot of tile kt + NS - 1
    for (int kt = 0; kt < nk; ++kt) {
      const int ahead = (nk - 1 - kt < NS - 2) ? nk - 1 - kt : NS - 2;    // tiles issued after tile kt
      if (ahead >= 2) wait_vmcnt<2 * PER_STAGE>();
      else if (ahead == 1) wait_vmcnt<PER_STAGE>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (kt + NS - 1 < nk) stage(slot_next, kt + NS - 1);
      slot_next = (slot_next + 1 == NS) ? 0 : slot_next + 1;
    }
    __builtin_amdgcn_s_barrier();                             // ring drained: the consumers reuse it for the epilogue
    return;
  }

  // ---------------- consumers ----------------
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NF = 4 * (BK / 32);                            // fragments per operand part per k-tile
  half8 ahA[NF], whA[NF], alA[NF], wlA[NF], ahB[NF], whB[NF], alB[NF], wlB[NF];
  int oah[NF], oal[NF], owh[NF], owl[NF];
#pragma unroll
  for (int kk = 0; kk < BK / 32; ++kk)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ra = wm * 64 + i * 16 + fr, rw = wn * 64 + i * 16 + fr;
      oah[kk * 4 + i] = OFF_A + SA::frag_hi(ra, kk, fq);
      oal[kk * 4 + i] = OFF_A + SA::frag_lo(BM, ra, kk, fq);
      owh[kk * 4 + i] = OFF_W + SW::frag_hi(rw, kk, fq);
      owl[kk * 4 + i] = OFF_W + SW::frag_lo(128, rw, kk, fq);
    }
#define OVM_WS_READ(S, base)                                                            \
  _Pragma("unroll") for (int f = 0; f < NF; ++f) {                                      \
    ah##S[f] = *(const half8*)((base) + oah[f]);                                        \
    wh##S[f] = *(const half8*)((base) + owh[f]);                                        \
    if (NPASS == 3) {                                                                   \
      al##S[f] = *(const half8*)((base) + oal[f]);                                      \
      wl##S[f] = *(const half8*)((base) + owl[f]);                                      \
    }                                                                                   \
  }
#define OVM_WS_MFMA(S)                                                                  \
  _Pragma("unroll") for (int kk = 0; kk < BK / 32; ++kk)                                \
  _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                      \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                    \
    if (NPASS == 3) {                                                                   \
      acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl##S[kk * 4 + ni], ah##S[kk * 4 + mi], acc[ni][mi], 0, 0, 0); \
      acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh##S[kk * 4 + ni], al##S[kk * 4 + mi], acc[ni][mi], 0, 0, 0); \
    }                                                                                   \
    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh##S[kk * 4 + ni], ah##S[kk * 4 + mi], acc[ni][mi], 0, 0, 0);   \
  }

  __builtin_amdgcn_s_barrier();                               // barrier(0): tile 0 landed
  OVM_WS_READ(A, smem)
  // steady state, two k-tiles per trip, branch-free: publish-wait, fetch the next tile's fragments, run this tile's MFMAs.
  // lgkmcnt(0) is issued as the builtin: inline asm is invisible to the compiler's wait-count pass, which would then
  // re-wait for the freshly issued reads in front of the MFMAs.
  int kt = 0, slot = 1;                                       // slot = ring slot of tile kt + 1
  // sched_barrier(0) pins the order "all fragment reads of the next tile, then this tile's MFMAs": left alone the
  // scheduler sinks the reads next to their uses (one trip later) and the MFMAs end up waiting for them.
  for (; kt + 2 < nk; kt += 2) {
    __builtin_amdgcn_s_waitcnt(0xC07F);                       // tile kt sits in registers: its slot may be refilled
    __builtin_amdgcn_s_barrier();                             // barrier(kt+1)
    OVM_WS_READ(B, smem + slot * STAGE)
    slot = (slot + 1 == NS) ? 0 : slot + 1;
    __builtin_amdgcn_sched_barrier(0);
    OVM_WS_MFMA(A)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                             // barrier(kt+2)
    OVM_WS_READ(A, smem + slot * STAGE)
    slot = (slot + 1 == NS) ? 0 : slot + 1;
    __builtin_amdgcn_sched_barrier(0);
    OVM_WS_MFMA(B)
    __builtin_amdgcn_sched_barrier(0);
  }
  if (kt + 1 < nk) {                                          // two tiles left
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    OVM_WS_READ(B, smem + slot * STAGE)
    __builtin_amdgcn_sched_barrier(0);
    OVM_WS_MFMA(A)
    OVM_WS_MFMA(B)
  } else {                                                    // one tile left
    OVM_WS_MFMA(A)
  }
#undef OVM_WS_READ
#undef OVM_WS_MFMA

  // Epilogue through LDS: the MFMA layout gives a lane 4 columns of 16 different rows (64-byte, or for fp16 outputs
  // 32-byte, runs per row). Each consumer stages its 64 x 64 fp32 tile in the drained ring and re-reads it row-major,
  // so that 16 lanes cover one row (256 B fp32 / 128 B fp16 per row and instruction: whole cache lines).
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  constexpr int TLD = 68;                                      // padded row stride (floats): conflict-free both ways
  float* tile = (float*)smem + wave * (64 * TLD);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      *(f32x4*)(tile + (mi * 16 + fr) * TLD + ni * 16 + fq * 4) = acc[ni][mi];
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int mb = m0 + wm * 64, nb = n0 + wn * 64;
  bool vt_tile = false;
  if (EPI == EPI_QKV && ksplit == 1) vt_tile = (nb / (p.N / 3)) == 2;   // wave-uniform: head blocks are 64 wide
  if (!vt_tile) {
    const int col = (lane & 15) * 4;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int row = it * 4 + (lane >> 4);
      const f32x4 v = *(const f32x4*)(tile + row * TLD + col);
      if (mb + row < p.M) {
        if (ksplit > 1) { if (nb + col < p.N) *(f32x4*)(p.part + ((size_t)ks * p.M + mb + row) * p.N + nb + col) = v; }   // N % 4 == 0 (launcher)
        else epilogue4<EPI>(p, mb + row, nb + col, v);
      }
    }
  } else {
    // V^T [b][head][d][Tpad]: tokens are the contiguous axis, so lanes run along m (one 2-byte element each, 128 B per store)
    const int m = mb + lane;
    if (m < p.M) {
      const int Dm = p.N / 3;
      const int f0 = nb - 2 * Dm;
      const int head = f0 >> 6;
      const int b = m / p.T, t = m - b * p.T;
      const int tp = (t & ~15) | (t & 3) | ((t & 4) << 1) | ((t & 8) >> 1);
      const size_t o0 = ((size_t)(b * p.heads + head) * 64) * p.Tpad + tp;
      for (int d = 0; d < 64; ++d) {
        float x = tile[lane * TLD + d];
        if (p.bias) x += p.bias[nb + d];
        half_t hh, ll; split_f16(x, hh, ll);
        p.Vhi[o0 + (size_t)d * p.Tpad] = hh;
        if (p.Vlo) p.Vlo[o0 + (size_t)d * p.Tpad] = ll;
      }
    }
  }
}


// Sums the split-K partial tiles of gemm_ws_kernel and applies the epilogue: one thread per (row, 4 columns).
template <int EPI>
__global__ void splitk_epilogue_kernel(const GemmParams p) {
  const int nq = p.N / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)p.M * nq) return;
  const int m = (int)(i / nq), n = (int)(i - (long)m * nq) * 4;
  f32x4 v = *(const f32x4*)(p.part + (size_t)m * p.N + n);
  for (int ks = 1; ks < p.ksplit; ++ks) v += *(const f32x4*)(p.part + ((size_t)ks * p.M + m) * p.N + n);
  epilogue4<EPI>(p, m, n, v);
}

// ---------------------------------------------------------------------------------------------
// Tail rows: M = B*T is 4097 for the reference canvas (64x64 patches + cls), one row past a multiple of
// the tile height. Running that row as a 33rd row of tiles costs a full extra round of workgroups on the
// 256 CUs, so up to 8 leftover rows are computed by this wave-per-4-columns dot-product kernel instead
// (fp32 FMA on the reconstructed hi+lo operands, same epilogues).
// ---------------------------------------------------------------------------------------------
template <int NPASS, int EPI, int AMODE, bool AIL>
__device__ __forceinline__ void gemm_tail_body(const GemmParams& p, int tb) {
  // tb = tail workgroup index; every wave computes 4 consecutive columns of one leftover row
  // Only the first p.tail_waves waves of a workgroup work (set by the launcher, gemm_tail_waves): the leftover row needs EVERY row of W,
  // and a second round of a few fat workgroups reads it through a few CUs (fc2 at ViT-L: 32 workgroups x 512 KB = +7.9 us on a 91-us
  // launch); thin workgroups spread the same bytes over >= 128 CUs.
  const int lane = threadIdx.x & 63;
  const int waves = p.tail_waves > 0 ? p.tail_waves : (int)(blockDim.x >> 6);
  if ((int)(threadIdx.x >> 6) >= waves) return;
  const int groups_per_row = (p.N + 4 * waves - 1) / (4 * waves);
  const int row = tb / groups_per_row;
  const int n = ((tb - row * groups_per_row) * waves + (threadIdx.x >> 6)) * 4;
  const int m = p.tail_begin + row;
  if (n >= p.N || m >= p.M_total) return;
  constexpr bool AI = AIL && NPASS == 3, WI = NPASS == 3;
  const uint32_t arow = a_row_off<AMODE, AI>(p, m);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2                                   // two iterations' loads in flight: the leftover-row round is latency, not bandwidth
  for (int k0 = lane * 8; k0 < p.K; k0 += 512) {
    const uint32_t ao = arow + a_k_off<AMODE, AI>(p, k0 & ~63) + (AI ? ((k0 & 32) * 2 + (k0 & 31)) : (k0 & 63));
    const half8 ah = *(const half8*)(p.Ahi + ao);
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (float)ah[j];
    if (NPASS == 3) {
      const half8 al = *(const half8*)(p.Alo + ao);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += (float)al[j];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t wo = (size_t)(n + r) * p.ldw + (WI ? ((size_t)(k0 >> 5) * 64 + (k0 & 31)) : (size_t)k0);
      const half8 wh = *(const half8*)(p.Whi + wo);
      float w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = (float)wh[j];
      if (NPASS == 3) {
        const half8 wl = *(const half8*)(p.Wlo + wo);
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] += (float)wl[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[r] = fmaf(a[j], w[j], acc[r]);
    }
  }
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = wave_sum(acc[r]);
  if (lane == 0) {
    GemmParams q = p; q.M = p.M_total;
    epilogue4<EPI>(q, m, n, v);
  }
}

// waves per leftover-row workgroup (4 columns each) such that a row's columns spread over >= 128 workgroups where N allows
inline int gemm_tail_waves(int N, int max_waves) { int w = max_waves; while (w > 1 && (N + 4 * w - 1) / (4 * w) < 128) w >>= 1; return w; }
// Host launcher (defined in gemm.hip). npass in {1,3}.
int launch_gemm(const GemmParams& p, int npass, int epi, int amode, hipStream_t stream);
// false when an operand offset of this launch would not fit the kernels' 32-bit element offsets (launchers then return OVM_ERR_CAPACITY)
bool gemm_offsets_fit(const GemmParams& p, int npass, int amode);
// gemm256.hip: 256 x 256 tiles, two wave groups ping-ponging LOAD / COMPUTE; needs interleaved A and W images, N % 256 == 0
bool gemm256_supported(const GemmParams& p, int npass);
int launch_gemm256(const GemmParams& p, int epi, int ksplit_hint, hipStream_t stream);
// Tile-height override for tuning (0 = heuristic): 128 or 256.
void gemm_set_force_bm(int bm);
void gemm_set_tail_rows(int on);
void gemm_set_stages(int n);
void gemm_set_splitk(int v);
// gemm_small.hip: fp32-A latency-oriented kernel used by the generic linear op
bool gemm_small_supported(const float* A, int lda, int K);
int launch_gemm_small(const float* A, int lda, int M, int K, const half_t* Whi, const half_t* Wlo, int N, int Kpad, const float* bias, int act,
                      const float* R, int ldr, float* C, int ldc, int npass, hipStream_t s);
// same with an optional second A term (x = A + A2, same row stride; e.g. the position embedding added to attention queries / keys)
// and a caller-owned split-K workspace (null: the process-global one, not usable under HIP-graph capture)
int launch_gemm_small_ex(const float* A, const float* A2, int lda, int M, int K, const half_t* Whi, const half_t* Wlo, int N, int Kpad,
                         const float* bias, int act, const float* R, int ldr, float* C, int ldc, int npass, float* ws, size_t ws_bytes,
                         hipStream_t s);
void gemm_small_set(int target_blocks, int max_ksplit);
void gemm_small_set_stages(int n);
void gemm_small_set_wpe(int n);
void glinear_set_small_max_tiles(int t);
void gbmm_set_tiled(int v);

}  // namespace ovm
