// 256 x 256-tile split-precision GEMM for the large ViT contractions (qkv, fc1; fc2 / proj through split-K):
//   C[m][n] = sum_k A[m][k] W[n][k],  A and W as interleaved split-fp16 images [row][k/32][hi 32 | lo 32] (128-byte k-groups).
//
// Why another kernel. With 128 x 128 tiles every k-group of 32 moves 32 KiB into LDS for 768 MFMA cycles per SIMD: the LDS-DMA
// side (~45 GB/s per CU in-kernel) needs ~0.75 us for what the matrix cores do in ~0.35 us. A 256 x 256 tile moves
// 64 KiB for 3072 MFMA cycles per SIMD - half the bytes per MFMA - which balances the two sides.
//
// Structure (after the 8-phase / two-wave-group scheme of cdna_hip_programming.md 5 and MI355X_MICROARCH.md "Two waves per
// SIMD"): 8 waves = 2 (M) x 4 (N), each wave owns 128 x 64 outputs (128 accumulator registers) and walks them as four
// 64 x 32 quadrants per k-group, two quadrants per COMPUTE segment. A k-group is four 16-KiB half-tiles in LDS (A rows 0-127 / 128-255, W rows 0-127 / 128-255),
// two k-groups resident (128 KiB). Every wave alternates a LOAD segment (fragment ds_reads + 4 LDS-DMA pieces = 1/8 of two
// half-tiles) and a COMPUTE segment (48 MFMAs = two quadrants x three passes), separated by raw s_barriers. The M = 1 wave group
// runs one barrier behind the M = 0 group, and waves w and w + 4 share a SIMD, so on every SIMD one wave computes while the
// other loads. Half-tiles are re-staged as soon as their last reader is done (W: after LOAD 1, A: after LOAD 3), two to six
// segments ahead of their first use; one counted vmcnt per k-group keeps the younger pieces in flight across the barriers.
//
// Segment schedule of k-group t (G0 = waves 0-3, G1 = waves 4-7, one interval later): see OVM_TILE below. RAW: k-group t+1 is
// complete when its youngest pieces (A0/A1(t+1), issued in La of t) have landed; both groups wait for them before the barrier that
// precedes G0's La of t+1, leaving the two younger half-tiles W0/W1(t+2) in flight: vmcnt(4).
#include <hip/hip_runtime.h>
#include "gemm.hpp"

namespace ovm {

namespace {

constexpr int kHalf = 128 * 128;              // one half-tile: 128 rows x 128 bytes
// LDS map: region r in {A0, A1, W0, W1} of buffer b at (2 r + b) * kHalf - the two buffers of a region are 16 KiB apart, so
// with the k-loop unrolled by two the buffer select folds into the 16-bit immediate offset of ds_read_b128
constexpr int kRegion = 2 * kHalf;

__device__ __forceinline__ int frag_off(int row, int chunk) { return row * 128 + (chunk ^ ((row >> 1) & 7)) * 16; }

// STAMP = 1 is a diagnostic build (never used by the engine): every wave of workgroup 0 records s_memtime after each barrier of
// the first k-groups into the unused top 32 KiB of the LDS and dumps them to p.stamps at the end (cdna_hip_programming.md 7,
// "In-kernel stamps": read the SHARES of the segments, not the run time of this build).
// NTW = 16-column MFMA tiles per wave along N: 4 -> the 256 x 256 tile; 3 -> a 256 x 192 tile (W1 holds 64 rows). The qkv contraction
// of a ViT-L image (M = 4097, N = 3072) is 16 x 12 = 192 tiles of 256 x 256 - a quarter of the CUs idle for the whole launch - but
// 16 x 16 = 256 tiles of 256 x 192: one full round at 3/4 of the per-tile work.
template <int EPI, int STAMP = 0, int NTW = 4>
__global__ __launch_bounds__(512) void gemm256_kernel(const GemmParams p) {
  constexpr int BN = 64 * NTW;
  constexpr int WP = (NTW == 4) ? 4 : 3;      // LDS-DMA pieces a wave issues per k-group for the two W half-tiles
  if ((int)blockIdx.x >= p.main_tiles) { gemm_tail_body<3, EPI, A_ROWMAJOR, true>(p, (int)blockIdx.x - p.main_tiles); return; }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned long long t_entry = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int tiles_m = (p.M + 255) / 256;
  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  const int tiles = p.main_tiles / ksplit;
  const int tiles_n = tiles / tiles_m;
  const int ks = (int)blockIdx.x / tiles;
  // XCD-aware order: the workgroups that share an XCD (id % 8) take a compact patch of GROUP_M row tiles x all column tiles
  int pid = xcd_remap((int)blockIdx.x - ks * tiles, tiles);
  constexpr int GROUP_M = 4;
  const int in_group = GROUP_M * tiles_n;
  const int gid = pid / in_group;
  const int first_m = gid * GROUP_M;
  const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
  const int tm = first_m + (pid % in_group) % gsz;
  const int tn = (pid % in_group) / gsz;
  const int m0 = tm * 256, n0 = tn * BN;
  const int nk_all = p.K / 32;
  const int kt0 = (ksplit > 1) ? ks * p.kchunk : 0;
  const int nk = (ksplit > 1) ? ((nk_all - kt0 < p.kchunk) ? nk_all - kt0 : p.kchunk) : nk_all;

  // ---- staging: wave w moves pieces w and w + 8 (8 rows x 128 B each) of every half-tile
  uint32_t aoff[2][2], woff[2][2];            // [half][piece]
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = (wave + 8 * q) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      int m = m0 + hf * 128 + row; if (m > p.M - 1) m = p.M - 1;
      aoff[hf][q] = (uint32_t)m * (uint32_t)p.lda + chunk * 8;
      woff[hf][q] = (uint32_t)(n0 + hf * 128 + row) * (uint32_t)p.ldw + chunk * 8;
    }
  auto stage_a = [&](int t, int hf) {
    char* base = smem + hf * kRegion + (t & 1) * kHalf + wave * 1024;
    const uint32_t ko = (uint32_t)(kt0 + t) * 64;
    glds16(p.Ahi + aoff[hf][0] + ko, base);
    glds16(p.Ahi + aoff[hf][1] + ko, base + 8 * 1024);
  };
  auto stage_w = [&](int t, int hf) {
    char* base = smem + (2 + hf) * kRegion + (t & 1) * kHalf + wave * 1024;
    const uint32_t ko = (uint32_t)(kt0 + t) * 64;
    glds16(p.Whi + woff[hf][0] + ko, base);
    if (NTW == 4 || hf == 0) glds16(p.Whi + woff[hf][1] + ko, base + 8 * 1024);      // (NTW = 3: W1 is rows 128..191 only)
  };

  // ---- fragment addresses (16x16x32 operands: lane (fr, fq) holds k = 8 fq .. 8 fq + 7 of row fr; hi chunk fq, lo chunk 4 + fq)
  const int fr = lane & 15, fq = lane >> 4;
  int oa_hi[2][4], oa_lo[2][4], ow_hi[NTW], ow_lo[NTW];
#pragma unroll
  for (int ra = 0; ra < 2; ++ra)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = ra * 64 + i * 16 + fr;
      oa_hi[ra][i] = grp * kRegion + frag_off(row, fq);
      oa_lo[ra][i] = grp * kRegion + frag_off(row, 4 + fq);
    }
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int row = wc * 16 * NTW + i * 16 + fr;               // row of the W tile; rows 0-127 live in region W0, the rest in W1
    ow_hi[i] = (2 + (row >> 7)) * kRegion + frag_off(row & 127, fq);
    ow_lo[i] = (2 + (row >> 7)) * kRegion + frag_off(row & 127, 4 + fq);
  }

  f32x4 acc[2][4][NTW];                       // [row half][mi][ni]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[a][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  half8 ah[4], al[4], wh[NTW], wl[NTW];

#define OVM_READ_A(RA, BASE)                                                           \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                      \
    ah[i] = *(const half8*)((BASE) + oa_hi[RA][i]);                                    \
    al[i] = *(const half8*)((BASE) + oa_lo[RA][i]);                                    \
  }
#define OVM_READ_W(BASE)                                                               \
  _Pragma("unroll") for (int i = 0; i < NTW; ++i) {                                    \
    wh[i] = *(const half8*)((BASE) + ow_hi[i]);                                        \
    wl[i] = *(const half8*)((BASE) + ow_lo[i]);                                        \
  }
#define OVM_QUAD(RA, CB)                                                               \
  __builtin_amdgcn_s_setprio(1);                                                       \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                     \
  if ((CB) * 2 + ni < NTW)                                                             \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                   \
    f32x4 c_ = acc[RA][mi][(CB) * 2 + ni];                                             \
    c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[(CB) * 2 + ni], ah[mi], c_, 0, 0, 0); \
    c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(CB) * 2 + ni], al[mi], c_, 0, 0, 0); \
    c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(CB) * 2 + ni], ah[mi], c_, 0, 0, 0); \
    acc[RA][mi][(CB) * 2 + ni] = c_;                                                   \
  }                                                                                    \
  __builtin_amdgcn_s_setprio(0);
  int stamp_i = 0;
  unsigned long long* stamp_lds = (unsigned long long*)(smem + 8 * 64 * 68 * 4) + wave * 128;
#define OVM_STAMP() do { if (STAMP && blockIdx.x == 0 && stamp_i < 124) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    if (lane == 0) stamp_lds[stamp_i] = t_; ++stamp_i; } } while (0)
#define OVM_BAR() do { __builtin_amdgcn_sched_barrier(0); OVM_STAMP(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); OVM_STAMP(); } while (0)

  // ---- prologue: k-group 0 complete, W0 / W1 of k-group 1
  stage_a(0, 0); stage_a(0, 1); stage_w(0, 0); stage_w(0, 1);
  if (nk > 1) { stage_w(1, 0); stage_w(1, 1); wait_vmcnt<WP>(); } else { wait_vmcnt<0>(); }
  OVM_BAR();
  if (grp == 1) OVM_BAR();                    // G1 runs one interval behind G0

// Four segments per k-group (two quadrants = 48 MFMAs per COMPUTE segment): half as many barriers per MFMA as one quadrant per
// segment (stamps: a barrier costs ~140 dead cycles to the last arriver; 24-MFMA segments ran at 56 % of the MFMA issue rate).
//   La: read A(rows 0-63) + all W of t | stage A0(t+1), A1(t+1)         Ca: quadrants (0,0) (0,1)
//   Lb: read A(rows 64-127)            | stage W0(t+2), W1(t+2), wait   Cb: quadrants (1,1) (1,0)   (G0 waits at the end of Cb)
// LOAD segments end with lgkmcnt(0) (they idle at the barrier anyway), so a region is free once the barrier after its last read
// has passed: W(t) after interval 1 (staged in 2, 3), A0(t) after 2, A1(t) after 3 (staged in intervals 4, 5 = La of t+1).
#define OVM_TILE(T_, B_)                                                               \
  {                                                                                    \
    const int t = (T_);                                                                \
    const char* base = smem + (B_) * kHalf;                                            \
    OVM_READ_A(0, base)                       /* La */                                 \
    OVM_READ_W(base)                                                                   \
    if (t + 1 < nk) { stage_a(t + 1, 0); stage_a(t + 1, 1); }                          \
    __builtin_amdgcn_s_waitcnt(0xC07F);       /* lgkmcnt(0) */                         \
    OVM_BAR();                                                                         \
    OVM_QUAD(0, 0)                            /* Ca */                                 \
    OVM_QUAD(0, 1)                                                                     \
    OVM_BAR();                                                                         \
    OVM_READ_A(1, base)                       /* Lb */                                 \
    if (t + 2 < nk) { stage_w(t + 2, 0); stage_w(t + 2, 1); }                          \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                \
    if (grp == 1) { if (t + 2 < nk) wait_vmcnt<WP>(); else wait_vmcnt<0>(); }           \
    OVM_BAR();                                                                         \
    OVM_QUAD(1, 1)                            /* Cb */                                 \
    OVM_QUAD(1, 0)                                                                     \
    if (grp == 0) { if (t + 2 < nk) wait_vmcnt<WP>(); else wait_vmcnt<0>(); }           \
    OVM_BAR();                                                                         \
  }
  int t2 = 0;
  for (; t2 + 1 < nk; t2 += 2) {
    OVM_TILE(t2, 0)
    OVM_TILE(t2 + 1, 1)
  }
  if (t2 < nk) OVM_TILE(t2, 0)
#undef OVM_TILE
  if (grp == 0) OVM_BAR();                    // same number of barriers for every wave
#undef OVM_READ_A
#undef OVM_READ_W
#undef OVM_QUAD

  const unsigned long long t_loop_end = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
  // ---- epilogue through LDS. The MFMA layout gives a lane 4 columns of 16 different rows: stored directly, every store
  // instruction touches 16 rows x 64 B, and with all 256 workgroups of a one-round grid reaching the epilogue together that
  // cost 30 % of the kernel (stamps: 57k of 192k cycles at the fc1 shape). Each wave stages 64 x 64 accumulators in the drained
  // ring and re-reads them row-major, so that 16 lanes cover one row: 256 contiguous bytes of fp32 / of the interleaved split
  // image [hi 32 | lo 32 | hi 32 | lo 32] per row and instruction.
  constexpr int TLD = 68;                                      // padded row stride (floats)
  float* tile = (float*)smem + wave * (64 * TLD);
  const int mb = m0 + grp * 128, nb = n0 + wc * 16 * NTW;
  // V^T destination per 16-column group (wave-uniform: head blocks are 64 wide and the Q | K | V boundaries multiples of 64). With
  // NTW = 4 a wave's 64 columns are one head, so all four groups agree; with NTW = 3 a wave can straddle the K | V boundary.
  bool vt[NTW];
#pragma unroll
  for (int ni = 0; ni < NTW; ++ni) vt[ni] = (EPI == EPI_QKV && ksplit == 1) ? ((nb + ni * 16) / (p.N / 3)) == 2 : false;
  bool any_rm = false, any_vt = false;
#pragma unroll
  for (int ni = 0; ni < NTW; ++ni) { any_rm = any_rm || !vt[ni]; any_vt = any_vt || vt[ni]; }
#pragma unroll
  for (int ra = 0; ra < 2; ++ra) {
    const int mrow0 = mb + ra * 64;
    // ---- phase 1: the row-major groups (everything but V^T) share the staging tile [token row][column]
    if (any_rm) {
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni)
        if (!vt[ni]) {
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) *(f32x4*)(tile + (mi * 16 + fr) * TLD + ni * 16 + fq * 4) = acc[ra][mi][ni];
        }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int col = (lane & 15) * 4;
      bool mine = false;
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni) mine = mine || ((col >> 4) == ni && !vt[ni]);
      if (mine) {
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
          const int row = it * 4 + (lane >> 4);
          const f32x4 v = *(const f32x4*)(tile + row * TLD + col);
          const int m = mrow0 + row, n = nb + col;
          if (m < p.M) {
            if (ksplit > 1) { if (n < p.N) *(f32x4*)(p.part + ((size_t)ks * p.M + m) * p.N + n) = v; }
            else epilogue4<EPI>(p, m, n, v);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    // ---- phase 2: V^T groups, staged TRANSPOSED ([d][token]) in the same tile (a wave that straddles the K | V boundary holds both
    // kinds, and the two images overlap - hence two phases). The store loop reads a token per lane for one d at a time, which on a
    // row-major image was 64 lanes at a stride of 68 floats = 8 lanes per bank (PMC r02: LDS conflict cycles 0.335 of the LDS
    // instructions in the qkv GEMM against 0.13 in fc1, same main loop); transposed it is 64 consecutive floats.
    if (EPI == EPI_QKV && any_vt) {
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni)
        if (vt[ni]) {
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[(ni * 16 + fq * 4 + e) * TLD + mi * 16 + fr] = acc[ra][mi][ni][e];
        }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // V^T [b][head][d][Tpad]: tokens are the contiguous axis, so lanes run along m (one 2-byte element each, 128 B per store)
      const int m = mrow0 + lane;
      const int Dm = p.N / 3;
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni)
        if (vt[ni] && m < p.M) {
          const int f0 = nb + ni * 16 - 2 * Dm;                       // feature index of the group's first column inside V
          const int head = f0 >> 6, d0 = f0 & 63;
          const int b = m / p.T, t = m - b * p.T;
          const int tp = (t & ~15) | (t & 3) | ((t & 4) << 1) | ((t & 8) >> 1);
          const size_t o0 = ((size_t)(b * p.heads + head) * 64 + d0) * p.Tpad + tp;
          for (int d = 0; d < 16; ++d) {
            float x = tile[(ni * 16 + d) * TLD + lane];
            if (p.bias) x += p.bias[nb + ni * 16 + d];
            half_t hh, ll; split_f16(x, hh, ll);
            p.Vhi[o0 + (size_t)d * p.Tpad] = hh;
            if (p.Vlo) p.Vlo[o0 + (size_t)d * p.Tpad] = ll;
          }
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
#undef OVM_BAR
  if (STAMP && blockIdx.x == 0 && p.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_exit = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (lane == 0) {
      for (int i = 0; i < 124; ++i) p.stamps[wave * 128 + i] = (i < stamp_i) ? stamp_lds[i] : 0ull;
      p.stamps[wave * 128 + 124] = (unsigned long long)stamp_i;
      p.stamps[wave * 128 + 125] = t_entry; p.stamps[wave * 128 + 126] = t_loop_end; p.stamps[wave * 128 + 127] = t_exit;
    }
  }
}

float* g_ws256 = nullptr; size_t g_ws256_cap = 0;
int g_gemm256_n192 = 1;              // ovm_tune_set("gemm256_n192", 0): qkv on 256 x 256 tiles as in round 2

template <int EPI>
int launch256(const GemmParams& p, int want_split, hipStream_t s) {
  constexpr int smem = 8 * 64 * 68 * 4;                    // = 139,264 B: the epilogue's eight padded 64 x 64 staging tiles (the ring needs 131,072)
  static_assert(smem >= 8 * kHalf, "ring");
  if (EPI == EPI_STORE && p.stamps) {                     // diagnostic build
    GemmParams q = p;
    q.M_total = p.M; q.tail_begin = p.M; q.ldw = 2 * p.K; q.ksplit = 1; q.kchunk = 0; q.part = nullptr;
    q.main_tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    (void)hipFuncSetAttribute((const void*)gemm256_kernel<EPI_STORE, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem + 8192);
    hipLaunchKernelGGL((gemm256_kernel<EPI_STORE, 1>), dim3(q.main_tiles), dim3(512), smem + 8192, s, q);
    return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
  }
  GemmParams q = p;
  q.M_total = p.M; q.tail_begin = p.M; q.ldw = 2 * p.K;
  q.ksplit = 1; q.kchunk = 0; q.part = nullptr;
  int tail_blocks = 0;
  const int tail = p.M % 256;
  if (tail > 0 && tail <= 8 && p.M > 256) {             // leftover rows (the cls token): dot-product workgroups of the same launch
    q.tail_begin = p.M - tail; q.M = q.tail_begin;
    q.tail_waves = gemm_tail_waves(p.N, 8);
    tail_blocks = tail * ((p.N + 4 * q.tail_waves - 1) / (4 * q.tail_waves));
  }
  const int tiles_m = (q.M + 255) / 256, tiles_n = (p.N + 255) / 256;
  const int nk = p.K / 32;
  int ks = want_split;
  if (ks > 1 && p.N % 4 == 0) {                           // (leftover rows keep their dot-product workgroups: those run the whole K and store final values)
    if (ks > nk / 8) ks = nk / 8;
    if (ks > 1) {
      const int chunk = (nk + ks - 1) / ks;
      ks = (nk + chunk - 1) / chunk;
      const size_t need = (size_t)ks * p.M * p.N * sizeof(float);
      float* ws = p.part_ws; size_t cap = p.part_cap;
      if (!ws) {
        if (g_ws256_cap < need) {
          if (g_ws256) { (void)hipDeviceSynchronize(); (void)hipFree(g_ws256); }
          g_ws256_cap = need;
          if (hipMalloc((void**)&g_ws256, need) != hipSuccess) { g_ws256 = nullptr; g_ws256_cap = 0; return OVM_ERR_HIP; }
        }
        ws = g_ws256; cap = g_ws256_cap;
      }
      if (cap < need) return OVM_ERR_CAPACITY;
      q.ksplit = ks; q.kchunk = chunk; q.part = ws;
    }
  }
  q.main_tiles = tiles_m * tiles_n * q.ksplit;
  // 256 x 192 tiles where they cost fewer rounds x tile size than 256 x 256 (ViT-L qkv at batch 1: 256 tiles x 3/4 against 192 x 1)
  if (EPI == EPI_QKV && q.ksplit == 1 && g_gemm256_n192 && p.N % 192 == 0) {
    const long t4 = (long)tiles_m * tiles_n, t3 = (long)tiles_m * (p.N / 192);
    if (((t3 + 255) / 256) * 3 < ((t4 + 255) / 256) * 4) {
      static bool attr3 = false;
      if (!attr3) {
        if (hipFuncSetAttribute((const void*)gemm256_kernel<EPI, 0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return OVM_ERR_HIP;
        attr3 = true;
      }
      q.main_tiles = (int)t3;
      hipLaunchKernelGGL((gemm256_kernel<EPI, 0, 3>), dim3(q.main_tiles + tail_blocks), dim3(512), smem, s, q);
      return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
    }
  }
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm256_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return OVM_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm256_kernel<EPI>), dim3(q.main_tiles + tail_blocks), dim3(512), smem, s, q);
  if (q.ksplit > 1) {
    const long n = (long)p.M * (p.N / 4);
    hipLaunchKernelGGL((splitk_epilogue_kernel<EPI>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, q);
  }
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

}  // namespace

void gemm256_set_n192(int v) { g_gemm256_n192 = v; }

bool gemm256_supported(const GemmParams& p, int npass) {
  return npass == 3 && p.a_il && p.Alo == p.Ahi + 32 && p.Wlo == p.Whi + 32 && p.K % 32 == 0 && p.lda == 2 * p.K && p.N % 256 == 0;
}

// ksplit_hint <= 1: no split-K
int launch_gemm256(const GemmParams& p, int epi, int ksplit_hint, hipStream_t s) {
  if (!gemm256_supported(p, 3)) return OVM_ERR_INVALID;
  if (p.M <= 0 || p.N <= 0) return OVM_OK;
  if (!gemm_offsets_fit(p, 3, A_ROWMAJOR)) return OVM_ERR_CAPACITY;
  switch (epi) {
    case EPI_STORE: return launch256<EPI_STORE>(p, ksplit_hint, s);
    case EPI_RESID: return launch256<EPI_RESID>(p, ksplit_hint, s);
    case EPI_GELU:  return launch256<EPI_GELU>(p, ksplit_hint, s);
    case EPI_QKV:   return launch256<EPI_QKV>(p, ksplit_hint, s);
  }
  return OVM_ERR_INVALID;
}

}  // namespace ovm
