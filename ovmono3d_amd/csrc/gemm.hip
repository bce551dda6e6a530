// Instantiations + host dispatch of the MFMA GEMM (see gemm.hpp).
#include "gemm.hpp"

namespace ovm {

static int g_force_bm = 0;
static int g_tail_rows = 1;
static int g_force_stages = 0;
static int g_splitk = 1;
void gemm_set_splitk(int v) { g_splitk = v; }

void gemm_set_stages(int n) { g_force_stages = n; }
void gemm_set_force_bm(int bm) { g_force_bm = bm; }
void gemm_set_tail_rows(int on) { g_tail_rows = on; }

template <int NPASS, int BK, int BM, int NSTAGE, int EPI, int AMODE, bool AIL = false>
static int launch_one(const GemmParams& p, hipStream_t s) {
  constexpr int smem = NSTAGE * (BM + 128) * BK * 2 * ((NPASS == 3) ? 2 : 1);
  if (p.K % BK != 0) return OVM_ERR_SHAPE;
  GemmParams q = p;
  q.M_total = p.M; q.tail_begin = p.M; q.main_tiles = 0;
  q.ldw = (NPASS == 3) ? 2 * p.K : p.K;
  int tail_blocks = 0;
  const int tail = p.M % 128;
  if (g_tail_rows && tail > 0 && tail <= 8 && p.M > 128 && p.K % 64 == 0) {
    // leftover rows (the cls token of the 4097-token canvas) ride along as extra dot-product workgroups
    q.tail_begin = p.M - tail;
    q.M = q.tail_begin;
    const int waves = gemm_tail_waves(p.N, BM / 32);
    q.tail_waves = waves;
    tail_blocks = tail * ((p.N + 4 * waves - 1) / (4 * waves));
  }
  const int tiles_m = (q.M + BM - 1) / BM;
  const int tiles_n = (p.N + 127) / 128;
  q.main_tiles = tiles_m * tiles_n;
  static bool attr_set = false;
  if (!attr_set) {
    if (smem > 65536 &&
        hipFuncSetAttribute((const void*)gemm_kernel<NPASS, BK, BM, NSTAGE, EPI, AMODE, AIL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return OVM_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_kernel<NPASS, BK, BM, NSTAGE, EPI, AMODE, AIL>), dim3(q.main_tiles + tail_blocks), dim3(BM * 2), smem, s, q);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

static float* g_splitk_ws_[2] = {nullptr, nullptr}; static size_t g_splitk_cap_[2] = {0, 0};

template <int NPASS, int BK, int NS, int EPI, int AMODE, bool AIL = false>
static int launch_ws(const GemmParams& p, hipStream_t s) {
  constexpr int smem = NS * (128 + 128) * BK * 2 * ((NPASS == 3) ? 2 : 1);
  if (p.K % BK != 0) return OVM_ERR_SHAPE;
  GemmParams q = p;
  q.M_total = p.M; q.tail_begin = p.M; q.main_tiles = 0;
  q.ldw = (NPASS == 3) ? 2 * p.K : p.K;
  q.ksplit = 1; q.kchunk = 0; q.part = nullptr;
  const int tiles_full = ((p.M + 127) / 128) * ((p.N + 127) / 128);
  const int nk = p.K / BK;
  // a thin grid with a long reduction (the RoI heads' fc1: <= 64 tiles, 392 k-tiles) leaves most CUs idle for hundreds of
  // k-steps: split K over workgroups (deterministic: partial tiles + one reduce/epilogue pass)
  // (also the GroundingDINO engine's Swin proj / fc2 GEMMs: 40-90 tiles with 16-64 k-tiles, residual epilogue: the reduce
  // pass applies any epilogue, EPI_QKV's V^T scatter excepted)
  const int min_nk = g_splitk >= 2 ? 16 : 32, min_steps = g_splitk >= 2 ? 4 : 8;     // experiment (gemm_splitk 2): slices of >= 4 k-steps from 16 k-steps on
  if (g_splitk && (EPI != EPI_QKV && EPI != EPI_PATCH && EPI != EPI_CONVT) && tiles_full <= 96 && nk >= min_nk && p.N % 4 == 0) {
    int ks = 256 / tiles_full; if (ks > 16) ks = 16; if (ks > nk / min_steps) ks = nk / min_steps;
    if (ks > 1) {
      const int chunk = (nk + ks - 1) / ks;
      ks = (nk + chunk - 1) / chunk;
      const size_t need = (size_t)ks * p.M * p.N * sizeof(float);
      if (p.part_ws) {                                         // caller-owned (graph-safe) workspace
        if (p.part_cap < need) return OVM_ERR_CAPACITY;
        q.ksplit = ks; q.kchunk = chunk; q.part = p.part_ws;
      } else {
        float*& g_splitk_ws = g_splitk_ws_[p.ws_slot & 1]; size_t& g_splitk_cap = g_splitk_cap_[p.ws_slot & 1];
        if (g_splitk_cap < need) {
          if (g_splitk_ws) { (void)hipDeviceSynchronize(); (void)hipFree(g_splitk_ws); }
          g_splitk_cap = need + need / 2 + (1 << 20);
          if (hipMalloc((void**)&g_splitk_ws, g_splitk_cap) != hipSuccess) { g_splitk_ws = nullptr; g_splitk_cap = 0; return OVM_ERR_HIP; }
        }
        q.ksplit = ks; q.kchunk = chunk; q.part = g_splitk_ws;
      }
    }
  }
  int tail_blocks = 0;
  const int tail = p.M % 128;
  if (q.ksplit == 1 && g_tail_rows && tail > 0 && tail <= 8 && p.M > 128 && p.K % 64 == 0) {
    q.tail_begin = p.M - tail;
    q.M = q.tail_begin;
    q.tail_waves = gemm_tail_waves(p.N, 8);                    // of the 8 waves per workgroup; 4 columns per wave
    tail_blocks = tail * ((p.N + 4 * q.tail_waves - 1) / (4 * q.tail_waves));
  }
  const int tiles_m = (q.M + 127) / 128;
  const int tiles_n = (p.N + 127) / 128;
  q.main_tiles = tiles_m * tiles_n * q.ksplit;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_ws_kernel<NPASS, BK, NS, EPI, AMODE, AIL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return OVM_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_ws_kernel<NPASS, BK, NS, EPI, AMODE, AIL>), dim3(q.main_tiles + tail_blocks), dim3(512), smem, s, q);
  if (q.ksplit > 1) {
    const long n = (long)p.M * (p.N / 4);
    hipLaunchKernelGGL((splitk_epilogue_kernel<EPI>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, q);
  }
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

// Tile height. Measured on MI355X at the ViT-L shapes (M = 4097; N, K in {1024, 3072, 4096}; profiles/r01):
// 128 rows with two LDS stages (2-3 workgroups per CU) beats 256 rows and the 3-stage ring everywhere, so that
// is the default; the other variants stay selectable for tuning (ovm_tune_set "gemm_bm" / "gemm_stages").
static int pick_bm(const GemmParams& p) {
  (void)p;
  return g_force_bm == 256 ? 256 : 128;
}

template <int EPI, int AMODE>
static int launch_prec(const GemmParams& p, int npass, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0) return OVM_OK;
  const int bm = pick_bm(p);
  // Default: up to two rounds of tiles (<= 512 on 256 CUs) run on the wave-specialised kernel (one workgroup per CU, no DMA
  // issue or global-memory waits in the MFMA waves: +11..+16 % at the ViT-L proj / fc2 shapes); larger grids keep the symmetric
  // two-workgroups-per-CU kernel, whose co-resident workgroup hides the per-tile prologue and epilogue (qkv with its V^T
  // scatter: 102 vs 110 us, fc1: 122 vs 135 us in favour of the symmetric kernel).
  const int tail_rows = p.M % 128;                    // <= 8 leftover rows ride along as dot-product workgroups, not as a row of tiles
  const long tiles_m = (g_tail_rows && tail_rows > 0 && tail_rows <= 8 && p.M > 128 && p.K % 64 == 0) ? p.M / 128 : (p.M + 127) / 128;
  const long tiles = tiles_m * ((p.N + 127) / 128);
  const int st = g_force_stages ? g_force_stages : (tiles <= 512 ? 6 : 2);
  if (npass == 3 && p.a_il) {                      // interleaved activations: the two default variants only
    if (st == 6 || st == 5) return launch_ws<3, 32, 3, EPI, AMODE, true>(p, s);
    return launch_one<3, 32, 128, 2, EPI, AMODE, true>(p, s);
  }
  if (st == 5) return npass == 3 ? launch_ws<3, 32, 4, EPI, AMODE>(p, s) : launch_ws<1, 64, 4, EPI, AMODE>(p, s);
  if (st == 6) return npass == 3 ? launch_ws<3, 32, 3, EPI, AMODE>(p, s) : launch_ws<1, 64, 3, EPI, AMODE>(p, s);
  if (npass == 3) {
    if (bm == 256) return st == 3 ? launch_one<3, 32, 256, 3, EPI, AMODE>(p, s) : launch_one<3, 32, 256, 2, EPI, AMODE>(p, s);
    return st == 3 ? launch_one<3, 32, 128, 3, EPI, AMODE>(p, s) : launch_one<3, 32, 128, 2, EPI, AMODE>(p, s);
  }
  if (bm == 256) return st == 3 ? launch_one<1, 64, 256, 3, EPI, AMODE>(p, s) : launch_one<1, 64, 256, 2, EPI, AMODE>(p, s);
  return st == 3 ? launch_one<1, 64, 128, 3, EPI, AMODE>(p, s) : launch_one<1, 64, 128, 2, EPI, AMODE>(p, s);
}

// The kernels address their operands with 32-bit ELEMENT offsets (gemm.hpp a_row_offset / woff, gemm256.hip aoff / woff:
// uint32 counts of halves; the implicit-GEMM row offset is formed in int): the last element a launch can touch must stay
// below 2^32 (2^31 for the convolution image). ViT-L, T = 5477, batch 64, interleaved fc2 input (lda 8192) = 2.87e9: inside;
// the next size up is refused here instead of wrapping around silently.
bool gemm_offsets_fit(const GemmParams& p, int npass, int amode) {
  const uint64_t lim = 1ull << 32;
  const uint64_t ldw = (npass == 3) ? 2ull * p.K : (uint64_t)p.K;
  const uint64_t npad = ((uint64_t)p.N + 255) / 256 * 256;           // weights are padded to whole tiles of either kernel
  if (npad * ldw > lim) return false;
  if (amode == A_CONV3X3) {
    if (p.cH <= 0 || p.cW <= 0 || p.cC <= 0) return false;
    const uint64_t Bn = ((uint64_t)p.M + (uint64_t)p.cH * p.cW - 1) / ((uint64_t)p.cH * p.cW);
    return Bn * (p.cH + 2) * (p.cW + 2) * (uint64_t)p.cC <= (1ull << 31);
  }
  const uint64_t rowlen = (npass == 3 && p.a_il) ? 2ull * p.K : (uint64_t)p.K;
  return (uint64_t)(p.M > 0 ? p.M - 1 : 0) * (uint64_t)p.lda + rowlen <= lim;
}

int launch_gemm(const GemmParams& p, int npass, int epi, int amode, hipStream_t s) {
  if (npass == 3 && (p.Alo == nullptr || p.Wlo == nullptr)) return OVM_ERR_INVALID;
  if (!gemm_offsets_fit(p, npass, amode)) return OVM_ERR_CAPACITY;
  if (amode == A_CONV3X3) {
    if (epi != EPI_STORE) return OVM_ERR_INVALID;
    return launch_prec<EPI_STORE, A_CONV3X3>(p, npass, s);
  }
  switch (epi) {
    case EPI_STORE: return launch_prec<EPI_STORE, A_ROWMAJOR>(p, npass, s);
    case EPI_RESID: return launch_prec<EPI_RESID, A_ROWMAJOR>(p, npass, s);
    case EPI_GELU:  return launch_prec<EPI_GELU, A_ROWMAJOR>(p, npass, s);
    case EPI_QKV:   return launch_prec<EPI_QKV, A_ROWMAJOR>(p, npass, s);
    case EPI_PATCH: return launch_prec<EPI_PATCH, A_ROWMAJOR>(p, npass, s);
    case EPI_CONVT: return launch_prec<EPI_CONVT, A_ROWMAJOR>(p, npass, s);
  }
  return OVM_ERR_INVALID;
}

}  // namespace ovm
