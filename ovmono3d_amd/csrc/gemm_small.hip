// Latency-oriented MFMA GEMM for the GroundingDINO branch:  y[M][N] = act(x[M][K] W^T + bias) (+ r)
//
// The branch issues hundreds of small projections (16 text tokens, 900 queries, ~6k encoder tokens, Swin stages of
// 1-20k tokens with 128-1024 channels). On those the 128x128 LDS-DMA kernel of gemm.hpp is bound by its per-k-step
// latency on a mostly empty chip, and every call needed a separate fp32 -> split-fp16 pass over x. This kernel
//   * reads fp32 x directly and splits it to hi/lo fp16 in registers on the way to LDS (no pre-pass, no scratch);
//   * uses 64x64 tiles (4 waves, 32x32 each) so small problems still spread over the CUs, with k-steps of 64 staged
//     global -> VGPR -> LDS (plain loads are cheap to issue; the next k-step's loads fly during the MFMAs);
//   * splits K over workgroups when the tile grid alone cannot fill the chip (partials to a workspace, then a
//     reduce + epilogue kernel; deterministic - no atomics).
// Same numerics contract as gemm.hpp: NPASS = 3 accumulates Al*Wh + Ah*Wl + Ah*Wh into one fp32 accumulator.
#include <hip/hip_runtime.h>
#include "gemm.hpp"
#include "kernels.hpp"

namespace ovm {

struct SmallGemmParams {
  const float* A; int lda;
  const float* A2;                        // optional second term of the A operand (x = A + A2), same layout
  const half_t* Whi; const half_t* Wlo;   // [Npad][Kpad], Npad % 128 == 0, Kpad % 64 == 0, zero padded
  int M, N, K, Kpad;
  const float* bias; int act; const float* R; int ldr; float* C; int ldc;
  int tiles_m, tiles_n, ksplit, kchunk;
  float* partial; int ldp;                // [ksplit][M][ldp] when ksplit > 1
  int vec_ok;                             // C / R rows 16-byte aligned and N % 4 == 0
};

__device__ __forceinline__ float apply_act(float x, int act) {
  if (act == 1) return fmaxf(x, 0.f);
  if (act == 2) return gelu_erf(x);
  return x;
}

template <int NPASS>
__device__ __forceinline__ void cvt8(const float4 a, const float4 b, half8& h, half8& l) {
  half_t h0, h1, h2, h3, h4, h5, h6, h7, l0 = 0, l1 = 0, l2 = 0, l3 = 0, l4 = 0, l5 = 0, l6 = 0, l7 = 0;
  if (NPASS == 3) {
    split_f16_nt(a.x, h0, l0); split_f16_nt(a.y, h1, l1); split_f16_nt(a.z, h2, l2); split_f16_nt(a.w, h3, l3);
    split_f16_nt(b.x, h4, l4); split_f16_nt(b.y, h5, l5); split_f16_nt(b.z, h6, l6); split_f16_nt(b.w, h7, l7);
  } else {
    h0 = cvt_f16_rn_nt(a.x); h1 = cvt_f16_rn_nt(a.y); h2 = cvt_f16_rn_nt(a.z); h3 = cvt_f16_rn_nt(a.w);
    h4 = cvt_f16_rn_nt(b.x); h5 = cvt_f16_rn_nt(b.y); h6 = cvt_f16_rn_nt(b.z); h7 = cvt_f16_rn_nt(b.w);
  }
  h = (half8){h0, h1, h2, h3, h4, h5, h6, h7};
  l = (half8){l0, l1, l2, l3, l4, l5, l6, l7};
}

// WPE = waves per SIMD the register allocation is held to: 4 (<= 128 VGPRs) lets a workgroup take the slots the ViT's attention
// workgroups leave free on a CU (128 VGPRs per SIMD, 64 KB of LDS) instead of waiting for one of them to retire.
template <int NPASS, int NSTAGE, int WPE = 1>
__global__ __launch_bounds__(256, WPE) void gemm_f32a_kernel(const SmallGemmParams p) {
  constexpr int PARTS = (NPASS == 3) ? 4 : 2;
  constexpr int PART = 64 * 128;                                    // 64 rows x 64 halves
  constexpr int P_AH = 0, P_AL = PART, P_WH = (NPASS == 3 ? 2 : 1) * PART, P_WL = 3 * PART;
  __shared__ __attribute__((aligned(16))) char smem[NSTAGE * PARTS * PART];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles = p.tiles_m * p.tiles_n;
  int pid = blockIdx.x;
  const int ks = pid / tiles;
  pid -= ks * tiles;
  const int tm = pid % p.tiles_m, tn = pid / p.tiles_m;
  const int m0 = tm * 64, n0 = tn * 64;
  const int kbeg = ks * p.kchunk;
  const int kend = (kbeg + p.kchunk < p.Kpad) ? kbeg + p.kchunk : p.Kpad;
  const int nk = (kend - kbeg) / 64;

  // staging: thread (row = tid/4, q = tid%4) moves 16 consecutive k of one A row and of one W row
  const int row = tid >> 2, q = tid & 3;
  int am = m0 + row; if (am > p.M - 1) am = p.M - 1;
  const float* ap = p.A + (size_t)am * p.lda + q * 16;
  const float* ap2 = p.A2 ? p.A2 + (size_t)am * p.lda + q * 16 : nullptr;
  // weights: one-pass [Npad][Kpad]; split mode the interleaved image [Npad][Kpad/32][hi 32 | lo 32] (k0 is a multiple of 64, q*16 of
  // 16: the 16 halves a thread moves never straddle a 32-group)
  const half_t* whp = (NPASS == 3) ? p.Whi + (size_t)(n0 + row) * 2 * p.Kpad + (q >> 1) * 64 + (q & 1) * 16
                                   : p.Whi + (size_t)(n0 + row) * p.Kpad + q * 16;
  const half_t* wlp = whp + 32;
  const int wkm = (NPASS == 3) ? 2 : 1;                     // k0 -> offset multiplier inside a row
  struct Stage { float4 a0, a1, a2, a3; uint4 wh0, wh1, wl0, wl1; };
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto gload = [&](Stage& r, int k0) {
    const int kq = k0 + q * 16;
    r.a0 = (kq < p.K) ? *(const float4*)(ap + k0) : z4;
    r.a1 = (kq + 4 < p.K) ? *(const float4*)(ap + k0 + 4) : z4;
    r.a2 = (kq + 8 < p.K) ? *(const float4*)(ap + k0 + 8) : z4;
    r.a3 = (kq + 12 < p.K) ? *(const float4*)(ap + k0 + 12) : z4;
    if (ap2) {                                                // wave-uniform
      auto add4 = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
      if (kq < p.K) add4(r.a0, *(const float4*)(ap2 + k0));
      if (kq + 4 < p.K) add4(r.a1, *(const float4*)(ap2 + k0 + 4));
      if (kq + 8 < p.K) add4(r.a2, *(const float4*)(ap2 + k0 + 8));
      if (kq + 12 < p.K) add4(r.a3, *(const float4*)(ap2 + k0 + 12));
    }
    r.wh0 = *(const uint4*)(whp + k0 * wkm); r.wh1 = *(const uint4*)(whp + k0 * wkm + 8);
    if (NPASS == 3) { r.wl0 = *(const uint4*)(wlp + k0 * wkm); r.wl1 = *(const uint4*)(wlp + k0 * wkm + 8); }
  };
  const int woff0 = row * 128 + swz_slot<64>(row, 2 * q) * 16;
  const int woff1 = row * 128 + swz_slot<64>(row, 2 * q + 1) * 16;
  auto lwrite = [&](const Stage& r, int buf) {
    char* base = smem + buf * PARTS * PART;
    half8 h, l;
    cvt8<NPASS>(r.a0, r.a1, h, l);
    *(half8*)(base + P_AH + woff0) = h;
    if (NPASS == 3) *(half8*)(base + P_AL + woff0) = l;
    cvt8<NPASS>(r.a2, r.a3, h, l);
    *(half8*)(base + P_AH + woff1) = h;
    if (NPASS == 3) *(half8*)(base + P_AL + woff1) = l;
    *(uint4*)(base + P_WH + woff0) = r.wh0; *(uint4*)(base + P_WH + woff1) = r.wh1;
    if (NPASS == 3) { *(uint4*)(base + P_WL + woff0) = r.wl0; *(uint4*)(base + P_WL + woff1) = r.wl1; }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const char* base) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      half8 ah[2], wh[2], al[2], wl[2];
      const int chunk = kk * 4 + fq;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra_ = wm * 32 + i * 16 + fr, rw_ = wn * 32 + i * 16 + fr;
        const int oa = ra_ * 128 + swz_slot<64>(ra_, chunk) * 16;
        const int ow = rw_ * 128 + swz_slot<64>(rw_, chunk) * 16;
        ah[i] = *(const half8*)(base + P_AH + oa);
        wh[i] = *(const half8*)(base + P_WH + ow);
        if (NPASS == 3) { al[i] = *(const half8*)(base + P_AL + oa); wl[i] = *(const half8*)(base + P_WL + ow); }
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          if (NPASS == 3) {
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ni], ah[mi], acc[ni][mi], 0, 0, 0);
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ni], al[mi], acc[ni][mi], 0, 0, 0);
          }
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ni], ah[mi], acc[ni][mi], 0, 0, 0);
        }
    }
  };

  if (NSTAGE == 2) {
    // double-buffered LDS, one k-step of loads in flight
    if (nk > 0) {
      Stage r;
      gload(r, kbeg);
      lwrite(r, 0);
      __syncthreads();
      int cur = 0;
      for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload(r, kbeg + (kt + 1) * 64);
        compute(smem + cur * PARTS * PART);
        if (more) lwrite(r, cur ^ 1);
        __syncthreads();
        cur ^= 1;
      }
    }
  } else {
    // single LDS buffer (32 KB: several workgroups per CU), two register sets = two k-steps of loads in flight
    Stage r0, r1;
    if (nk > 0) gload(r0, kbeg);
    if (nk > 1) gload(r1, kbeg + 64);
    for (int kt = 0; kt < nk; kt += 2) {
      lwrite(r0, 0);
      __syncthreads();
      if (kt + 2 < nk) gload(r0, kbeg + (kt + 2) * 64);
      compute(smem);
      __syncthreads();
      if (kt + 1 < nk) {
        lwrite(r1, 0);
        __syncthreads();
        if (kt + 3 < nk) gload(r1, kbeg + (kt + 3) * 64);
        compute(smem);
        __syncthreads();
      }
    }
  }

  // lane holds m = m0 + wm*32 + mi*16 + fr, n = n0 + wn*32 + ni*16 + fq*4 .. +3
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = m0 + wm * 32 + mi * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 32 + ni * 16 + fq * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[ni][mi];
      if (p.ksplit > 1) {
        float* dst = p.partial + ((size_t)ks * p.M + m) * p.ldp + n;      // ldp % 4 == 0, n % 4 == 0
        *(f32x4*)dst = v;
        continue;
      }
      if (p.vec_ok) {
        if (p.bias) { const f32x4 b = *(const f32x4*)(p.bias + n); v += b; }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
        if (p.R) { const f32x4 r = *(const f32x4*)(p.R + (size_t)m * p.ldr + n); v += r; }
        *(f32x4*)(p.C + (size_t)m * p.ldc + n) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= p.N) break;
          float x = v[e] + (p.bias ? p.bias[n + e] : 0.f);
          x = apply_act(x, p.act);
          if (p.R) x += p.R[(size_t)m * p.ldr + n + e];
          p.C[(size_t)m * p.ldc + n + e] = x;
        }
      }
    }
  }
}

// sums the split-K partials and applies the epilogue; one thread per 4 consecutive n
__global__ void splitk_reduce_kernel(const SmallGemmParams p) {
  const int nq = p.ldp / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)p.M * nq) return;
  const int m = (int)(i / nq), n = (int)(i % nq) * 4;
  if (n >= p.N) return;
  f32x4 v = *(const f32x4*)(p.partial + (size_t)m * p.ldp + n);
  for (int ks = 1; ks < p.ksplit; ++ks) v += *(const f32x4*)(p.partial + ((size_t)ks * p.M + m) * p.ldp + n);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (n + e >= p.N) break;
    float x = v[e] + (p.bias ? p.bias[n + e] : 0.f);
    x = apply_act(x, p.act);
    if (p.R) x += p.R[(size_t)m * p.ldr + n + e];
    p.C[(size_t)m * p.ldc + n + e] = x;
  }
}

namespace {
int g_target_blocks = 512;     // split K until the grid has about this many workgroups
int g_max_ksplit = 16;
int g_wpe = 1;
int g_stages = 1;        // single LDS buffer: 32 KB per workgroup, more workgroups per CU (measured faster on these shapes)
struct Ws { float* p = nullptr; size_t cap = 0; } g_ws;
}  // namespace

void gemm_small_set_stages(int n) { g_stages = (n == 1) ? 1 : 2; }
void gemm_small_set_wpe(int n) { g_wpe = (n >= 4) ? 4 : 1; }
void gemm_small_set(int target_blocks, int max_ksplit) {
  if (target_blocks >= 0) g_target_blocks = target_blocks;
  if (max_ksplit >= 1) g_max_ksplit = max_ksplit;
}

bool gemm_small_supported(const float* A, int lda, int K) {
  return (K % 4 == 0) && (lda % 4 == 0) && (((uintptr_t)A & 15) == 0);
}

int launch_gemm_small(const float* A, int lda, int M, int K, const half_t* Whi, const half_t* Wlo, int N, int Kpad, const float* bias, int act,
                      const float* R, int ldr, float* C, int ldc, int npass, hipStream_t s) {
  return launch_gemm_small_ex(A, nullptr, lda, M, K, Whi, Wlo, N, Kpad, bias, act, R, ldr, C, ldc, npass, nullptr, 0, s);
}

int launch_gemm_small_ex(const float* A, const float* A2, int lda, int M, int K, const half_t* Whi, const half_t* Wlo, int N, int Kpad,
                         const float* bias, int act, const float* R, int ldr, float* C, int ldc, int npass, float* ws, size_t ws_bytes,
                         hipStream_t s) {
  SmallGemmParams p;
  p.A = A; p.A2 = A2; p.lda = lda; p.Whi = Whi; p.Wlo = Wlo; p.M = M; p.N = N; p.K = K; p.Kpad = Kpad;
  p.bias = bias; p.act = act; p.R = R; p.ldr = ldr; p.C = C; p.ldc = ldc;
  p.tiles_m = (M + 63) / 64; p.tiles_n = (N + 63) / 64;
  const int tiles = p.tiles_m * p.tiles_n;
  const int nk = Kpad / 64;
  int ksplit = 1;
  if (tiles < g_target_blocks && nk >= 8) {
    ksplit = (g_target_blocks + tiles - 1) / tiles;
    if (ksplit > nk / 4) ksplit = nk / 4;                 // at least four k-steps per workgroup: a split costs a reduce launch
    if (ksplit > g_max_ksplit) ksplit = g_max_ksplit;
    if (ksplit < 1) ksplit = 1;
  }
  if (ws && ksplit > 1) {                                  // caller-owned workspace: as many slices as fit
    const size_t per = (size_t)M * ((N + 3) / 4 * 4) * sizeof(float);
    const size_t fit = per ? ws_bytes / per : 0;
    if ((size_t)ksplit > fit) ksplit = fit >= 2 ? (int)fit : 1;
  }
  int steps = (nk + ksplit - 1) / ksplit;
  ksplit = (nk + steps - 1) / steps;                      // no empty split
  p.ksplit = ksplit; p.kchunk = steps * 64;
  p.ldp = (N + 3) / 4 * 4; p.partial = nullptr;
  p.vec_ok = (N % 4 == 0) && (ldc % 4 == 0) && (((uintptr_t)C & 15) == 0) && (!R || ((ldr % 4 == 0) && (((uintptr_t)R & 15) == 0))) &&
             (!bias || (((uintptr_t)bias & 15) == 0));
  if (ksplit > 1) {
    const size_t need = (size_t)ksplit * M * p.ldp * sizeof(float);
    if (ws) {
      if (ws_bytes < need) return OVM_ERR_CAPACITY;
      p.partial = ws;
    } else if (g_ws.cap < need) {
      if (g_ws.p) { (void)hipDeviceSynchronize(); (void)hipFree(g_ws.p); }
      g_ws.cap = need + need / 2 + (1 << 20);
      if (hipMalloc((void**)&g_ws.p, g_ws.cap) != hipSuccess) { g_ws.p = nullptr; g_ws.cap = 0; return OVM_ERR_HIP; }
    }
    if (!ws) p.partial = g_ws.p;
  }
  const dim3 grid((unsigned)(tiles * ksplit));
  if (g_stages == 2) {
    if (npass == 3) hipLaunchKernelGGL((gemm_f32a_kernel<3, 2>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_f32a_kernel<1, 2>), grid, dim3(256), 0, s, p);
  } else {
    if (npass == 3 && g_wpe == 4) hipLaunchKernelGGL((gemm_f32a_kernel<3, 1, 4>), grid, dim3(256), 0, s, p);
    else if (npass == 3) hipLaunchKernelGGL((gemm_f32a_kernel<3, 1>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_f32a_kernel<1, 1>), grid, dim3(256), 0, s, p);
  }
  if (ksplit > 1) {
    const long n = (long)M * (p.ldp / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
  }
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

}  // namespace ovm
