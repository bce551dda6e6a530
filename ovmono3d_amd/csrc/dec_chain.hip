// GroundingDINO decoder layer as TWO row-chain kernels around the query self-attention (round 3).
//
// A decoder layer (reference roi_heads_gdino.py:186 -> GroundingDINO @856dde2 DeformableTransformerDecoderLayer, 900 queries x 256)
// is fourteen small projections, four LayerNorms, a text cross-attention over ~20 tokens, a deformable sampling and a box update -
// every one of them local to a query ROW except the self-attention over all queries. Sequenced as separate launches (rounds 1-2:
// ~35 launches per layer, ~220 for the decoder, each 7-13 us on a mostly empty chip: 2.5 ms of the detector's critical path) the
// layer is launch-latency bound. Here a workgroup owns 16 query rows and walks the whole chain with the rows resident in LDS:
//
//   chain A:  sine(ref) -> ref_point_head (2 linears) -> qpos;   [q | k] = (hs + qpos) W_qk;   v = hs W_v          -> global
//   (attn_f32_kernel: self-attention over the 900 queries, the one cross-row step)
//   chain B:  out_proj + residual + LN1;  text cross-attention (q = (hs + qpos) W_q; keys / values of the ~20 text tokens read from
//             global; softmax in LDS) + out_proj + LN2;  deformable cross-attention (offsets | weights = (hs + qpos) W_ow, softmax
//             over the 16 samples, bilinear taps on the encoder memory's value image) + out_proj + LN3;  FFN 256 -> 2048 -> 256 in
//             512-wide chunks of the hidden layer (never materialised whole) + LN4;  box MLP (3 linears) + refine -> next ref
//
// Linear layers: v_mfma_f32_16x16x32_f16 with the WEIGHT rows on the MFMA's A side - read straight from global / L2 into registers
// (every weight element is used once per workgroup, so LDS staging would only add a hop) from a copy of the split image in
// MFMA-FRAGMENT ORDER [tile of 16 rows][k-step][hi | lo][lane][8 halves]: one load instruction = 1 KiB contiguous (from the row-major
// image a quarter-wave hits 16 different rows: 64 cache lines per instruction, 16 bytes used of each - measured 13 us per
// 256 x 256 projection against 7 with this layout) - and the 16 activation rows on the B side from a split-fp16 copy in LDS; three passes per product in the order of
// every other GEMM here (lo x hi, hi x lo, hi x hi; fp32 accumulate). 57 workgroups (900 / 16) x 8 waves; a layer is 3 launches.
#include <hip/hip_runtime.h>
#include "gemm.hpp"
#include "dec_chain.hpp"

namespace ovm {

namespace {

constexpr int R = 16;                 // query rows per workgroup
constexpr int NWV = 8;                // waves per workgroup
constexpr int NT = 64 * NWV;
constexpr int KMAX = 512;             // longest K a single chain_lin call sees (the FFN's second layer runs in chunks of this)
constexpr int XLD = KMAX + 8;         // split-operand row stride in halves (+8: rows start 16 B apart in the banks)

struct Lds {
  float* x0; float* qp; float* t1; float* t2;      // [R][ldf]
  half_t* xh; half_t* xl;                           // [R][XLD]
  float* u;                                         // union: FFN hidden chunk [R][KMAX + 4] | offsets+weights [R][NOW + 4] | sine [R][2D + 4] | scores
  int ldf;
  int skip;                                         // -DOVM_DIAG builds: timing-ablation mask (DecChainParams::dbg_skip), else 0
};

__device__ __forceinline__ Lds carve(char* smem, int D, int skip) {
  Lds l; l.ldf = D + 4; l.skip = skip;
  float* f = (float*)smem;
  l.x0 = f; f += R * l.ldf; l.qp = f; f += R * l.ldf; l.t1 = f; f += R * l.ldf; l.t2 = f; f += R * l.ldf;
  l.xh = (half_t*)f; l.xl = l.xh + R * XLD;
  l.u = (float*)(l.xl + R * XLD);
  return l;
}

// Y[R][ldy] (op)= act(X[R][K] (+ X2) . W[n_off .. n_off + N)[k_off .. k_off + K)^T + bias) (+ Res)        all activations in LDS, fp32
//   accum: Y += (bias is then the caller's business: pass null); N a multiple of 16 (rows up to the weight's 128-row padding exist
//   and are zero), Nvalid <= N columns are written. Ends WITHOUT a barrier: the caller synchronises before Y is read.
//   FULL: K is a multiple of 256 - every k-batch has its 8 steps, nothing is predicated around the burst loads.
template <bool FULL>
__device__ __forceinline__ void chain_lin_t(const Lds l, const float* X, const float* X2, int ldx, int K, const ChainLin w, int k_off, int n_off, int N,
                                            int Nvalid, const float* bias, int act, const float* Res, int ldr, bool accum, float* Y, int ldy) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __syncthreads();                                        // X complete; previous readers of the split copy done
#ifdef OVM_DIAG
  if (!(l.skip & 16))
#endif
  for (int i = tid; i < R * (K >> 2); i += NT) {          // fp32 -> split fp16, four at a time
    const int row = i / (K >> 2), c = (i - row * (K >> 2)) << 2;
    f32x4 v = *(const f32x4*)(X + row * ldx + c);
    if (X2) v += *(const f32x4*)(X2 + row * ldx + c);
    half4 h4, l4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { half_t hh, ll; split_f16_nt(v[e], hh, ll); h4[e] = hh; l4[e] = ll; }
    *(half4*)(l.xh + row * XLD + c) = h4;
    *(half4*)(l.xl + row * XLD + c) = l4;
  }
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  const int ks_n = K >> 5;
  // Work unit = (16-column tile, batch of up to 8 k-steps). A unit's 16 weight fragments (hi + lo, 16 B per lane each; 1 KiB
  // contiguous per load instruction in the fragment-ordered image) are loaded in one burst, and the burst of unit u + 1 is issued
  // BEFORE the MFMAs of unit u: with 57 workgroups on the chip only the loads in flight hide the memory latency. The bursts are
  // UNCONDITIONAL (past the end they re-load the last unit): a burst inside an `if` makes the compiler wait for the older buffer
  // with a count that also drains the new burst (one static vmcnt must hold on both paths) - no overlap at all.
  const int nb = (ks_n + 7) >> 3;                              // k-batches per tile
  const int units = ((N >> 4) - wave + NWV - 1) / NWV * nb;    // this wave's tiles x batches (tiles wave, wave + NWV, ...)
  const int KS = w.Kpad >> 5;                                  // k-steps of the whole matrix (stride between tiles of the image)
  half8 wb[2][16];
  auto wload = [&](int u, half8* dst) {
    const int tile = wave + (u / nb) * NWV, kb = u - (u / nb) * nb;
    const half_t* wp = w.w + ((size_t)((n_off >> 4) + tile) * KS + (size_t)((k_off >> 5) + kb * 8)) * 1024 + lane * 8;
    const int nks = FULL ? 8 : ((ks_n - kb * 8) < 8 ? (ks_n - kb * 8) : 8);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (FULL || j < nks) { dst[2 * j] = *(const half8*)(wp + j * 1024); dst[2 * j + 1] = *(const half8*)(wp + j * 1024 + 512); }
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int u, const half8* wbuf) {
    const int tile = wave + (u / nb) * NWV, kb = u - (u / nb) * nb;
    const int nks = FULL ? 8 : ((ks_n - kb * 8) < 8 ? (ks_n - kb * 8) : 8);
    const half_t* xhp = l.xh + fr * XLD + fq * 8 + kb * 256;
    const half_t* xlp = l.xl + fr * XLD + fq * 8 + kb * 256;
    if (kb == 0) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (FULL || j < nks) {
        const half8 wh = wbuf[2 * j], wl = wbuf[2 * j + 1];
#ifdef OVM_DIAG
        if (l.skip & 32) { asm volatile("" ::"v"(wh), "v"(wl)); continue; }
#endif
        const half8 xh = *(const half8*)(xhp + j * 32);
        const half8 xl = *(const half8*)(xlp + j * 32);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc, 0, 0, 0);
      }
    if (kb == nb - 1) {
      // lane (fr, fq) holds row m = fr, columns n = tile * 16 + 4 fq + {0..3}
      const int nr = tile * 16 + fq * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = nr + e;
        if (n < Nvalid) {
          float y = acc[e];
          if (bias) y += bias[n];
          if (act == 1) y = fmaxf(y, 0.f);
          if (Res) y += Res[fr * ldr + n];
          float* yp = Y + fr * ldy + n;
          *yp = accum ? (*yp + y) : y;
        }
      }
    }
  };
  if (units > 0) {
#ifdef OVM_DIAG
    if (l.skip & 64) {                                         // no weight loads at all (garbage operands): what the MFMA / LDS side costs alone
      for (int u = 0; u < units; ++u) compute(u, wb[0]);
      return;
    }
#endif
    wload(0, wb[0]);
    for (int u = 0; u < units; u += 2) {                       // unrolled by two: the buffer a unit reads is fixed at compile time
      wload(u + 1 < units ? u + 1 : units - 1, wb[1]);
      compute(u, wb[0]);
      wload(u + 2 < units ? u + 2 : units - 1, wb[0]);
      if (u + 1 < units) compute(u + 1, wb[1]);
    }
  }
}

__device__ __forceinline__ void chain_lin(const Lds l, const float* X, const float* X2, int ldx, int K, const ChainLin w, int k_off, int n_off, int N,
                                          int Nvalid, const float* bias, int act, const float* Res, int ldr, bool accum, float* Y, int ldy) {
#ifdef OVM_DIAG
  if (l.skip & 2) { __syncthreads(); return; }
#endif
  if ((K & 255) == 0) chain_lin_t<true>(l, X, X2, ldx, K, w, k_off, n_off, N, Nvalid, bias, act, Res, ldr, accum, Y, ldy);
  else chain_lin_t<false>(l, X, X2, ldx, K, w, k_off, n_off, N, Nvalid, bias, act, Res, ldr, accum, Y, ldy);
}

// LayerNorm of the R rows of X (LDS) -> Y (LDS, may alias X): 32 lanes per row
__device__ __forceinline__ void chain_ln(const float* X, int ldx, int D, const ChainLn w, float eps, float* Y, int ldy) {
  __syncthreads();
  const int tid = threadIdx.x, row = tid >> 5, l32 = tid & 31;
  float sum = 0.f;
  for (int c = l32; c < D; c += 32) sum += X[row * ldx + c];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 32);
  const float mean = sum / (float)D;
  float sq = 0.f;
  for (int c = l32; c < D; c += 32) { const float d = X[row * ldx + c] - mean; sq += d * d; }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 32);
  const float rstd = 1.0f / sqrtf(sq / (float)D + eps);
  for (int c = l32; c < D; c += 32) Y[row * ldy + c] = (X[row * ldx + c] - mean) * rstd * w.g[c] + w.b[c];
}

__device__ __forceinline__ void load_rows(const float* g, int ldg, int row0, int nrows_total, int D, float* dst, int ldd) {
  for (int i = threadIdx.x; i < R * (D >> 2); i += NT) {
    const int row = i / (D >> 2), c = (i - row * (D >> 2)) << 2;
    int gr = row0 + row; if (gr > nrows_total - 1) gr = nrows_total - 1;       // rows past the end repeat the last one (never stored)
    *(f32x4*)(dst + row * ldd + c) = *(const f32x4*)(g + (size_t)gr * ldg + c);
  }
}
__device__ __forceinline__ void store_rows(const float* src, int lds_, int row0, int nrows_total, int N, float* g, int ldg) {
  for (int i = threadIdx.x; i < R * (N >> 2); i += NT) {
    const int row = i / (N >> 2), c = (i - row * (N >> 2)) << 2;
    if (row0 + row < nrows_total) *(f32x4*)(g + (size_t)(row0 + row) * ldg + c) = *(const f32x4*)(src + row * lds_ + c);
  }
}

}  // namespace

__device__ __forceinline__ void chain_tail(const DecChainParams& p, const Lds& l, int row0);

// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void dec_chain_a_kernel(const DecChainParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Lds l = carve(smem, p.D, p.dbg_skip);
  const int D = p.D, row0 = blockIdx.x * R, tid = threadIdx.x;
  load_rows(p.hs, D, row0, p.Q, D, l.x0, l.ldf);
  // sine embedding of the reference boxes (DETR convention: slots [y, x, w, h], temperature 10000): u[row][slot * F + f]
  const int F = D >> 1, lds_s = 2 * D + 4;
#ifdef OVM_DIAG
  if (!(p.dbg_skip & 1))
#endif
  for (int i = tid; i < R * 4 * F; i += NT) {
    const int row = i / (4 * F), rem = i - row * 4 * F, slot = rem / F, f = rem - slot * F;
    int gr = row0 + row; if (gr > p.Q - 1) gr = p.Q - 1;
    const int c = slot == 0 ? 1 : (slot == 1 ? 0 : slot);
    const float dim_t = p.sine_dim_t[f >> 1];                 // 10000^(2 (f / 2) / F), tabulated at create (powf per element: 13 us per layer)
    const float e = p.ref[(size_t)gr * 4 + c] * 6.283185307179586f / dim_t;
    l.u[row * lds_s + rem] = (f & 1) ? cosf(e) : sinf(e);
  }
  chain_lin(l, l.u, nullptr, lds_s, 2 * D, p.ref0, 0, 0, D, D, p.ref0.bias, 1, nullptr, 0, false, l.t1, l.ldf);
  chain_lin(l, l.t1, nullptr, l.ldf, D, p.ref1, 0, 0, D, D, p.ref1.bias, 0, nullptr, 0, false, l.qp, l.ldf);
  __syncthreads();
  store_rows(l.qp, l.ldf, row0, p.Q, D, p.qpos, D);
  // [q | k] = (hs + qpos) W_qk: two halves of D columns through t1 / t2;   v = hs W_v
  chain_lin(l, l.x0, l.qp, l.ldf, D, p.sa_qk, 0, 0, D, D, p.sa_qk.bias, 0, nullptr, 0, false, l.t1, l.ldf);
  chain_lin(l, l.x0, l.qp, l.ldf, D, p.sa_qk, 0, D, D, D, p.sa_qk.bias + D, 0, nullptr, 0, false, l.t2, l.ldf);
  __syncthreads();
  store_rows(l.t1, l.ldf, row0, p.Q, D, p.qk, 2 * D);
  store_rows(l.t2, l.ldf, row0, p.Q, D, p.qk + D, 2 * D);
  chain_lin(l, l.x0, nullptr, l.ldf, D, p.sa_v, 0, 0, D, D, p.sa_v.bias, 0, nullptr, 0, false, l.t1, l.ldf);
  __syncthreads();
  store_rows(l.t1, l.ldf, row0, p.Q, D, p.v, D);
}

// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void dec_chain_b_kernel(const DecChainParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Lds l = carve(smem, p.D, p.dbg_skip);
  const int D = p.D, row0 = blockIdx.x * R, tid = threadIdx.x;
  const int H = p.heads, dh = D / H;
  load_rows(p.hs, D, row0, p.Q, D, l.x0, l.ldf);
  load_rows(p.qpos, D, row0, p.Q, D, l.qp, l.ldf);
  load_rows(p.ctx, D, row0, p.Q, D, l.t1, l.ldf);
  // ---- self-attention output projection + residual + LN1
  chain_lin(l, l.t1, nullptr, l.ldf, D, p.sa_out, 0, 0, D, D, p.sa_out.bias, 0, l.x0, l.ldf, false, l.t2, l.ldf);
  chain_ln(l.t2, l.ldf, D, p.ln1, p.eps, l.x0, l.ldf);
  // ---- text cross-attention: q = (hs + qpos) W_q; scores over the T text tokens per (row, head); softmax; values
  chain_lin(l, l.x0, l.qp, l.ldf, D, p.ca_q, 0, 0, D, D, p.ca_q.bias, 0, nullptr, 0, false, l.t1, l.ldf);
  __syncthreads();
#ifdef OVM_DIAG
  if (!(p.dbg_skip & 4))
#endif
  {
    const int T = p.T, npair = R * H;
    const float scale = 1.0f / sqrtf((float)dh);
    float* sc = l.u;                                         // [R * H][T]
    for (int i = tid; i < npair * T; i += NT) {
      const int pair = i / T, key = i - pair * T, row = pair / H, hh = pair - row * H;
      const float* q = l.t1 + row * l.ldf + hh * dh;
      const float* k = p.tk + (size_t)key * p.ldt + hh * dh;
      float s = 0.f;
      for (int d = 0; d < dh; d += 4) {
        const f32x4 kv = *(const f32x4*)(k + d);
        s = fmaf(q[d], kv[0], s); s = fmaf(q[d + 1], kv[1], s); s = fmaf(q[d + 2], kv[2], s); s = fmaf(q[d + 3], kv[3], s);
      }
      sc[i] = s * scale;
    }
    __syncthreads();
    if (tid < npair) {
      float* s = sc + tid * T;
      float mx = -INFINITY;
      for (int k = 0; k < T; ++k) mx = fmaxf(mx, s[k]);
      float den = 0.f;
      for (int k = 0; k < T; ++k) { s[k] = expf(s[k] - mx); den += s[k]; }
      const float inv = 1.0f / den;
      for (int k = 0; k < T; ++k) s[k] *= inv;
    }
    __syncthreads();
    for (int i = tid; i < R * D; i += NT) {
      const int row = i / D, c = i - row * D, hh = c / dh;
      const float* s = sc + (row * H + hh) * T;
      float o = 0.f;
      for (int k = 0; k < T; ++k) o = fmaf(s[k], p.tv[(size_t)k * p.ldt + c], o);
      l.t2[row * l.ldf + c] = o;
    }
  }
  chain_lin(l, l.t2, nullptr, l.ldf, D, p.ca_out, 0, 0, D, D, p.ca_out.bias, 0, l.x0, l.ldf, false, l.t1, l.ldf);
  chain_ln(l.t1, l.ldf, D, p.ln2, p.eps, l.x0, l.ldf);
  // ---- deformable cross-attention on the encoder memory: offsets | attention logits = (hs + qpos) W_ow
  const int LP = p.L * p.P, NOW = H * LP * 3, ldo = NOW + 4;
  chain_lin(l, l.x0, l.qp, l.ldf, D, p.offw, 0, 0, (NOW + 15) & ~15, NOW, p.offw.bias, 0, nullptr, 0, false, l.u, ldo);
  __syncthreads();
#ifdef OVM_DIAG
  if (!(p.dbg_skip & 8))
#endif
  {
    const int dq = dh >> 2;
    for (int i = tid; i < R * H * dq; i += NT) {
      const int d = (i % dq) * 4; const int rr = i / dq; const int hh = rr % H, row = rr / H;
      int gr = row0 + row; if (gr > p.Q - 1) gr = p.Q - 1;
      const float* offp = l.u + row * ldo + hh * LP * 2;
      const float* lgp = l.u + row * ldo + H * LP * 2 + hh * LP;
      float mx = -INFINITY;
      for (int k = 0; k < LP; ++k) mx = fmaxf(mx, lgp[k]);
      float den = 0.f;
      for (int k = 0; k < LP; ++k) den += expf(lgp[k] - mx);
      const float inv = 1.0f / den;
      const float* rf = p.ref + (size_t)gr * 4;
      const float r0 = rf[0], r1 = rf[1], mulx = rf[2] * (0.5f / (float)p.P), muly = rf[3] * (0.5f / (float)p.P);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      int s = 0;
      for (int lv = 0; lv < p.L; ++lv) {
        const int Hh = p.lh[lv], Ww = p.lw[lv];
        const float* vb = p.val + (size_t)p.lstart[lv] * p.ldv + (size_t)hh * dh + d;
        for (int pt = 0; pt < p.P; ++pt, ++s) {
          const float lx = __fadd_rn(__fmul_rn(offp[2 * s], mulx), r0);
          const float ly = __fadd_rn(__fmul_rn(offp[2 * s + 1], muly), r1);
          const float gx = 2.f * lx - 1.f, gy = 2.f * ly - 1.f;
          const float ix = ((gx + 1.f) * (float)Ww - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)Hh - 1.f) * 0.5f;
          const float fx = floorf(ix), fy = floorf(iy);
          const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
          const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
          // branch-free taps (see msdeform_fused4_kernel): clamped address, weight 0 outside the level
          const bool vx0 = x0 >= 0 && x0 < Ww, vx1 = x1 >= 0 && x1 < Ww, vy0 = y0 >= 0 && y0 < Hh, vy1 = y1 >= 0 && y1 < Hh;
          const int cx0 = min(max(x0, 0), Ww - 1), cx1 = min(max(x1, 0), Ww - 1), cy0 = min(max(y0, 0), Hh - 1), cy1 = min(max(y1, 0), Hh - 1);
          const f32x4 a00 = *(const f32x4*)(vb + (size_t)(cy0 * Ww + cx0) * p.ldv), a01 = *(const f32x4*)(vb + (size_t)(cy0 * Ww + cx1) * p.ldv);
          const f32x4 a10 = *(const f32x4*)(vb + (size_t)(cy1 * Ww + cx0) * p.ldv), a11 = *(const f32x4*)(vb + (size_t)(cy1 * Ww + cx1) * p.ldv);
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          { const float w_ = (vy0 && vx0) ? wy0 * wx0 : 0.f; v += w_ * a00; }
          { const float w_ = (vy0 && vx1) ? wy0 * wx1 : 0.f; v += w_ * a01; }
          { const float w_ = (vy1 && vx0) ? wy1 * wx0 : 0.f; v += w_ * a10; }
          { const float w_ = (vy1 && vx1) ? wy1 * wx1 : 0.f; v += w_ * a11; }
          const float aw = expf(lgp[s] - mx) * inv;
          acc += v * aw;
        }
      }
      *(f32x4*)(l.t2 + row * l.ldf + hh * dh + d) = acc;
    }
  }
  chain_lin(l, l.t2, nullptr, l.ldf, D, p.msda_out, 0, 0, D, D, p.msda_out.bias, 0, l.x0, l.ldf, false, l.t1, l.ldf);
  chain_ln(l.t1, l.ldf, D, p.ln3, p.eps, l.x0, l.ldf);
  // ---- FFN: the hidden layer in chunks of <= 512 columns; y accumulates in t1 (starts as bias + residual)
  {
    const int ffn = p.ffn, chunk = ffn < KMAX ? ffn : KMAX, ldh = KMAX + 4;
    __syncthreads();
    if (p.ffn_split > 1) {                                   // this workgroup's chunk only; chain C adds the partials up
      const int c0 = (int)blockIdx.y * chunk;
      if (blockIdx.y == 0) store_rows(l.x0, l.ldf, row0, p.Q, D, p.ffn_x, D);
      chain_lin(l, l.x0, nullptr, l.ldf, D, p.fc1, 0, c0, chunk, chunk, p.fc1.bias + c0, 1, nullptr, 0, false, l.u, ldh);
      chain_lin(l, l.u, nullptr, ldh, chunk, p.fc2, c0, 0, D, D, nullptr, 0, nullptr, 0, false, l.t1, l.ldf);
      __syncthreads();
      store_rows(l.t1, l.ldf, row0, p.Q, D, p.ffn_part + (size_t)blockIdx.y * p.Q * D, D);
      return;
    }
    for (int i = tid; i < R * D; i += NT) { const int row = i / D, c = i - row * D; l.t1[row * l.ldf + c] = l.x0[row * l.ldf + c] + p.fc2.bias[c]; }
    for (int c0 = 0; c0 < ffn; c0 += chunk) {
      chain_lin(l, l.x0, nullptr, l.ldf, D, p.fc1, 0, c0, chunk, chunk, p.fc1.bias + c0, 1, nullptr, 0, false, l.u, ldh);
      chain_lin(l, l.u, nullptr, ldh, chunk, p.fc2, c0, 0, D, D, nullptr, 0, nullptr, 0, true, l.t1, l.ldf);
    }
  }
  chain_tail(p, l, row0);
}

// LN4 of the FFN output in t1 -> decoder state; iterative box refinement
__device__ __forceinline__ void chain_tail(const DecChainParams& p, const Lds& l, int row0) {
  const int D = p.D, tid = threadIdx.x;
  chain_ln(l.t1, l.ldf, D, p.ln4, p.eps, l.x0, l.ldf);
  __syncthreads();
  store_rows(l.x0, l.ldf, row0, p.Q, D, p.hs, D);
  // ---- iterative box refinement: delta = MLP(hs); ref' = sigmoid(delta + logit(clamp(ref)))
  if (p.ref_next) {
    chain_lin(l, l.x0, nullptr, l.ldf, D, p.bb0, 0, 0, D, D, p.bb0.bias, 1, nullptr, 0, false, l.t1, l.ldf);
    chain_lin(l, l.t1, nullptr, l.ldf, D, p.bb1, 0, 0, D, D, p.bb1.bias, 1, nullptr, 0, false, l.t2, l.ldf);
    chain_lin(l, l.t2, nullptr, l.ldf, D, p.bb2, 0, 0, 16, 4, p.bb2.bias, 0, nullptr, 0, false, l.t1, l.ldf);
    __syncthreads();
    if (tid < R * 4) {
      const int row = tid >> 2, c = tid & 3;
      if (row0 + row < p.Q) {
        const float eps = 1e-5f;
        const float xc = fminf(fmaxf(p.ref[(size_t)(row0 + row) * 4 + c], eps), 1.f - eps);
        const float x = l.t1[row * l.ldf + c] + logf(xc / (1.f - xc));
        p.ref_next[(size_t)(row0 + row) * 4 + c] = 1.0f / (1.0f + expf(-x));
      }
    }
  }
}

// chain C (ffn_split > 1): t1 = (x + b2) + partial 0 + partial 1 + ... (the single workgroup's accumulation order), then the tail
__global__ __launch_bounds__(NT) void dec_chain_c_kernel(const DecChainParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Lds l = carve(smem, p.D, p.dbg_skip);
  const int D = p.D, row0 = blockIdx.x * R, tid = threadIdx.x;
  for (int i = tid; i < R * D; i += NT) {
    const int row = i / D, c = i - row * D;
    int gr = row0 + row; if (gr > p.Q - 1) gr = p.Q - 1;
    float y = p.ffn_x[(size_t)gr * D + c] + p.fc2.bias[c];
    for (int sl = 0; sl < p.ffn_split; ++sl) y += p.ffn_part[((size_t)sl * p.Q + gr) * D + c];
    l.t1[row * l.ldf + c] = y;
  }
  chain_tail(p, l, row0);
}

size_t dec_chain_lds_bytes(int D) {
  return (size_t)4 * R * (D + 4) * sizeof(float) + (size_t)2 * R * XLD * sizeof(half_t) + (size_t)R * (KMAX + 4) * sizeof(float);
}

bool dec_chain_supported(int D, int heads, int ffn, int L, int P, int T, int npass) {
  if (npass != 3 || D % 32 != 0 || D > 256 || D % heads != 0 || (D / heads) % 4 != 0) return false;
  if (ffn % 32 != 0 || (ffn > KMAX && ffn % KMAX != 0)) return false;
  if (2 * D > KMAX || heads * L * P * 3 > KMAX || L > 8 || (L * P) % 2 != 0) return false;
  if ((size_t)R * heads * T > (size_t)R * (KMAX + 4)) return false;            // scores [R * heads][T] live in the union region
  return true;
}

int launch_dec_chain(const DecChainParams& p, int part, hipStream_t s) {
  const int smem = (int)dec_chain_lds_bytes(p.D);
  static bool set = false;
  if (!set) {
    if (hipFuncSetAttribute((const void*)dec_chain_a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)dec_chain_b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)dec_chain_c_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return OVM_ERR_HIP;
    set = true;
  }
  const dim3 grid((p.Q + R - 1) / R), block(NT);
  if (p.ffn_split > 1) {
    const int chunk = p.ffn < KMAX ? p.ffn : KMAX;
    if (p.ffn_split * chunk != p.ffn || !p.ffn_x || !p.ffn_part) return OVM_ERR_INVALID;
  } else if (part == 2) return OVM_ERR_INVALID;
  if (part == 0) hipLaunchKernelGGL(dec_chain_a_kernel, grid, block, smem, s, p);
  else if (part == 1) hipLaunchKernelGGL(dec_chain_b_kernel, dim3(grid.x, p.ffn_split > 1 ? p.ffn_split : 1), block, smem, s, p);
  else hipLaunchKernelGGL(dec_chain_c_kernel, grid, block, smem, s, p);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

}  // namespace ovm
