// Leftover-query workgroups of the attention kernels (shared by attn.hip and attn64.hip).
#pragma once
#include "kernels.hpp"

namespace ovm {

// ---------------------------------------------------------------------------------------------
// Tail query rows. T = 4097 leaves ONE query past 32 blocks of 128; as a 33rd block per head it would
// add a second, nearly empty round of workgroups. Up to 8 leftover queries per (batch, head) are handled by
// extra workgroups of the same launch instead: scores and probabilities in LDS, fp32 FMAs on the
// reconstructed (hi + lo) operands, all global reads coalesced along the contiguous axis.
// ---------------------------------------------------------------------------------------------
template <int NPASS>
__device__ __forceinline__ void attn_tail_body(const AttnParams& p, int tb, char* smem) {
  float* sc = (float*)smem;                       // [Tpad] scores -> probabilities, in V^T's permuted token order
  float* red = sc + p.Tpad;                       // [16]: per-wave maxima, per-wave sums
  const int nw = blockDim.x >> 6;                 // 4 or 8 waves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntail = p.T - p.Tq;
  const int bh = tb / ntail, q = p.Tq + (tb - bh * ntail);
  const int b = bh / p.heads, head = bh - b * p.heads;
  const int T = p.T, Tpad = p.Tpad;
  const size_t qk_base = (size_t)bh * T * 64;
  const size_t v_base = (size_t)bh * 64 * Tpad;
  // lane l of a wave holds q[8*(l&7) .. +8): 8 lanes cover one key row (128 B contiguous), 8 keys per wave pass
  float qv[8];
  {
    const half8 qh = *(const half8*)(p.Qhi + qk_base + (size_t)q * 64 + 8 * (lane & 7));
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[j] = (float)qh[j];
    if (NPASS == 3) {
      const half8 ql = *(const half8*)(p.Qlo + qk_base + (size_t)q * 64 + 8 * (lane & 7));
#pragma unroll
      for (int j = 0; j < 8; ++j) qv[j] += (float)ql[j];
    }
  }
  float mx = -1e30f;
  for (int t0 = wave * 8; t0 < Tpad; t0 += 8 * nw) {
    const int t = t0 + (lane >> 3);
    float s = 0.f;
    if (t < T) {
      const half8 kh = *(const half8*)(p.Khi + qk_base + (size_t)t * 64 + 8 * (lane & 7));
      half8 kl;
      if (NPASS == 3) kl = *(const half8*)(p.Klo + qk_base + (size_t)t * 64 + 8 * (lane & 7));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float kv = (float)kh[j];
        if (NPASS == 3) kv += (float)kl[j];
        s = fmaf(qv[j], kv, s);
      }
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (t >= T) s = -1e30f;
    if ((lane & 7) == 0) {
      const int tp = (t & ~15) | (t & 3) | ((t & 4) << 1) | ((t & 8) >> 1);
      sc[tp] = s;
    }
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  if (nw == 8) mx = fmaxf(mx, fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
  float sum = 0.f;
  for (int t = tid; t < Tpad; t += blockDim.x) {
    const float e = __builtin_amdgcn_exp2f(sc[t] - mx);   // scores are in log2 units; masked slots hold -1e30 -> 0
    sc[t] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[8 + wave] = sum;
  __syncthreads();
  float tot = red[8] + red[9] + red[10] + red[11];
  if (nw == 8) tot += red[12] + red[13] + red[14] + red[15];
  const float inv = 1.0f / tot;
  // O[d] = sum_t p[t] V^T[d][t]: wave w owns d = w, w + nw, ...; lanes run along the token axis (coalesced)
  for (int d = wave; d < 64; d += nw) {
    float o = 0.f;
    for (int j = lane * 8; j < Tpad; j += 512) {
      const half8 h8 = *(const half8*)(p.Vhi + v_base + (size_t)d * Tpad + j);
      half8 l8;
      if (NPASS == 3) l8 = *(const half8*)(p.Vlo + v_base + (size_t)d * Tpad + j);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = (float)h8[e];
        if (NPASS == 3) v += (float)l8[e];
        o = fmaf(sc[j + e], v, o);
      }
    }
    o = wave_sum(o);
    if (lane == 0) {
      half_t hh, ll; split_f16(o * inv, hh, ll);
      const size_t oo = ((size_t)b * T + q) * p.ldo + (p.o_il ? il_col(head * 64 + d) : head * 64 + d);
      p.Ohi[oo] = hh;
      if (p.Olo) p.Olo[oo] = ll;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same leftover queries SPLIT OVER THE KEYS (round 3). One workgroup per leftover query reads the whole K and V^T of its head -
// 2 MB through one CU at the ~50 GB/s a CU sustains: ~60 us, as a second round behind the 256 main workgroups that is a QUARTER of
// the launch (rocprofv3, ViT-L: T = 4096 -> 201 us, T = 4097 -> 266 us per launch). Here kSplit workgroups per (head, leftover
// query) take a contiguous run of 64-key tiles each (16 heads x 16 = 256 workgroups of ~4 us), leave their partial online-softmax
// state (m, l, o[64]) in a workspace, and attn_tail_combine_kernel (below) merges the partials in index order - deterministic - and
// writes the row.
// ---------------------------------------------------------------------------------------------
constexpr int kTailSplit = 16;
constexpr int kTailRec = 68;                       // floats per partial: m, l, pad, pad, o[64]

template <int NPASS>
__device__ __forceinline__ void attn_tail_split_body(const AttnParams& p, int tb, char* smem) {
  float* sc = (float*)smem;                       // [chunk keys] scores -> probabilities, in V^T's permuted token order
  const int nw = blockDim.x >> 6;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntail = p.T - p.Tq;
  const int sp = tb % kTailSplit; const int tq = tb / kTailSplit;       // (bh, leftover query) = tq, key split = sp
  const int bh = tq / ntail, q = p.Tq + (tq - bh * ntail);
  const int b = bh / p.heads, head = bh - b * p.heads;
  const int T = p.T, Tpad = p.Tpad;
  const int ntile = Tpad >> 6, per = (ntile + kTailSplit - 1) / kTailSplit;
  const int k0 = sp * per * 64, k1 = (k0 + per * 64 < Tpad) ? k0 + per * 64 : Tpad;     // this workgroup's keys [k0, k1) (may be empty)
  const int nk = k1 > k0 ? k1 - k0 : 0;
  float* red = sc + per * 64;                     // [16]
  const size_t qk_base = (size_t)bh * T * 64;
  const size_t v_base = (size_t)bh * 64 * Tpad;
  float qv[8];
  {
    const half8 qh = *(const half8*)(p.Qhi + qk_base + (size_t)q * 64 + 8 * (lane & 7));
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[j] = (float)qh[j];
    if (NPASS == 3) {
      const half8 ql = *(const half8*)(p.Qlo + qk_base + (size_t)q * 64 + 8 * (lane & 7));
#pragma unroll
      for (int j = 0; j < 8; ++j) qv[j] += (float)ql[j];
    }
  }
  float mx = -1e30f;
  for (int t0 = wave * 8; t0 < nk; t0 += 8 * nw) {
    const int t = k0 + t0 + (lane >> 3);
    float s = 0.f;
    if (t < T) {
      const half8 kh = *(const half8*)(p.Khi + qk_base + (size_t)t * 64 + 8 * (lane & 7));
      half8 kl;
      if (NPASS == 3) kl = *(const half8*)(p.Klo + qk_base + (size_t)t * 64 + 8 * (lane & 7));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float kv = (float)kh[j];
        if (NPASS == 3) kv += (float)kl[j];
        s = fmaf(qv[j], kv, s);
      }
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (t >= T) s = -1e30f;
    if ((lane & 7) == 0) {
      const int tp = (t & ~15) | (t & 3) | ((t & 4) << 1) | ((t & 8) >> 1);
      sc[tp - k0] = s;                            // k0 is a multiple of 64: the permutation stays inside the chunk
    }
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
  for (int w = 1; w < nw; ++w) mx = fmaxf(mx, red[w]);
  float sum = 0.f;
  for (int t = tid; t < nk; t += blockDim.x) {
    const float e = __builtin_amdgcn_exp2f(sc[t] - mx);   // masked slots hold -1e30 -> 0 (an all-masked chunk: every term exp2(0) = 1
    sc[t] = e;                                            //  with mx = -1e30, but then nk covers only padding and l is discarded below)
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[8 + wave] = sum;
  __syncthreads();
  float tot = 0.f;
  for (int w = 0; w < nw; ++w) tot += red[8 + w];
  const bool any_valid = nk > 0 && k0 < T;
  float* rec = p.tail_ws + ((size_t)tq * kTailSplit + sp) * kTailRec;
  if (tid == 0) { rec[0] = any_valid ? mx : -1e30f; rec[1] = any_valid ? tot : 0.f; }
  // o[d] = sum_t p[t] V^T[d][k0 + t]: wave w owns d = w, w + nw, ...; lanes run along the token axis (coalesced)
  for (int d = wave; d < 64; d += nw) {
    float o = 0.f;
    for (int j = lane * 8; j < nk; j += 512) {
      const half8 h8 = *(const half8*)(p.Vhi + v_base + (size_t)d * Tpad + k0 + j);
      half8 l8;
      if (NPASS == 3) l8 = *(const half8*)(p.Vlo + v_base + (size_t)d * Tpad + k0 + j);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = (float)h8[e];
        if (NPASS == 3) v += (float)l8[e];
        o = fmaf(sc[j + e], v, o);
      }
    }
    o = wave_sum(o);
    if (lane == 0) rec[4 + d] = any_valid ? o : 0.f;
  }
}

// dispatch used by the kernels: split form when the launcher provided a workspace
template <int NPASS>
__device__ __forceinline__ void attn_tail_any(const AttnParams& p, int tb, char* smem) {
  if (p.tail_ws) attn_tail_split_body<NPASS>(p, tb, smem);
  else attn_tail_body<NPASS>(p, tb, smem);
}

}  // namespace ovm
