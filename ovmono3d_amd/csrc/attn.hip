// Fused (flash-style) multi-head attention for the ViT blocks, head dim 64, gfx950.
//
//   softmax(q k^T) v with q pre-scaled by dh^-0.5 in the QKV epilogue (dinov2 Attention.forward as the
//   reference reaches it at dino.py:89-90; the xFormers path of nohup.out:696-701 computes the same).
//
// Work split: one workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.
// K and V^T tiles of 64 keys arrive by global_load_lds (two LDS stages, source-side swizzle).
// Orientation ("key on the MFMA row"): S^T = K Q^T with v_mfma_f32_32x32x16_f16, so a lane holds one
// query column and 32 of the 64 keys in registers - softmax needs one cross-half shuffle per tile and
// the S^T accumulator is directly the B operand of O^T = V^T P^T (no LDS round trip for P).
// V^T is stored with the token order permuted inside groups of 16 (bits 2<->3) by the QKV epilogue so
// the PV A-fragment, whose k order is {4h..4h+3, 8+4h..8+4h+3}, is a single ds_read_b128.
//
// NPASS=3 uses split operands (hi + lo) for both products, three MFMAs per product into the same accumulator.
#include "kernels.hpp"

namespace ovm {

__device__ __forceinline__ int swz128(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <int NPASS>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PART = 64 * 128;                       // 64 rows x 128 B
  constexpr int NPART = (NPASS == 3) ? 4 : 2;          // Khi, Vhi, (Klo, Vlo)
  constexpr int STAGE = PART * NPART;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nqb = (p.Tq + 127) >> 7;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = bid / nqb, qb = bid - bh * nqb;
  const int b = bh / p.heads, head = bh - b * p.heads;
  const int T = p.T;
  const size_t qk_base = (size_t)bh * T * 64;
  const size_t v_base = (size_t)bh * 64 * p.Tpad;
  const int h = lane >> 5, r = lane & 31;

  // ---- Q^T fragments from HBM (B operand: lane holds Q[q][16s + 8h + j]) ----
  int q = qb * 128 + wave * 32 + r;
  const bool q_ok = q < p.Tq;
  if (!q_ok) q = T - 1;
  half8 qh[4], ql[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qh[s] = *(const half8*)(p.Qhi + qk_base + (size_t)q * 64 + 16 * s + 8 * h);
    if (NPASS == 3) ql[s] = *(const half8*)(p.Qlo + qk_base + (size_t)q * 64 + 16 * s + 8 * h);
  }

  // ---- DMA plan: per part 8 wave-instructions (8 rows x 128 B each); this wave issues 2 per part ----
  auto stage = [&](int buf, int it) {
    char* base = smem + buf * STAGE;
    const int k0 = it * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int instr = wave + 4 * t;
      const int row = instr * 8 + (lane >> 3);
      const int chunk = swz128(row, lane & 7);
      int key = k0 + row; if (key > T - 1) key = T - 1;
      const size_t ko = qk_base + (size_t)key * 64 + chunk * 8;
      const size_t vo = v_base + (size_t)row * p.Tpad + k0 + chunk * 8;
      char* dst = base + instr * 1024;
      glds16(p.Khi + ko, dst);
      glds16(p.Vhi + vo, dst + PART);
      if (NPASS == 3) {
        glds16(p.Klo + ko, dst + 2 * PART);
        glds16(p.Vlo + vo, dst + 3 * PART);
      }
    }
  };

  f32x16 o0[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o0[t][i] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int nt = (T + 63) >> 6;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < nt; ++it) {
    if (it + 1 < nt) stage(cur ^ 1, it + 1);
    const char* base = smem + cur * STAGE;

    // ---- S^T = K Q^T : two 32-key tiles ----
    f32x16 s0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s0[i][e] = 0.f;
      const int row = 32 * i + r;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int off = row * 128 + swz128(row, 2 * s + h) * 16;
        const half8 kh = *(const half8*)(base + off);
        if (NPASS == 3) {
          const half8 kl = *(const half8*)(base + 2 * PART + off);
          s0[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], s0[i], 0, 0, 0);
          s0[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], s0[i], 0, 0, 0);
        }
        s0[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], s0[i], 0, 0, 0);
      }
    }
    // ---- online softmax over the 64 keys of this tile (keys >= T masked) ----
    const int kbase = it * 64;
    const bool tail = (kbase + 64 > T);
    float mx = -1e30f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = s0[i][e];
        if (tail) {
          const int key = kbase + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= T) v = -1e30f;
        }
        s0[i][e] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    half8 ph[2][2], pl[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __expf(s0[i][e] - m_new);
        psum += pv;
        half_t hh, ll; split_f16(pv, hh, ll);
        ph[i][e >> 3][e & 7] = hh;
        if (NPASS == 3) pl[i][e >> 3][e & 7] = ll;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) o0[t][e] *= alpha;

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 32 * t + r;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          const int off = row * 128 + swz128(row, 4 * i + 2 * sp + h) * 16;
          const half8 vh = *(const half8*)(base + PART + off);
          if (NPASS == 3) {
            const half8 vl = *(const half8*)(base + 3 * PART + off);
            o0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[i][sp], o0[t], 0, 0, 0);
            o0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[i][sp], o0[t], 0, 0, 0);
          }
          o0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[i][sp], o0[t], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // ---- normalise and store: lane holds query q, dh = 32t + 8g + 4h + {0..3} ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
    const size_t orow = ((size_t)b * T + q) * p.ldo + head * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 hv, lv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          half_t hh, ll; split_f16(o0[t][4 * g + e] * inv, hh, ll); hv[e] = hh; lv[e] = ll;
        }
        const int d = 32 * t + 8 * g + 4 * h;
        *(half4*)(p.Ohi + orow + d) = hv;
        if (p.Olo) *(half4*)(p.Olo + orow + d) = lv;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Tail query rows. T = 4097 leaves ONE query past 32 blocks of 128; as a 33rd block per head it would
// add a second, nearly empty round of workgroups. Up to 8 leftover queries per (batch, head) are handled
// here instead: one workgroup per (query, batch*head), scores and probabilities in LDS, fp32 FMAs on the
// reconstructed (hi + lo) operands.
// ---------------------------------------------------------------------------------------------
template <int NPASS>
__global__ __launch_bounds__(256) void attn_tail_kernel(const AttnParams p, int q_begin) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sc = (float*)smem;                       // [Tpad] scores -> probabilities, in V^T's permuted token order
  __shared__ float red[8];
  __shared__ float part[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, q = q_begin + blockIdx.y;
  const int b = bh / p.heads, head = bh - b * p.heads;
  const int T = p.T, Tpad = p.Tpad;
  const size_t qk_base = (size_t)bh * T * 64;
  const size_t v_base = (size_t)bh * 64 * Tpad;
  float qv[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) {
    qv[d] = (float)p.Qhi[qk_base + (size_t)q * 64 + d];
    if (NPASS == 3) qv[d] += (float)p.Qlo[qk_base + (size_t)q * 64 + d];
  }
  float mx = -1e30f;
  for (int t = tid; t < Tpad; t += 256) {
    float s = -1e30f;
    if (t < T) {
      s = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const half8 kh = *(const half8*)(p.Khi + qk_base + (size_t)t * 64 + c * 8);
        half8 kl;
        if (NPASS == 3) kl = *(const half8*)(p.Klo + qk_base + (size_t)t * 64 + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float kv = (float)kh[j];
          if (NPASS == 3) kv += (float)kl[j];
          s = fmaf(qv[c * 8 + j], kv, s);
        }
      }
    }
    const int tp = (t & ~15) | (t & 3) | ((t & 4) << 1) | ((t & 8) >> 1);
    sc[tp] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int t = tid; t < Tpad; t += 256) {
    const float e = __expf(sc[t] - mx);          // masked slots hold -1e30 -> 0
    sc[t] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  // O[d] = sum_t p[t] V[t][d]; wave w covers a quarter of the (permuted) token axis, lane = d
  const int d = lane;
  const int q4 = Tpad / 4;                        // Tpad % 64 == 0
  float o = 0.f;
  const half_t* vh = p.Vhi + v_base + (size_t)d * Tpad + wave * q4;
  const half_t* vl = (NPASS == 3) ? p.Vlo + v_base + (size_t)d * Tpad + wave * q4 : nullptr;
  const float* pr = sc + wave * q4;
  for (int j = 0; j < q4; j += 8) {
    const half8 h8 = *(const half8*)(vh + j);
    half8 l8;
    if (NPASS == 3) l8 = *(const half8*)(vl + j);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = (float)h8[e];
      if (NPASS == 3) v += (float)l8[e];
      o = fmaf(pr[j + e], v, o);
    }
  }
  part[wave][d] = o;
  __syncthreads();
  if (wave == 0) {
    const float r = (part[0][d] + part[1][d] + part[2][d] + part[3][d]) * inv;
    half_t hh, ll; split_f16(r, hh, ll);
    const size_t oo = ((size_t)b * T + q) * p.ldo + head * 64 + d;
    p.Ohi[oo] = hh;
    if (p.Olo) p.Olo[oo] = ll;
  }
}

static int g_attn_tail = 1;
void attn_set_tail_rows(int on) { g_attn_tail = on; }

int launch_attention(const AttnParams& p, int npass, hipStream_t s) {
  if (p.T <= 0 || p.B <= 0) return OVM_OK;
  if (p.Tpad % 64 != 0 || p.Tpad < ((p.T + 63) / 64) * 64) return OVM_ERR_SHAPE;
  if (npass == 3 && (!p.Qlo || !p.Klo || !p.Vlo)) return OVM_ERR_INVALID;
  AttnParams pm = p;
  const int tail = p.T % 128;
  if (g_attn_tail && tail > 0 && tail <= 8 && p.T > 128 && p.Tpad * 4 <= 60000) {
    const dim3 tg(p.heads * p.B, tail);
    if (npass == 3) hipLaunchKernelGGL(attn_tail_kernel<3>, tg, dim3(256), p.Tpad * 4, s, p, p.T - tail);
    else hipLaunchKernelGGL(attn_tail_kernel<1>, tg, dim3(256), p.Tpad * 4, s, p, p.T - tail);
    pm.Tq = p.T - tail;
  } else {
    pm.Tq = p.T;
  }
  const int nqb = (pm.Tq + 127) / 128;
  const dim3 grid(nqb * p.heads * p.B), block(256);
  if (npass == 3) hipLaunchKernelGGL(attn_kernel<3>, grid, block, 2 * 4 * 64 * 128, s, pm);
  else hipLaunchKernelGGL(attn_kernel<1>, grid, block, 2 * 2 * 64 * 128, s, pm);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

}  // namespace ovm
