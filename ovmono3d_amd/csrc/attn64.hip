// Flash attention for the ViT blocks, head dim 64, split precision (f16x3) - the 64-queries-per-wave form (round 3).
//
// Same arithmetic as attn_kernel<3, 8> (attn.hip), query by query and MFMA by MFMA: S^T = K Q^T and O^T += V^T P^T with
// v_mfma_f32_32x32x16_f16, three passes per product in the order (lo x hi, hi x lo, hi x hi), scores in log2 units, probabilities
// carried at 2^14, the same online-softmax recurrences - so the two kernels agree BIT FOR BIT (tests/test_gpu_ops.py). What
// changes is who shares what:
//
//   attn_kernel<3, 8>: 8 waves x 32 queries. Per 64-key tile every wave reads the whole K tile and the whole V^T tile from LDS as
//                      MFMA A-fragments: 32 KiB per wave, 256 KiB per workgroup and tile = 2,048 of the tile's 3,072 MFMA cycles per
//                      SIMD at 128 B / clk - and in the measured kernel neither those reads nor the softmax VALU work overlap the
//                      matrix pipe (3,072 + 2,048 + ~2,560 cycles ~ the ~8,000 cycles per tile the kernel takes: profiles/r02).
//   attn64_kernel:     4 waves x 64 queries (two sub-tiles of 32 per wave, one wave per SIMD, up to 512 VGPRs). Every K / V^T fragment
//                      read feeds SIX MFMAs instead of three: half the fragment bytes per MFMA (128 KiB per workgroup and tile),
//                      half the LDS-DMA instructions and barrier participants per tile. The Q fragments of both sub-tiles (64 VGPRs)
//                      and four score / output accumulators per kind live in registers.
//
// Workgroup = 256 queries of one (batch, head), as before: grid, K / V^T rings (3 slots each, 96 KiB), XCD remap, leftover-query
// workgroups (attn_tail_body) are those of the 8-wave kernel.
#include "kernels.hpp"
#include "attn_tail.hpp"
#include <type_traits>

namespace ovm {

namespace {

__device__ __forceinline__ int swz128(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// (same instruction sequences as attn.hip: the bit-for-bit equivalence of the two kernels rests on them)
__device__ __forceinline__ void split2_pk(float p0, float p1, uint32_t& hp, uint32_t& lp) {
  float d0, d1;
  asm("s_nop 0\n\tv_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(p0), "v"(p1));
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(d0) : "v"(p0), "v"(hp));
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(d1) : "v"(p1), "v"(hp));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lp) : "v"(d0), "v"(d1));
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kPShift = 14.0f;

}  // namespace

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn64_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x >= p.main_blocks) { attn_tail_any<3>(p, (int)blockIdx.x - p.main_blocks, smem); return; }
  constexpr int NW = 4, QW = 64;                       // waves per workgroup, queries per wave
  constexpr int PART = 64 * 128;                       // 64 rows x 128 B
  constexpr int SLOT = PART * 2;                       // hi + lo
  constexpr int RD = 3;
  char* const Kring = smem;
  char* const Vring = smem + RD * SLOT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nqb = (p.Tq + QW * NW - 1) / (QW * NW);
  const int bid = xcd_remap(blockIdx.x, p.main_blocks);
  const int bh = bid / nqb, qb = bid - bh * nqb;
  const int b = bh / p.heads, head = bh - b * p.heads;
  const int T = p.T;
  const size_t qk_base = (size_t)bh * T * 64;
  const size_t v_base = (size_t)bh * 64 * p.Tpad;
  const int h = lane >> 5, r = lane & 31;

  // queries: sub-tile j of wave w holds the 32 queries the 8-wave kernel gives to its wave 2 w + j
  int q[2]; bool q_ok[2];
  half8 qh[2][4], ql[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    q[j] = qb * (QW * NW) + wave * QW + 32 * j + r;
    q_ok[j] = q[j] < p.Tq;
    if (!q_ok[j]) q[j] = T - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qh[j][s] = *(const half8*)(p.Qhi + qk_base + (size_t)q[j] * 64 + 16 * s + 8 * h);
      ql[j][s] = *(const half8*)(p.Qlo + qk_base + (size_t)q[j] * 64 + 16 * s + 8 * h);
    }
  }

  // DMA plan: a part is 8 wave-instructions of 8 rows x 128 B; each of the 4 waves issues 2 of them per part
  constexpr int PPW = 8 / NW;
  int drow[PPW], dch[PPW];
#pragma unroll
  for (int t = 0; t < PPW; ++t) {
    drow[t] = (wave + NW * t) * 8 + (lane >> 3);
    dch[t] = swz128(drow[t], lane & 7) * 8;
  }
  auto stageK = [&](int slot, int it) {
    char* base = Kring + slot * SLOT;
    const int k0 = it * 64;
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
      int key = k0 + drow[t]; if (key > T - 1) key = T - 1;
      const size_t o = qk_base + (size_t)key * 64 + dch[t];
      glds16(p.Khi + o, base + (wave + NW * t) * 1024);
      glds16(p.Klo + o, base + PART + (wave + NW * t) * 1024);
    }
  };
  auto stageV = [&](int slot, int it) {
    char* base = Vring + slot * SLOT;
    const int k0 = it * 64;
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
      const size_t o = v_base + (size_t)drow[t] * p.Tpad + k0 + dch[t];
      glds16(p.Vhi + o, base + (wave + NW * t) * 1024);
      glds16(p.Vlo + o, base + PART + (wave + NW * t) * 1024);
    }
  };
  // S^T of a 64-key tile for both query sub-tiles: every K fragment pair (hi, lo) is read once and used six times
  // One wave per SIMD: nothing hides the latency of an MFMA that waits for its own accumulator (a 32x32x16 issues every 32 cycles but
  // delivers after 64), so consecutive MFMAs always go to DIFFERENT accumulators - the four (sub-tile, key half) score tiles here,
  // the four (sub-tile, dh half) output tiles below. Per accumulator the order of the products is unchanged (bit-identical results).
  auto qk = [&](const char* kb, f32x16 (*s)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[j][i][e] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      half8 kh[2], kl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 32 * i + r;
        const int off = row * 128 + swz128(row, 2 * st + h) * 16;
        kh[i] = *(const half8*)(kb + off);
        kl[i] = *(const half8*)(kb + PART + off);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[i], qh[j][st], s[j][i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[i], ql[j][st], s[j][i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[i], qh[j][st], s[j][i], 0, 0, 0);
    }
  };

  f32x16 o0[2][2];                                     // [sub-tile][dh half]
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) o0[j][t][i] = 0.f;
  float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};

  const int nt = (T + 63) >> 6;
  stageK(0, 0);
  stageV(0, 0);
  if (nt > 1) stageK(1, 1);
  if (nt > 2) stageK(2, 2);
  if (nt > 1) stageV(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 sc[2][2], sn[2][2];                           // [sub-tile][key half]: this tile's and the next tile's scores
  qk(Kring, sc);
  __syncthreads();                                     // K slot 0 is free for tile 3 from here on

  auto tile = [&](int it, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    if (!LAST) {
      if (it + RD < nt) stageK(it % RD, it + RD);
      if (it + RD - 1 < nt) stageV((it + RD - 1) % RD, it + RD - 1);
      qk(Kring + ((it + 1) % RD) * SLOT, sn);            // next tile's scores (matrix pipe) beside this tile's softmax (vector pipe)
    }
    const int kbase = it * 64;
    u32x4 phu[2][2][2], plu[2][2][2];                    // [sub-tile][key half][8-key group]
    float alpha[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (LAST) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int key = kbase + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (key >= T) sc[j][i][e] = -1e30f;
          }
      }
      float mx = -1e30f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sc[j][i][e]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[j], mx);
      alpha[j] = __builtin_amdgcn_exp2f(m_run[j] - m_new);
      m_run[j] = m_new;
      const f32x2 shift2 = {kPShift - m_new, kPShift - m_new};
      f32x2 psum2 = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            f32x2 d = {sc[j][i][8 * sp + e], sc[j][i][8 * sp + e + 1]};
            d += shift2;
            f32x2 pv = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
            psum2 += pv;
            uint32_t hp_, lp_; split2_pk(pv[0], pv[1], hp_, lp_);
            phu[j][i][sp][e >> 1] = hp_; plu[j][i][sp][e >> 1] = lp_;
          }
        }
      const float psum = psum2[0] + psum2[1];
      l_run[j] = l_run[j] * alpha[j] + psum;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o0[j][t][e] *= alpha[j];
    }
    // ---- O^T += V^T P^T for both sub-tiles: every V^T fragment pair read once, used six times ----
    const char* vb = Vring + (it % RD) * SLOT;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        half8 vh[2], vl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int row = 32 * t + r;
          const int off = row * 128 + swz128(row, 4 * i + 2 * sp + h) * 16;
          vh[t] = *(const half8*)(vb + off);
          vl[t] = *(const half8*)(vb + PART + off);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            o0[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[t], __builtin_bit_cast(half8, phu[j][i][sp]), o0[j][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            o0[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[t], __builtin_bit_cast(half8, plu[j][i][sp]), o0[j][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            o0[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[t], __builtin_bit_cast(half8, phu[j][i][sp]), o0[j][t], 0, 0, 0);
      }
    if (!LAST) {
      // the next iteration needs K(it + 2) and V(it + 1), issued one iteration ago; what this iteration issued may stay in flight
      constexpr int PP = PPW * 2;                            // pieces per wave per staged tile (hi + lo)
      const int mine = ((it + RD < nt) ? PP : 0) + ((it + RD - 1 < nt) ? PP : 0);
      if (mine == 2 * PP) __builtin_amdgcn_s_waitcnt((2 * PP) | (7 << 4) | (15 << 8));
      else if (mine == PP) __builtin_amdgcn_s_waitcnt(PP | (7 << 4) | (15 << 8));
      else __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                          // raw barrier: __syncthreads() would drain vmcnt to 0 again
#pragma unroll
      for (int j = 0; j < 2; ++j) { sc[j][0] = sn[j][0]; sc[j][1] = sn[j][1]; }
    }
  };
  for (int it = 0; it + 1 < nt; ++it) tile(it, std::false_type{});
  tile(nt - 1, std::true_type{});

  // ---- normalise and store: lane holds query q[j], dh = 32t + 8g + 4h + {0..3} ----
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float l_tot = l_run[j] + __shfl_xor(l_run[j], 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_ok[j]) {
      const size_t orow = ((size_t)b * T + q[j]) * p.ldo + (p.o_il ? head * 128 : head * 64);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          half4 hv, lv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            half_t hh, ll; split_f16(o0[j][t][4 * g + e] * inv, hh, ll); hv[e] = hh; lv[e] = ll;
          }
          const int d = 32 * t + 8 * g + 4 * h;
          const int oc = p.o_il ? il_col(d) : d;
          *(half4*)(p.Ohi + orow + oc) = hv;
          if (p.Olo) *(half4*)(p.Olo + orow + oc) = lv;
        }
    }
  }
}

int launch_attention64(const AttnParams& pm, int tail_blocks, hipStream_t s) {
  static bool set = false;
  if (!set) {
    if (hipFuncSetAttribute((const void*)attn64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return OVM_ERR_HIP;
    set = true;
  }
  const int smem = 3 * 4 * 64 * 128;                   // two 3-slot rings of hi + lo tiles = 96 KiB
  hipLaunchKernelGGL(attn64_kernel, dim3(pm.main_blocks + tail_blocks), dim3(256), smem, s, pm);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

}  // namespace ovm
