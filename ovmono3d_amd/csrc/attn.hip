// Fused (flash-style) multi-head attention for the ViT blocks, head dim 64, gfx950.
//
//   softmax(q k^T) v with q pre-scaled by dh^-0.5 * log2(e) in the QKV epilogue (probabilities = exp2) (dinov2 Attention.forward as the
//   reference reaches it at dino.py:89-90; the xFormers path of nohup.out:696-701 computes the same).
//
// Work split: one workgroup = NW waves (8 by default, 4 in co-run mode) = 32 NW queries of one (batch, head); each wave owns 32
// queries. K and V^T tiles of 64 keys arrive by global_load_lds (3-slot rings shared by the workgroup, source-side swizzle).
// Orientation ("key on the MFMA row"): S^T = K Q^T with v_mfma_f32_32x32x16_f16, so a lane holds one
// query column and 32 of the 64 keys in registers - softmax needs one cross-half shuffle per tile and
// the S^T accumulator is directly the B operand of O^T = V^T P^T (no LDS round trip for P).
// V^T is stored with the token order permuted inside groups of 16 (bits 2<->3) by the QKV epilogue so
// the PV A-fragment, whose k order is {4h..4h+3, 8+4h..8+4h+3}, is a single ds_read_b128.
//
// NPASS=3 uses split operands (hi + lo) for both products, three MFMAs per product into the same accumulator.
#include "kernels.hpp"
#include "attn_tail.hpp"
#include <type_traits>

namespace ovm {

__device__ __forceinline__ int swz128(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// Two probabilities -> packed fp16 hi parts and packed fp16 lo parts (lo = p - float(hi)) in four VALU instructions: packed convert,
// two mixed-precision FMAs that read the fp16 halves directly (v_fma_mix_f32: p * 1.0 - hi, exact), packed convert. The compiler's
// own code for `(half)p` / `(float)h` / `p - hf` / `(half)` spent seven (it converts to fp16 twice - scalar for the residual chain,
// packed for the stored value - and back once): 48 fewer VALU instructions per key tile and wave.
__device__ __forceinline__ void split2_pk(float p0, float p1, uint32_t& hp, uint32_t& lp) {
  float d0, d1;
  // p0 / p1 come straight out of v_exp_f32: the trans -> VALU wait state is the asm's own business (common.hpp, cvt_f16_rn)
  asm("s_nop 0\n\tv_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(p0), "v"(p1));
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(d0) : "v"(p0), "v"(hp));
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(d1) : "v"(p1), "v"(hp));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lp) : "v"(d0), "v"(d1));
}
__device__ __forceinline__ uint32_t cvt2_pk(float p0, float p1) {
  uint32_t hp;
  asm("s_nop 0\n\tv_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(p0), "v"(p1));
  return hp;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt_n() {
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

constexpr float kPShift = 14.0f;       // log2 of the scale the main kernel carries its probabilities at

// Software-pipelined over key tiles inside each wave: iteration t issues the S^T MFMAs of tile t+1, the
// softmax VALU work of tile t and the PV MFMAs of tile t as one basic block, so matrix and vector pipes
// overlap without relying on a partner wave. K tiles therefore run one tile ahead of V tiles (two
// LDS rings with a phase offset). Scores are in log2 units (Q is pre-scaled by dh^-0.5 * log2 e), so the
// probabilities are a bare v_exp_f32.
#ifndef OVM_ATTN_ILV
#define OVM_ATTN_ILV 0     // 1: consecutive MFMAs of a wave go to different accumulators instead of twelve in a row per accumulator - measured
                           // equal (263 vs 261 us per ViT-L launch, profiles/r03: back-to-back accumulation is forwarded by the matrix pipe), +13 VGPRs
#endif
#ifndef OVM_ATTN_RD
#define OVM_ATTN_RD 3      // 4 (three tiles in flight, 128 KB) measured no faster than 3
#endif
template <int NPASS, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x >= p.main_blocks) { attn_tail_any<NPASS>(p, (int)blockIdx.x - p.main_blocks, smem); return; }
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  // diagnostic (ovm_debug_set_ptr "attn_stamps" with the lock-step kernel): every workgroup records its start / end s_memtime and its XCC id
  const unsigned long long t_wg0 = p.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
  constexpr int PART = 64 * 128;                       // 64 rows x 128 B
  constexpr int SLOT = PART * ((NPASS == 3) ? 2 : 1);  // hi (+ lo)
  // Ring depth 3 for K and for V^T (96 KB: one workgroup per CU): two key tiles of LDS-DMA stay in flight and the end-of-tile
  // wait only covers the tile issued one iteration earlier.
  constexpr int RD = OVM_ATTN_RD;
  char* const Kring = smem;
  char* const Vring = smem + RD * SLOT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nqb = (p.Tq + 32 * NW - 1) / (32 * NW);
  const int bid = xcd_remap(blockIdx.x, p.main_blocks);
  const int bh = bid / nqb, qb = bid - bh * nqb;
  const int b = bh / p.heads, head = bh - b * p.heads;
  const int T = p.T;
  const size_t qk_base = (size_t)bh * T * 64;
  const size_t v_base = (size_t)bh * 64 * p.Tpad;
  const int h = lane >> 5, r = lane & 31;

  int q = qb * (32 * NW) + wave * 32 + r;
  const bool q_ok = q < p.Tq;
  if (!q_ok) q = T - 1;
  half8 qh[4], ql[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qh[s] = *(const half8*)(p.Qhi + qk_base + (size_t)q * 64 + 16 * s + 8 * h);
    if (NPASS == 3) ql[s] = *(const half8*)(p.Qlo + qk_base + (size_t)q * 64 + 16 * s + 8 * h);
  }

  // DMA plan: a part is 8 wave-instructions of 8 rows x 128 B; each wave issues 8 / NW of them per part
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
      if (NPASS == 3) glds16(p.Klo + o, base + PART + (wave + NW * t) * 1024);
    }
  };
  auto stageV = [&](int slot, int it) {
    char* base = Vring + slot * SLOT;
    const int k0 = it * 64;
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
      const size_t o = v_base + (size_t)drow[t] * p.Tpad + k0 + dch[t];
      glds16(p.Vhi + o, base + (wave + NW * t) * 1024);
      if (NPASS == 3) glds16(p.Vlo + o, base + PART + (wave + NW * t) * 1024);
    }
  };
  // S^T tile pair (64 keys x 32 queries) from a K slot
  auto qk = [&](const char* kb, f32x16* s) {
#if OVM_ATTN_ILV
    // consecutive MFMAs alternate between the two score tiles (per accumulator the order of the products is unchanged)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[i][e] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      half8 kh[2], kl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 32 * i + r;
        const int off = row * 128 + swz128(row, 2 * st + h) * 16;
        kh[i] = *(const half8*)(kb + off);
        if (NPASS == 3) kl[i] = *(const half8*)(kb + PART + off);
      }
      if (NPASS == 3) {
#pragma unroll
        for (int i = 0; i < 2; ++i) s[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[i], qh[st], s[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) s[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[i], ql[st], s[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) s[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[i], qh[st], s[i], 0, 0, 0);
    }
#else
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[i][e] = 0.f;
      const int row = 32 * i + r;
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int off = row * 128 + swz128(row, 2 * st + h) * 16;
        const half8 kh = *(const half8*)(kb + off);
        if (NPASS == 3) {
          const half8 kl = *(const half8*)(kb + PART + off);
          s[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[st], s[i], 0, 0, 0);
          s[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[st], s[i], 0, 0, 0);
        }
        s[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[st], s[i], 0, 0, 0);
      }
    }
#endif
  };

  f32x16 o0[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o0[t][i] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int nt = (T + 63) >> 6;
  stageK(0, 0);
  stageV(0, 0);
  if (nt > 1) stageK(1, 1);
  if (RD >= 3) {
    if (nt > 2) stageK(2, 2);
    if (nt > 1) stageV(1, 1);
  }
  if (RD >= 4) {
    if (nt > 3) stageK(3, 3);
    if (nt > 2) stageV(2, 2);
  }
  int prev_mine = 0;                                   // pieces this wave issued in the previous iteration (RD = 4)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 sc[2], sn[2];
  qk(Kring, sc);
  __syncthreads();                                     // K slot 0 is free for tile 2 from here on

  // One key tile. LAST (compile time) = the final tile: keys >= T are masked and no further tile is prefetched or scored;
  // every other tile is branch-free (no ragged-tile test, unconditional rescale). Besides removing two branches per tile
  // this lets the register allocator get by with 168 VGPRs instead of 238 (same kernel time), which leaves register room on
  // every SIMD for waves of the kernels that run concurrently on the GroundingDINO side stream.
  auto tile = [&](int it, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    if (!LAST) {
      // K(it + RD) goes into the slot of K(it) (scored during the previous iteration), V(it + RD - 1) into that of V(it - 1)
      if (it + RD < nt) stageK(it % RD, it + RD);
      if (it + RD - 1 < nt) stageV((it + RD - 1) % RD, it + RD - 1);
      qk(Kring + ((it + 1) % RD) * SLOT, sn);            // next tile's scores
    }
    const int kbase = it * 64;
    if (LAST) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kbase + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= T) sc[i][e] = -1e30f;
        }
    }
    float mx = -1e30f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sc[i][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    // packed fp32 arithmetic (v_pk_add_f32 handles two scores per lane and instruction) for the shift, the row sum and the
    // hi/lo residual; the exponentials and the converts have no packed form
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 shift2 = {kPShift - m_new, kPShift - m_new};
    f32x2 psum2 = {0.f, 0.f};
    u32x4 phu[2][2], plu[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          f32x2 d = {sc[i][8 * sp + e], sc[i][8 * sp + e + 1]};
          d += shift2;
          f32x2 pv = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
          psum2 += pv;
          // Probabilities are carried scaled by 2^kPShift (<= 16384, no fp16 overflow; the scale cancels in O / l). That moves
          // the fp16 subnormal range down to 3.7e-9 of the row maximum, so both parts can use the compiler's packed converts
          // (v_cvt_pk_f16_f32 flushes subnormal results): what a flush can drop is bounded by 4097 keys x 3.7e-9 (hi) and
          // 4097 x 7.6e-6 x 2^-11 (lo) of the largest term - 1.5e-5 each in the worst case.
          if (NPASS == 3) { uint32_t hp_, lp_; split2_pk(pv[0], pv[1], hp_, lp_); phu[i][sp][e >> 1] = hp_; plu[i][sp][e >> 1] = lp_; }
          else phu[i][sp][e >> 1] = cvt2_pk(pv[0], pv[1]);
        }
      }
    const float psum = psum2[0] + psum2[1];
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) o0[t][e] *= alpha;
    // ---- O^T += V^T P^T ----
    const char* vb = Vring + (it % RD) * SLOT;
#if OVM_ATTN_ILV
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
          if (NPASS == 3) vl[t] = *(const half8*)(vb + PART + off);
        }
        if (NPASS == 3) {
#pragma unroll
          for (int t = 0; t < 2; ++t) o0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[t], __builtin_bit_cast(half8, phu[i][sp]), o0[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < 2; ++t) o0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[t], __builtin_bit_cast(half8, plu[i][sp]), o0[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) o0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[t], __builtin_bit_cast(half8, phu[i][sp]), o0[t], 0, 0, 0);
      }
#else
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 32 * t + r;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          const int off = row * 128 + swz128(row, 4 * i + 2 * sp + h) * 16;
          const half8 vh = *(const half8*)(vb + off);
          if (NPASS == 3) {
            const half8 vl = *(const half8*)(vb + PART + off);
            o0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, __builtin_bit_cast(half8, phu[i][sp]), o0[t], 0, 0, 0);
            o0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, __builtin_bit_cast(half8, plu[i][sp]), o0[t], 0, 0, 0);
          }
          o0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, __builtin_bit_cast(half8, phu[i][sp]), o0[t], 0, 0, 0);
        }
    }
#endif
    if (!LAST) {
      if (RD >= 3) {
        // the next iteration needs K(it + 2) and V(it + 1), issued RD - 2 iterations ago; what was issued since may stay in flight
        constexpr int PP = PPW * ((NPASS == 3) ? 2 : 1);      // pieces per wave per staged tile
        const int mine = ((it + RD < nt) ? PP : 0) + ((it + RD - 1 < nt) ? PP : 0);
        const int allow = mine + ((RD >= 4) ? prev_mine : 0);
        prev_mine = mine;
        if (allow >= 4 * PP) __builtin_amdgcn_s_waitcnt((4 * PP) | (7 << 4) | (15 << 8));
        else if (allow == 3 * PP) __builtin_amdgcn_s_waitcnt((3 * PP) | (7 << 4) | (15 << 8));
        else if (allow == 2 * PP) __builtin_amdgcn_s_waitcnt((2 * PP) | (7 << 4) | (15 << 8));
        else if (allow == PP) __builtin_amdgcn_s_waitcnt(PP | (7 << 4) | (15 << 8));
        else __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // raw barrier: __syncthreads() would drain vmcnt to 0 again
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      sc[0] = sn[0];
      sc[1] = sn[1];
    }
  };
  for (int it = 0; it + 1 < nt; ++it) tile(it, std::false_type{});
  tile(nt - 1, std::true_type{});                      // masks nothing when T is a multiple of 64

  // ---- normalise and store: lane holds query q, dh = 32t + 8g + 4h + {0..3} ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
    const size_t orow = ((size_t)b * T + q) * p.ldo + (p.o_il ? head * 128 : head * 64);
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
        const int oc = p.o_il ? il_col(d) : d;
        *(half4*)(p.Ohi + orow + oc) = hv;
        if (p.Olo) *(half4*)(p.Olo + orow + oc) = lv;
      }
  }
  if (p.stamps && tid == 0) {
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p.stamps[3 * blockIdx.x] = t_wg0; p.stamps[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime(); p.stamps[3 * blockIdx.x + 2] = xcc;
  }
}

// ---------------------------------------------------------------------------------------------
// Two-wave-group variant (8 waves, split precision): the matrix work and the softmax of a key tile run in DIFFERENT barrier
// intervals, and the second wave group (waves 4-7) runs one interval behind the first. Waves w and w + 4 share a SIMD, so on
// every SIMD one wave issues 48 MFMAs (O^T += V^T P^T of tile t-1, then S^T = K Q^T of tile t: an M segment) while its partner
// does the ~250 VALU instructions of a softmax (an S segment) - the two pipes of a SIMD work at the same time by construction
// instead of by the instruction scheduler's luck inside one wave (PMC of the lock-step kernel above: MFMA busy 0.37, VALU 0.27).
//   G0: M(t) in interval 2t, S(t) in 2t+1        G1: M(t) in 2t+1, S(t) in 2t+2
// K / V^T tiles live in 4-slot rings (128 KiB). M(t) reads K(t) and V(t-1); both slots are free once G1's M(t) is over, and
// S(t) re-stages slot (t-1) % 4 of the K ring with K(t+3) and slot (t-2) % 4 of the V ring with V(t+2): at least two intervals
// after the last read, and three (G1) to four (G0) intervals before M(t+3) needs them. One counted vmcnt per tile, placed
// before the barrier in front of G0's M segment (G0: end of its S, two younger staging segments in flight; G1: end of its M, one).
// Per-lane arithmetic and accumulation order are those of attn_kernel: the results are bit-identical.
// ---------------------------------------------------------------------------------------------
template <int NPASS, int STAMP = 0>
__global__ __launch_bounds__(512, 1) void attn_pp_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = 8, RD = 4;
  if ((int)blockIdx.x >= p.main_blocks) { attn_tail_any<NPASS>(p, (int)blockIdx.x - p.main_blocks, smem); return; }
  static_assert(NPASS == 3, "split precision only");
  constexpr int PART = 64 * 128;
  constexpr int SLOT = 2 * PART;
  char* const Kring = smem;
  char* const Vring = smem + RD * SLOT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int nqb = (p.Tq + 32 * NW - 1) / (32 * NW);
  const int bid = xcd_remap(blockIdx.x, p.main_blocks);
  const int bh = bid / nqb, qb = bid - bh * nqb;
  const int b = bh / p.heads, head = bh - b * p.heads;
  const int T = p.T;
  const size_t qk_base = (size_t)bh * T * 64;
  const size_t v_base = (size_t)bh * 64 * p.Tpad;
  const int h = lane >> 5, r = lane & 31;

  int q = qb * (32 * NW) + wave * 32 + r;
  const bool q_ok = q < p.Tq;
  if (!q_ok) q = T - 1;
  half8 qh[4], ql[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qh[s] = *(const half8*)(p.Qhi + qk_base + (size_t)q * 64 + 16 * s + 8 * h);
    ql[s] = *(const half8*)(p.Qlo + qk_base + (size_t)q * 64 + 16 * s + 8 * h);
  }
  const int drow = wave * 8 + (lane >> 3);
  const int dch = swz128(drow, lane & 7) * 8;
  auto stageK = [&](int slot, int it) {
    char* base = Kring + slot * SLOT + wave * 1024;
    int key = it * 64 + drow; if (key > T - 1) key = T - 1;
    const size_t o = qk_base + (size_t)key * 64 + dch;
    glds16(p.Khi + o, base);
    glds16(p.Klo + o, base + PART);
  };
  auto stageV = [&](int slot, int it) {
    char* base = Vring + slot * SLOT + wave * 1024;
    const size_t o = v_base + (size_t)drow * p.Tpad + it * 64 + dch;
    glds16(p.Vhi + o, base);
    glds16(p.Vlo + o, base + PART);
  };
  f32x16 sc[2];
  f32x16 o0[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o0[t][i] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  half8 ph[2][2], pl[2][2];
  // Fragment offsets inside a 64-row slot part: row 32 j + r, 16-byte chunk c -> (32 j + r) * 128 + ((c ^ ((r >> 1) & 7)) << 4);
  // the row-block term 4096 j and the lo part (+ PART) are immediates of the ds_read
  int xk[4], xv[4];
  {
    const int x = (r >> 1) & 7;
#pragma unroll
    for (int st = 0; st < 4; ++st) xk[st] = r * 128 + (((2 * st + h) ^ x) << 4);          // K: chunk 2 st + h
#pragma unroll
    for (int c = 0; c < 4; ++c) xv[c] = r * 128 + (((4 * (c >> 1) + 2 * (c & 1) + h) ^ x) << 4);   // V^T: chunk 4 i + 2 sp + h, c = 2 i + sp
  }
  // M segment: 8 double steps of {4 ds_read_b128, 6 MFMAs}. A double step advances TWO independent accumulators by one
  // (hi, lo) fragment pair each - PV: O^T rows 0-31 and 32-63 (t = 0, 1; inside each the order i, sp of attn_kernel), then QK:
  // keys 0-31 and 32-63 (i = 0, 1; order st) - with their MFMAs interleaved a, b, a, b, a, b: dependent v_mfma_f32_32x32x16_f16
  // on one accumulator issue only every 64 cycles (stamps: 3100 cycles for a segment of 48 chained MFMAs), two interleaved chains
  // run at the 32-cycle issue rate. The reads of double step d + 1 are issued in front of the MFMAs of d (two register sets).
  // They are inline asm with hand-counted lgkmcnt waits (cdna_hip_programming.md 5.7, form iii): left to the compiler every wait
  // here came out as lgkmcnt(0) directly behind the youngest read. LDS reads return in order: before the MFMAs of d exactly the
  // four reads of d + 1 may be pending.
// (diagnostic builds: STAMP 2 drops the fragment reads, STAMP 3 the MFMAs - timing-only ablations, results are wrong)
#define OVM_DSR(DST, ADDR, OFF) do { if (STAMP != 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF)); \
                                     else asm volatile("" : "=v"(DST) : "v"(ADDR)); } while (0)
#define OVM_MF(A, B, C) (STAMP == 3 ? (C) : __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0))
#define OVM_LGKM(N) do { asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N)); __builtin_amdgcn_sched_barrier(0); } while (0)
  bool fine_stamp = false;                                      // diagnostic (STAMP 2): stamps inside one M segment
  unsigned long long* stamp_lds = (unsigned long long*)(smem + 2 * RD * SLOT) + wave * 128;
  auto m_segment = [&](unsigned vb, unsigned kb, auto qk_tag) {
    constexpr bool QK = decltype(qk_tag)::value;
    constexpr int ND = QK ? 8 : 4;
    unsigned va[4], ka[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { va[j] = vb + (unsigned)xv[j]; ka[j] = kb + (unsigned)xk[j]; }
    half8 fh[2][2], fl[2][2];                                  // [set][chain a / b]
    if (STAMP == 2 && fine_stamp && blockIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) stamp_lds[109] = t_; }
    // double step d < 4: V^T slot, chunk set d (= 2 i + sp), row blocks 0 (a) and 1 (b); d >= 4: K slot, chunk set d - 4 (= st)
#define OVM_RD(D)                                                                            \
    do {                                                                                     \
      if ((D) < 4) {                                                                         \
        OVM_DSR(fh[(D) & 1][0], va[(D) & 3], 0); OVM_DSR(fl[(D) & 1][0], va[(D) & 3], 8192);             \
        OVM_DSR(fh[(D) & 1][1], va[(D) & 3], 4096); OVM_DSR(fl[(D) & 1][1], va[(D) & 3], 4096 + 8192);   \
      } else {                                                                               \
        OVM_DSR(fh[(D) & 1][0], ka[(D) & 3], 0); OVM_DSR(fl[(D) & 1][0], ka[(D) & 3], 8192);             \
        OVM_DSR(fh[(D) & 1][1], ka[(D) & 3], 4096); OVM_DSR(fl[(D) & 1][1], ka[(D) & 3], 4096 + 8192);   \
      }                                                                                      \
    } while (0)
#define OVM_DSTEP(D)                                                                         \
    if ((D) < ND) {                                                                          \
      if ((D) + 1 < ND) { OVM_RD((D) + 1); OVM_LGKM(4); } else { OVM_LGKM(0); }              \
      const half8 ah_ = fh[(D) & 1][0], al_ = fl[(D) & 1][0], bh_ = fh[(D) & 1][1], bl_ = fl[(D) & 1][1];  \
      if ((D) < 4) {                                                                         \
        constexpr int i_ = ((D) >> 1) & 1, sp_ = (D) & 1;                                    \
        o0[0] = OVM_MF(al_, ph[i_][sp_], o0[0]);   \
        o0[1] = OVM_MF(bl_, ph[i_][sp_], o0[1]);   \
        o0[0] = OVM_MF(ah_, pl[i_][sp_], o0[0]);   \
        o0[1] = OVM_MF(bh_, pl[i_][sp_], o0[1]);   \
        o0[0] = OVM_MF(ah_, ph[i_][sp_], o0[0]);   \
        o0[1] = OVM_MF(bh_, ph[i_][sp_], o0[1]);   \
      } else {                                                                               \
        constexpr int st_ = (D) & 3;                                                         \
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   \
        sc[0] = OVM_MF(al_, qh[st_], st_ == 0 ? zero16 : sc[0]);  \
        sc[1] = OVM_MF(bl_, qh[st_], st_ == 0 ? zero16 : sc[1]);  \
        sc[0] = OVM_MF(ah_, ql[st_], sc[0]);       \
        sc[1] = OVM_MF(bh_, ql[st_], sc[1]);       \
        sc[0] = OVM_MF(ah_, qh[st_], sc[0]);       \
        sc[1] = OVM_MF(bh_, qh[st_], sc[1]);       \
      }                                                                                      \
      if (STAMP == 3) asm volatile("" ::"v"(ah_), "v"(al_), "v"(bh_), "v"(bl_));          \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      if (STAMP == 2 && fine_stamp && blockIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) stamp_lds[110 + (D)] = t_; } \
    }
    OVM_RD(0);
    OVM_DSTEP(0) OVM_DSTEP(1) OVM_DSTEP(2) OVM_DSTEP(3) OVM_DSTEP(4) OVM_DSTEP(5) OVM_DSTEP(6) OVM_DSTEP(7)
#undef OVM_DSTEP
#undef OVM_RD
  };
  // scores of the first tile (M(0) has no previous tile)
  auto qk_first = [&](const char* kb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[i][e] = 0.f;
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const char* a = kb + xk[st] + i * 4096;
        const half8 kh = *(const half8*)a;
        const half8 kl = *(const half8*)(a + PART);
        sc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[st], sc[i], 0, 0, 0);
        sc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[st], sc[i], 0, 0, 0);
        sc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[st], sc[i], 0, 0, 0);
      }
    }
  };
  // softmax of the scores in sc (tile it): probabilities into ph / pl, running max / sum, O rescaled
  auto softmax = [&](int it, bool last) {
    if (last) {
      const int kbase = it * 64;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kbase + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= T) sc[i][e] = -1e30f;
        }
    }
    float mx = -1e30f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sc[i][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 shift2 = {kPShift - m_new, kPShift - m_new};
    f32x2 psum2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          f32x2 d = {sc[i][8 * sp + e], sc[i][8 * sp + e + 1]};
          d += shift2;
          f32x2 pvv = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
          psum2 += pvv;
          u32x4 hu_ = __builtin_bit_cast(u32x4, ph[i][sp]), lu_ = __builtin_bit_cast(u32x4, pl[i][sp]);
          uint32_t hp_, lp_; split2_pk(pvv[0], pvv[1], hp_, lp_);
          hu_[e >> 1] = hp_; lu_[e >> 1] = lp_;
          ph[i][sp] = __builtin_bit_cast(half8, hu_); pl[i][sp] = __builtin_bit_cast(half8, lu_);
        }
      }
    const float psum = psum2[0] + psum2[1];
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) o0[t][e] *= alpha;
  };
  // STAMP = 1: diagnostic build (ovm_debug_set_ptr "attn_stamps"), workgroup 0 records s_memtime before and after every barrier
  int stamp_i = 0;
#define OVM_PSTAMP() do { if (STAMP && blockIdx.x == 0 && stamp_i < 100) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    if (lane == 0) stamp_lds[stamp_i] = t_; ++stamp_i; } } while (0)
#define OVM_PBAR() do { __builtin_amdgcn_sched_barrier(0); OVM_PSTAMP(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); OVM_PSTAMP(); } while (0)

  const int g_prio_mode = p.prio_mode;
  if (g_prio_mode == 2 && grp == 1) __builtin_amdgcn_s_setprio(1);   // experiment: static priority for the younger wave group
  const int nt = (T + 63) >> 6;
  // prologue: the first four K and V^T tiles
#pragma unroll
  for (int i = 0; i < RD; ++i) {
    if (i < nt) { stageK(i, i); stageV(i, i); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  OVM_PBAR();
  if (grp == 1) OVM_PBAR();                             // G1 runs one interval behind G0
  qk_first(Kring);                                      // M(0): scores of tile 0
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  OVM_PBAR();
  // one tile step: S(t), barrier, M(t+1), barrier. LAST (the final tile, peeled out of the loop so that the loop body has one
  // straight-line M segment and no phi copies of the score registers): masked ragged keys, PV only.
  auto tile_step = [&](int t, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    // ---- S(t)
    int issued = 0;                                     // pieces this wave issues in this segment (wave-uniform)
    if (!LAST) {
      if (t >= 1 && t - 1 + RD < nt) { stageK((t - 1) & (RD - 1), t - 1 + RD); issued += 2; }
      if (t >= 2 && t - 2 + RD < nt) { stageV((t - 2) & (RD - 1), t - 2 + RD); issued += 2; }
    }
    if (g_prio_mode == 1) __builtin_amdgcn_s_setprio(1);   // experiment: the VALU-heavy segment wins issue arbitration against the partner's MFMAs
    softmax(t, LAST);
    if (g_prio_mode == 1) __builtin_amdgcn_s_setprio(0);
    // the row sums / maxima are complete HERE: without this the optimiser sinks the 16-deep dependent v_pk_add chain of the row
    // sum (and its s_nops) below the barrier, behind the MFMAs of the M segment, where nothing overlaps it
    asm volatile("" : "+v"(l_run), "+v"(m_run));
    if (grp == 0) {
      // K(t+1) / V(t) were staged two S segments ago; what S(t-1) and S(t) issued may stay in flight
      const int prev = ((t >= 2 && t - 2 + RD < nt) ? 2 : 0) + ((t >= 3 && t - 3 + RD < nt) ? 2 : 0);   // issued by S(t-1)
      const int allow = issued + prev;
      if (allow >= 8) wait_vmcnt_n<8>(); else if (allow >= 6) wait_vmcnt_n<6>(); else if (allow >= 4) wait_vmcnt_n<4>();
      else if (allow >= 2) wait_vmcnt_n<2>(); else wait_vmcnt_n<0>();
    }
    OVM_PBAR();
    // ---- M(t+1): O^T += V^T(t) P^T(t), then the scores of tile t+1
    // LDS byte addresses: the dynamic segment starts at offset 0 (no static __shared__ in this kernel)
    fine_stamp = (t == 20);
    const unsigned vb_ = (unsigned)(RD * SLOT) + (unsigned)((t & (RD - 1)) * SLOT), kb_ = (unsigned)(((t + 1) & (RD - 1)) * SLOT);
    if (!LAST) m_segment(vb_, kb_, std::true_type{});
    else m_segment(vb_, kb_, std::false_type{});
    if (grp == 1) {
      // before the barrier in front of G0's M(t+2): K(t+2) / V(t+1) were staged in S(t-1); S(t) may stay in flight
      if (issued >= 4) wait_vmcnt_n<4>(); else if (issued >= 2) wait_vmcnt_n<2>(); else wait_vmcnt_n<0>();
    }
    OVM_PBAR();
  };
#pragma clang loop unroll(disable)
  for (int t = 0; t + 1 < nt; ++t) tile_step(t, std::false_type{});
  tile_step(nt - 1, std::true_type{});
  if (grp == 0) OVM_PBAR();                             // same number of barriers for every wave
#undef OVM_PBAR
  if (STAMP && blockIdx.x == 0 && p.stamps) {
    __syncthreads();
    if (lane == 0) { for (int i = 0; i < 126; ++i) p.stamps[wave * 128 + i] = (i < stamp_i || i >= 109) ? stamp_lds[i] : 0ull; p.stamps[wave * 128 + 127] = stamp_i; }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
    const size_t orow = ((size_t)b * T + q) * p.ldo + (p.o_il ? head * 128 : head * 64);
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
        const int oc = p.o_il ? il_col(d) : d;
        *(half4*)(p.Ohi + orow + oc) = hv;
        if (p.Olo) *(half4*)(p.Olo + orow + oc) = lv;
      }
  }
}

// The partials of a (head, leftover query) are combined by a tiny follow-up kernel (launch_attention enqueues it right behind the
// attention kernel): the kernel boundary is the release / acquire. An in-kernel "last arriver combines" needs a device-scope fence per
// workgroup, and on this chip that is an L2 write-back of everything the main workgroups have just stored - measured: no faster
// than the unsplit tail.
__global__ __launch_bounds__(64) void attn_tail_combine_kernel(const AttnParams p) {
  const int tq = blockIdx.x, tid = threadIdx.x;
  const int ntail = p.T - p.Tq;
  const int bh = tq / ntail, q = p.Tq + (tq - bh * ntail);
  const int b = bh / p.heads, head = bh - b * p.heads;
  const float* recs = p.tail_ws + (size_t)tq * kTailSplit * kTailRec;
  float M = -1e30f;
  for (int i = 0; i < kTailSplit; ++i) M = fmaxf(M, recs[i * kTailRec]);
  float Ltot = 0.f, o = 0.f;
  for (int i = 0; i < kTailSplit; ++i) {          // fixed order: deterministic
    const float w = __builtin_amdgcn_exp2f(recs[i * kTailRec] - M);
    Ltot = fmaf(recs[i * kTailRec + 1], w, Ltot);
    o = fmaf(recs[i * kTailRec + 4 + tid], w, o);
  }
  half_t hh, ll; split_f16(o / Ltot, hh, ll);
  const size_t oo = ((size_t)b * p.T + q) * p.ldo + (p.o_il ? il_col(head * 64 + tid) : head * 64 + tid);
  p.Ohi[oo] = hh;
  if (p.Olo) p.Olo[oo] = ll;
}

static int g_attn_tail = 1;
void attn_set_tail_rows(int on) { g_attn_tail = on; }
static int g_attn_tail_split = 1;   // leftover queries split over the keys when the caller provides a workspace (ovm_tune_set "attn_tail_split" 0: one workgroup each)
void attn_set_tail_split(int on) { g_attn_tail_split = on; }
size_t attn_tail_ws_floats(int B, int heads) { return (size_t)B * heads * 8 * kTailSplit * kTailRec; }
static int g_attn_lds_pad = 0;      // experiment: extra dynamic LDS per workgroup (lowers workgroups per CU)
void attn_set_lds_pad(int v) { g_attn_lds_pad = v; }
static unsigned long long* g_attn_stamps = nullptr;
void attn_set_stamps(unsigned long long* p) { g_attn_stamps = p; }
static int g_attn_prio = 0;
void attn_set_prio(int v) { g_attn_prio = v; }
static int g_attn_pp = 0;           // 1: two-wave-group kernel for the 8-wave split-precision case (ovm_tune_set "attn_pp"); measured equal
                                    // to the lock-step kernel (6.36 vs 6.23 ms per ViT-L image, same box), so it stays an option
void attn_set_pp(int v) { g_attn_pp = v; }
// 1: the 4-wave x 64-query kernel (attn64.hip) for split precision. Built in round 3 to halve the LDS fragment traffic per MFMA; on
// the chip it is SLOWER than 8 waves x 32 queries (339 vs 261 us per ViT-L launch at T = 4097, profiles/r03): with one wave per SIMD
// nothing overlaps the softmax VALU work or the fragment latency. It stays selectable (bit-identical results) as the record of that
// measurement; the 8-wave kernel remains the default.
static int g_attn_q64 = 0;
void attn_set_q64(int v) { g_attn_q64 = v; }
static int g_attn_waves = 0;        // 0 = automatic (8 by default, 4 in co-run mode)
void attn_set_waves(int v) { g_attn_waves = (v == 4 || v == 8) ? v : 0; }

int launch_attention(const AttnParams& p, int npass, hipStream_t s) {
  if (p.T <= 0 || p.B <= 0) return OVM_OK;
  if (p.Tpad % 64 != 0 || p.Tpad < ((p.T + 63) / 64) * 64) return OVM_ERR_SHAPE;
  if (npass == 3 && (!p.Qlo || !p.Klo || !p.Vlo)) return OVM_ERR_INVALID;
  AttnParams pm = p;
  pm.stamps = g_attn_stamps;
  pm.prio_mode = g_attn_prio;
  // Default: 8-wave workgroups - 256 queries share one K / V^T tile stream, i.e. half the LDS-DMA traffic of two 4-wave
  // workgroups per CU at the same 2 waves per SIMD (7.6 -> 6.7 ms per ViT-L image). Co-run mode (another stream's short kernels
  // run beside this launch): ONE 4-wave workgroup per CU; that kernel is slower by itself
  // (~9 ms) but leaves half of the wave slots free, which shortens the other stream's chain by more than it costs
  // (28.6 -> 27.5 ms per image end to end; 8 waves per CU in either arrangement do not).
  const int nw = g_attn_waves ? g_attn_waves : (p.corun ? 4 : 8);
  const int qpb = 32 * nw;
  const int tail = p.T % qpb;
  int tail_blocks = 0;
  pm.Tq = p.T;
  if (g_attn_tail && tail > 0 && tail <= 8 && p.T > qpb && (p.Tpad + 16) * 4 <= 2 * 2 * 64 * 128) {
    pm.Tq = p.T - tail;                            // leftover queries ride along as extra workgroups
    if (!g_attn_tail_split) { pm.tail_ws = nullptr; pm.tail_cnt = nullptr; }
    // with a workspace from the caller: kTailSplit workgroups per leftover query, each over a run of key tiles (attn_tail.hpp)
    tail_blocks = tail * p.heads * p.B * (pm.tail_ws ? kTailSplit : 1);
  } else { pm.tail_ws = nullptr; pm.tail_cnt = nullptr; }
  const int nqb = (pm.Tq + qpb - 1) / qpb;
  pm.main_blocks = nqb * p.heads * p.B;
  const dim3 grid(pm.main_blocks + tail_blocks), block(64 * nw);
  const int pad = g_attn_lds_pad;                  // experiment knob; the 3-slot rings (96 KB) already keep a workgroup alone on its CU
  if (npass == 3) {
    const int smem = OVM_ATTN_RD * 4 * 64 * 128 + pad;
    if (nw == 8 && g_attn_q64 && !g_attn_pp) {
      const int r64 = launch_attention64(pm, tail_blocks, s);
      if (r64 == OVM_OK && tail_blocks > 0 && pm.tail_ws) hipLaunchKernelGGL(attn_tail_combine_kernel, dim3(tail * p.heads * p.B), dim3(64), 0, s, pm);
      return r64;
    }
    if (nw == 8 && g_attn_pp) {
      constexpr int smem_pp = 2 * 4 * 2 * 64 * 128;        // two 4-slot rings of hi + lo tiles = 128 KiB
      static bool setpp = false;
      if (!setpp) { (void)hipFuncSetAttribute((const void*)attn_pp_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); setpp = true; }
#ifdef OVM_DIAG      // in-kernel s_memtime stamps and the timing-only ablations (STAMP 2 / 3 give WRONG results): diagnostic builds only
      if (pm.stamps) {
        (void)hipFuncSetAttribute((const void*)attn_pp_kernel<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_pp_kernel<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_pp_kernel<3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (g_attn_pp == 2) hipLaunchKernelGGL((attn_pp_kernel<3, 2>), grid, block, smem_pp + 8192, s, pm);
        else if (g_attn_pp == 3) hipLaunchKernelGGL((attn_pp_kernel<3, 3>), grid, block, smem_pp + 8192, s, pm);
        else hipLaunchKernelGGL((attn_pp_kernel<3, 1>), grid, block, smem_pp + 8192, s, pm);
      } else
#endif
      hipLaunchKernelGGL((attn_pp_kernel<3>), grid, block, smem_pp + pad, s, pm);
    } else if (nw == 8) {
      static bool set8 = false;
      if (!set8) { (void)hipFuncSetAttribute((const void*)attn_kernel<3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set8 = true; }
      hipLaunchKernelGGL((attn_kernel<3, 8>), grid, block, smem, s, pm);
    } else {
      static bool set4 = false;
      if (!set4 || pad > 0) { (void)hipFuncSetAttribute((const void*)attn_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set4 = true; }
      hipLaunchKernelGGL((attn_kernel<3, 4>), grid, block, smem, s, pm);
    }
  } else if (nw == 8) hipLaunchKernelGGL((attn_kernel<1, 8>), grid, block, OVM_ATTN_RD * 2 * 64 * 128, s, pm);
  else hipLaunchKernelGGL((attn_kernel<1, 4>), grid, block, OVM_ATTN_RD * 2 * 64 * 128, s, pm);
  if (tail_blocks > 0 && pm.tail_ws) hipLaunchKernelGGL(attn_tail_combine_kernel, dim3(tail * p.heads * p.B), dim3(64), 0, s, pm);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

}  // namespace ovm
