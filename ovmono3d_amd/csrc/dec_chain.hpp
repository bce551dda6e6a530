// Row-chain kernels of the GroundingDINO decoder (dec_chain.hip): parameter block and launcher.
#pragma once
#include <hip/hip_runtime.h>
#include "common.hpp"

namespace ovm {

struct ChainLin { const half_t* w; const float* bias; int N, K, Kpad; };   // interleaved split image [Npad][Kpad/32][hi 32 | lo 32]
struct ChainLn { const float* g; const float* b; };

struct DecChainParams {
  int Q, D, T, heads, ffn;
  float eps;
  float* hs;                        // [Q][D] decoder state: chain B updates it in place
  const float* ref;                 // [Q][4] this layer's reference boxes (cx, cy, w, h)
  float* ref_next;                  // [Q][4] refined boxes for the next layer, or null (last layer)
  float* qpos;                      // [Q][D]  query position embedding (chain A writes, chain B reads)
  float* qk; float* v;              // [Q][2D] = [q | k], [Q][D]: operands of the self-attention (chain A writes)
  const float* ctx;                 // [Q][D] self-attention output (chain B reads)
  const float* tk; const float* tv; int ldt;      // this layer's text keys / values [T][ldt]
  const float* val; int ldv;                      // this layer's deformable value image [S][ldv]
  int L, P; int lh[8], lw[8], lstart[8];
  ChainLin ref0, ref1, sa_qk, sa_v, sa_out, ca_q, ca_out, offw, msda_out, fc1, fc2, bb0, bb1, bb2;
  ChainLn ln1, ln2, ln3, ln4;
};

bool dec_chain_supported(int D, int heads, int ffn, int L, int P, int T, int npass);
// part 0: chain A (before the query self-attention), part 1: chain B (after it)
int launch_dec_chain(const DecChainParams& p, int part, hipStream_t s);

}  // namespace ovm
