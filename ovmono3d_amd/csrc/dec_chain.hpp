// Row-chain kernels of the GroundingDINO decoder (dec_chain.hip): parameter block and launcher.
#pragma once
#include <hip/hip_runtime.h>
#include "common.hpp"

namespace ovm {

// w: the split weight image in MFMA-fragment order [Npad/16][Kpad/32][hi | lo][lane 0..63][8 halves] (gdino.hip: make_frag)
struct ChainLin { const half_t* w; const float* bias; int N, K, Kpad; };
struct ChainLn { const float* g; const float* b; };

struct DecChainParams {
  int Q, D, T, heads, ffn;
  float eps;
  float* hs;                        // [Q][D] decoder state: chain B updates it in place
  const float* ref;                 // [Q][4] this layer's reference boxes (cx, cy, w, h)
  const float* sine_dim_t;          // [D / 4]: 10000^(2 i / (D / 2)) for i = 0 .. D / 4 - 1 (sine embedding's frequency table)
  float* ref_next;                  // [Q][4] refined boxes for the next layer, or null (last layer)
  float* qpos;                      // [Q][D]  query position embedding (chain A writes, chain B reads)
  float* qk; float* v;              // [Q][2D] = [q | k], [Q][D]: operands of the self-attention (chain A writes)
  const float* ctx;                 // [Q][D] self-attention output (chain B reads)
  const float* tk; const float* tv; int ldt;      // this layer's text keys / values [T][ldt]
  const float* val; int ldv;                      // this layer's deformable value image [S][ldv]
  int L, P; int lh[8], lw[8], lstart[8];
  ChainLin ref0, ref1, sa_qk, sa_v, sa_out, ca_q, ca_out, offw, msda_out, fc1, fc2, bb0, bb1, bb2;
  ChainLn ln1, ln2, ln3, ln4;
  // FFN split (ffn_split > 1): chain B runs as a (row blocks) x (hidden chunks) grid - every workgroup of a row block repeats the
  // cheap front of the layer, computes ONE 512-column chunk of the hidden layer and writes its partial second-layer product; chain C
  // (one workgroup per row block) adds the partials in chunk order to (x + bias) - the order the single workgroup used - and finishes
  // the layer (LN4, state, box refinement). 57 workgroups streaming 4 MB of FFN weights each become 228 streaming 1 MB.
  int ffn_split;
  float* ffn_x;                     // [Q][D]  FFN input (= residual), written by the chunk-0 workgroups
  float* ffn_part;                  // [ffn_split][Q][D] partial products
  unsigned long long* dbg_stamps;   // -DOVM_DIAG: s_memtime stamps of workgroup 0 (chain A: [0..31], chain B: [32..95])
  int dbg_skip;                     // -DOVM_DIAG builds only (env OVM_DEC_CHAIN_SKIP): timing ablations, results are wrong when non-zero
};

bool dec_chain_supported(int D, int heads, int ffn, int L, int P, int T, int npass);
// part 0: chain A (before the query self-attention), part 1: chain B (after it), part 2: chain C (only with ffn_split > 1)
int launch_dec_chain(const DecChainParams& p, int part, hipStream_t s);

}  // namespace ovm
