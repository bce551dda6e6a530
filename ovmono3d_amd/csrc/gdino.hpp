// Kernels and launchers of the native GroundingDINO engine (scope row a10: the network ROIHeads3DGDINO calls at reference
// cubercnn/modeling/roi_heads/roi_heads_gdino.py:186). The network is sequenced in C++ (gdino.hip, ovm_gdino_forward); the fused
// kernels below replace the generic one-op-per-launch chain of round 1 (bmm + softmax + bmm, LayerNorm + add + gather + split).
#pragma once
#include "common.hpp"
#include "kernels.hpp"

namespace ovm {

// ---- row operator: gather -> (+ residual) -> LayerNorm -> (+ add) -> fp32 / split-fp16 outputs, one wave per row ------------
struct RowOpParams {
  const float* x; int ldx;            // source rows
  const int* idx; int nidx; int seg;  // optional gather: out row r = concat_j x[idx[r*nidx+j]][0:seg] (idx < 0: zeros); D = nidx*seg
  const float* res; int ldr;          // optional residual added before the norm (rows r)
  const float* gamma; const float* beta; float eps;   // optional LayerNorm over D (gamma == null: none)
  int zero_masked;                    // gather rows with idx < 0 stay zero AFTER the norm (Swin pads after layernorm_before)
  const float* add; int ld_add; int add_rows;         // optional y2 = y + add[r % add_rows]
  int M, D;
  float* y; int ldy;                  // fp32 y (or null)
  float* y2; int ldy2;                // fp32 y2 (or null)
  half_t* hi; half_t* lo; int ldh;    // split fp16 of y, zero filled up to ldh (or null)
  int il;                             // hi / lo form ONE interleaved image [row][k/32][hi 32 | lo 32] (lo = hi + 32, ldh = 2 D, D % 32 == 0): what gemm256 streams
  half_t* hi2; half_t* lo2; int ldh2; // split fp16 of y2
};
int launch_rowop(const RowOpParams& p, hipStream_t s);

// ---- attention with exact fp32 products on the matrix cores (v_mfma_f32_16x16x4_f32), flash-style over key chunks -----------
// o[b1][b2][q][:] = softmax_k(scale * q.k + bias_h[b2][q][k] + bias_b[b1][q][k]) v ; DH in {16, 32, 64}
struct AttnF32Params {
  const float *q, *k, *v; int ldq, ldk, ldv;
  long sq1, sq2, sk1, sk2, sv1, sv2;               // element strides of the two batch levels (b1 outer, b2 = head)
  float* o; int ldo; long so1, so2;                // fp32 out (or null)
  half_t* ohi; half_t* olo; int ldoh; long soh1, soh2;   // split fp16 out (or null)
  int nb1, nb2, Tq, Tk, DH; float scale;
  const float* bias_h; long sbh; int ldbh;         // [nb2][Tq][ldbh] or null
  const float* bias_b; long sbb; int ldbb;         // [nb1][Tq][ldbb] or null (sbb = 0: shared by all b1)
  // decomposed relative-position bias of the SAM image encoder (segment_anything add_decomposed_rel_pos): keys lie on a
  // rel_gh x rel_gw grid (key = kh * rel_gw + kw) and score(q, key) += rel_h[q][kh] + rel_w[q][kw]; both tables depend on the
  // query's content and are laid out [b1][q][b2][ldrel] (launch_relpos_tables writes them). null: none
  const float* rel_h; const float* rel_w; int rel_gw; int ldrel;
  int bias_vec;                                   // set by launch_attn_f32
};
int launch_attn_f32(const AttnF32Params& p, hipStream_t s);

// ---- Swin window attention with its qkv projection inside (round 3): one workgroup per (window, head), windows of 144 tokens, head
// dimension 32. [q | k | v] of the head = x_window W_head^T + b over split-fp16 rows straight from HBM (three-pass MFMA, weights in
// MFMA-fragment order) lands in LDS in the layouts attn_f32_kernel stages; then that kernel's score / softmax / value loop.
struct SwinQkvAttnParams {
  const half_t* xhi; const half_t* xlo; int ldx;   // window-partitioned LayerNorm rows [nW * 144][ldx] (planar split fp16)
  const half_t* wfrag; int KS;                     // qkv weight image in fragment order (make_frag), KS = Kpad / 32 k-steps
  const float* bias;                               // [3 C]
  int C, nh, nW;
  const float* relbias;                            // [nh][144][144]
  const float* mask;                               // [nW][144][144] or null
  half_t* ohi; half_t* olo; int ldo;               // context rows [nW * 144][ldo] (planar split fp16), head h at columns 32 h ..
  float scale;
};
bool swin_qkv_attn_supported(int C, int nh, int ws, int npass);
int launch_swin_qkv_attn(const SwinQkvAttnParams& p, hipStream_t s);

// rel_h[(m * H + h)][kh] = q[m][h*DH : (h+1)*DH] . Rh[qh - kh + gh - 1], rel_w likewise over kw with Rw and qw, for the rows m of
// fp32 q (row stride ldq); a row's position on its gh x gw attention grid is (t / gw, t % gw) with t = m % (gh * gw)
int launch_relpos_tables(const float* q, int ldq, int M, int H, int DH, int gh, int gw, const float* Rh, const float* Rw, float* rel_h,
                         float* rel_w, int ldrel, hipStream_t s);

// ---- bi-directional image <-> text attention of the fusion layer (4 heads x 256, T text tokens, S image tokens) ---------------
struct BiAttnParams {
  const float* qv; int ldq;           // [S][E] image queries  (vision_proj)
  const float* kt; int ldk;           // [T][E] text keys      (text_proj)
  const float* vv; int ldvv;          // [S][E] image values   (values_vision_proj)
  const float* vt; int ldvt;          // [T][E] text values    (values_text_proj)
  int S, T, H, dh; float scale;
  half_t* cv_hi; half_t* cv_lo; int ldcv;   // [S][E] image-side context (split fp16: A operand of out_vision_proj)
  float* cv;                          // or fp32 [S][E]
  float* ct;                          // [T][E] text-side context (fp32)
  float* sc;                          // workspace [H][T][S] raw scaled scores
  float* stat;                        // workspace [H][T][2] (max, sum)
  float* part;                        // workspace [nchunk][T][E]
  int chunk, nchunk;                  // S split for the text side
};
int launch_biattn(const BiAttnParams& p, hipStream_t s);      // 4 launches

// ---- multi-scale deformable attention with the softmax over (levels x points) and the sampling locations fused ----------------
struct MsDeformParams {
  const float* value; int ldv;        // [S][H*dh]
  const float* ow; int ldow;          // [Q][H*L*P*2 offsets | H*L*P logits]
  const float* ref; int ldref;        // mode 0 (encoder): [Q][2] reference point (x, y), offsets normalised by (W_l, H_l)
                                      // mode 1 (decoder): [Q][4] box (cx, cy, w, h): loc = c + off * wh * 0.5 / P
  int mode;
  int Q, H, dh, L, P;
  int lh[8], lw[8], lstart[8];
  float* out; int ldo;                // fp32 [Q][H*dh] (or null)
  half_t* ohi; half_t* olo; int ldoh; // split fp16 (or null)
};
int launch_msdeform_fused(const MsDeformParams& p, hipStream_t s);
void msdeform_set_vec(int v);

// ---- small element-wise helpers of the decoder ---------------------------------------------------------------------------------
// ref_out = sigmoid(delta + logit(clamp(ref, eps, 1 - eps)))  (iterative box refinement), [n][4]
int launch_box_refine(const float* delta, int ldd, const float* ref, float eps, float* out, int n, hipStream_t s);
// out[q][0:T] = x[q][0:T], out[q][T:ld] = -inf
int launch_pad_logits(const float* x, int ldx, int Q, int T, float* out, int ld, hipStream_t s);
// BERT embeddings: out[t] = LN(word[ids[t]] + pos[pids[t]] + type[0])
int launch_bert_embed(const float* word, const float* pos, const float* typ, const int* ids, const int* pids, int T, int D, const float* g,
                      const float* b, float eps, float* out, hipStream_t s);
// sigmoid(gather_rows(coord + prop_logit, idx)) for the two-stage selection: out[i][0:4]
int launch_select_ref(const float* coord, int ldc, const float* prop_logit, const int* idx, int n, float* out, hipStream_t s);
// top-k with a caller-owned key buffer (graph-safe variant of launch_topk): keys [pow2 >= n]
int launch_topk_keys(const float* scores, int n, int k, int* out_idx, unsigned long long* keys, int N, hipStream_t s);

}  // namespace ovm
