// Declarations of every kernel launcher of the path (definitions in the .hip files).
#pragma once
#include "common.hpp"
#include "gemm.hpp"

namespace ovm {

struct ImageDesc {             // one input image as the caller holds it in HBM
  const uint8_t* data;
  int H, W;                    // network-resolution size (after ResizeShortestEdge)
  int64_t sC, sH, sW;          // element strides (CHW: HW, W, 1; NHWC: 1, 3W, 3)
};

struct LnOut {
  half_t* hi; half_t* lo; int ld;   // fp16 split output (row stride ld), or null
  float* f32; int ldf;              // fp32 output, or null
  int padH, padW;                   // >0: hi/lo rows go to the interior of a zero-bordered NHWC image
  int il;                           // hi/lo form an interleaved image [row][k/32][hi 32 | lo 32] (lo = hi + 32, ld = 2 D)
};

struct AttnParams {
  const half_t *Qhi, *Qlo, *Khi, *Klo, *Vhi, *Vlo;   // Q,K [B][h][T][64]; V^T [B][h][64][Tpad] (permuted tokens)
  half_t *Ohi, *Olo; int ldo;                        // out rows b*T + t, column head*64 + d
  int o_il;                                          // output is an interleaved split image (Olo = Ohi + 32, ldo = 2 D)
  int B, heads, T, Tpad;
  int corun;                                         // another stream's short kernels run beside this launch: keep to one workgroup per CU
  int Tq, main_blocks;                               // set by the launcher: queries / workgroups of the tiled part
  int prio_mode;                                     // experiment knob of the two-wave-group kernel (ovm_tune_set "attn_prio")
  unsigned long long* stamps;                        // diagnostic build of the two-wave-group kernel: s_memtime per barrier, [8 waves][128]
  // leftover queries split over the keys (attn_tail.hpp): [B * heads * leftover][kTailSplit][68] floats and one counter per
  // (batch, head, leftover query), zero between launches; null = one workgroup per leftover query (rounds 1-2)
  float* tail_ws; int* tail_cnt;
};

constexpr int kMaxLevels = 4;   // pyramid levels: 3 (scales 2, 1, 0.5 - the DINOv2 tower) or 4 (4, 2, 1, 0.5 - CLIP / ViTDet style)

struct RoiParams {
  const float* feat[kMaxLevels]; int fh[kMaxLevels], fw[kMaxLevels]; float scale[kMaxLevels];   // NHWC fp32 levels
  int C, nlevels, min_level, max_level, out;                 // out = pooled resolution (7)
  const float* boxes;                                        // [n][4] xyxy network res
  const int* batch_idx;                                      // [n]
  int n;
  half_t *Ohi, *Olo; int ldo;                                // [n][out*out*C], (ph, pw, c) order
};

struct ImageMeta {             // per image, device resident
  float K[9];
  float ratio;                 // im_scales_ratio = orig_h / net_h     (rcnn3d.py:92)
  int net_h, net_w, orig_h, orig_w;
};

constexpr int kRecFloats = 48;
// Detection record (48 x 4 bytes): [0:4] box xyxy (original res after postprocess), [4] fused score,
// [5] class (int32 bits), [6:30] bbox3D 8x3, [30:33] center_cam, [33:35] center_2D, [35:38] dims (W,H,L),
// [38:47] pose 3x3, [47] image index (int32 bits)

struct CubeDecodeParams {
  const float* head;  int ldh;     // [n][>=13]: deltas(2) dims(3) pose6(6) z(1) uncert(1)
  const float* boxes;              // [n][4] network res
  const float* scores;             // [n] 2D scores
  const int* classes;              // [n]
  const int* batch_idx;            // [n]
  const ImageMeta* meta;           // [B]
  int n; float virtual_focal;
  int postprocess;                 // 1: detector_postprocess (rescale+clip+non-empty); 0: keep network-res boxes
  float* rec;                      // [n][48]
  int* keep;                       // [n] 1 if the post-processed 2D box is non-empty
};

int launch_patch_gather(const ImageDesc* d_imgs, int B, int G, int patch, int Kpad, const float* mean, const float* stdv,
                        half_t* Ahi, half_t* Alo, hipStream_t s);
int launch_ln_gelu_split(half_t* hi, half_t* lo, int M, int D, const float* gamma, const float* beta, float eps, hipStream_t s);
int launch_cls_init(float* X, const float* cls, const float* pos, int B, int T, int D, hipStream_t s);
int launch_ln_rows(const float* X, int ldx, int M, int D, const float* gamma, const float* beta, float eps,
                   const LnOut& o, hipStream_t s);
int launch_tokens_cast(const float* X, int B, int T, int G2, int D, int ldo, const float* depth_tok,
                       half_t* Ohi, half_t* Olo, hipStream_t s);
int launch_tokens_writeback(float* X, const float* F, int B, int T, int G2, int D, hipStream_t s);
int launch_maxpool2(const half_t* Ihi, const half_t* Ilo, int B, int G, int D, half_t* Ohi, half_t* Olo, hipStream_t s);
int launch_depth_resize(const float* Dp, int B, int Hd, int Wd, int G, float* out, hipStream_t s);
int launch_resize_bilinear_f32(const float* src, int B, int Hd, int Wd, int Ho, int Wo, float* out, hipStream_t s);
int launch_zero(void* p, size_t bytes, hipStream_t s);
int launch_attention(const AttnParams& p, int npass, hipStream_t s);
// attn64.hip: the 4-wave x 64-query form of the split-precision kernel (pm: Tq / main_blocks already set by launch_attention)
int launch_attention64(const AttnParams& pm, int tail_blocks, hipStream_t s);
void attn_set_q64(int v);
void attn_set_tail_split(int on);
// floats of AttnParams::tail_ws for up to 8 leftover queries per (batch, head); tail_cnt needs B * heads * 8 ints, zero-initialised once
size_t attn_tail_ws_floats(int B, int heads);
void attn_set_tail_rows(int on);
void attn_set_lds_pad(int v);
void attn_set_waves(int v);
void attn_set_pp(int v);
void attn_set_prio(int v);
void attn_set_stamps(unsigned long long* p);
int launch_roi_align(const RoiParams& p, hipStream_t s);
int launch_cube_decode(const CubeDecodeParams& p, hipStream_t s);
int launch_compact_records(const float* rec, const int* keep, int n, int B, float* out, int* counts, hipStream_t s);

}  // namespace ovm
