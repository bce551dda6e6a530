// RPN inference + 2D box head + Fast R-CNN inference (mode "B" of the path), device side.
//
// Discrete steps (top-k, per-level / per-class NMS, final top-k) are done with one generic primitive,
// a batched bitonic sort of 64-bit composite keys in HBM/LDS, plus a bit-matrix NMS. Keys carry
// (group | inverted order-preserving score bits | candidate id), so ordering - including ties - is
// deterministic: higher score first, lower index first on exact ties.
#pragma once
#include <vector>
#include "kernels.hpp"

namespace ovm {

struct SplitPtr { half_t* hi; half_t* lo; };

struct Det2dWorkspace {
  int maxB, nlev, C, num_classes, R, pre_topk, topk;
  int HW[kMaxLevels], Wl[kMaxLevels];  // per level pixels / width
  int A_tot;                           // total anchors per image
  int Nrpn;                            // pow2 >= A_tot (sort length)
  int Ncand;                           // pow2 >= R * num_classes
  int Nmerge;                          // pow2 >= nlev * pre_topk
  SplitPtr rpn_t[kMaxLevels];          // conv3x3+ReLU output per level, fp16 split [B*HW][C]
  float* rpn_o[kMaxLevels];            // [B*HW][16] fp32: 3 objectness logits + 12 deltas
  unsigned long long* keys;            // [B][max(Nrpn, Ncand)]
  // RPN candidates, slot = l*pre_topk + i
  float* cbox; float* cscore; int* cgroup; int* cseg; int* ckeep;     // [B][nlev*pre_topk](x4)
  int* gstart; int* gend;              // [B][max(nlev, num_classes)]
  unsigned long long* mask;            // [B][Nmask][16]
  unsigned long long* mkeys;           // [B][Nmerge]
  float* prop_boxes; float* prop_scores; int* prop_bidx; int* prop_count;   // [B][R]
  // box head
  float* probs;                        // [B*R][64]
  float* dbox;                         // [B*R][num_classes][4] decoded + clipped
  float* sbox; int* sgroup; int* sseg; int* skeep;                    // [B][Ncand] gathered by sorted position
};

struct Det2dModel {
  int npass, B, nlev, C, F, roiK, num_classes;
  SplitPtr rpad[kMaxLevels]; float stride[kMaxLevels];
  const half_t *rpn_conv_hi, *rpn_conv_lo; const float* rpn_conv_bias;
  const half_t *rpn_out_hi, *rpn_out_lo; const float* rpn_out_bias;
  const half_t *fc1_hi, *fc1_lo; const float* fc1_bias;
  const half_t *fc2_hi, *fc2_lo; const float* fc2_bias;
  const half_t *out_hi, *out_lo; const float* out_bias;
  float anchor_sizes[kMaxLevels], anchor_ratios[3];
  int pre_topk, post_topk; float rpn_nms;
  float score_thresh, nms_thresh; int topk;
  const ImageMeta* meta;
  RoiParams roi;
  SplitPtr RF, H1, H2; float* HO;
};

int det2d_alloc(Det2dWorkspace* w, int B, int nlev, const int* sides, int C, int num_classes, int maxR, int pre_topk, int post_topk, int topk,
                std::vector<void*>* allocs);
int det2d_forward(const Det2dModel& m, Det2dWorkspace& w, float* boxes, float* scores, int* classes, int* image_idx,
                  float* scores_full, int* out_counts, hipStream_t s);
// Standalone class-agnostic NMS (torchvision.ops.nms semantics): keep_idx in decreasing-score order.
int launch_nms_single(const float* boxes, const float* scores, const int* valid, int n, float thresh, int* keep_idx, int* n_keep,
                      hipStream_t s);
int launch_topk(const float* scores, int n, int k, int* out_idx, hipStream_t s);
// GroundingDINO output glue without a host round trip: three launches on caller-owned scratch (gdino_post_ws_bytes; nq <= 2048), nothing
// allocated, nothing synchronised; up to kGdinoPostSpansByValue phrase spans travel as kernel arguments.
constexpr int kGdinoPostSpansByValue = 128;
size_t gdino_post_ws_bytes(int nq, int K);
int launch_gdino_post_ws(const float* logits, int nq, int ld, const float* cxcywh, const int* spans, int K, int img_h, int img_w,
                         float box_thr, float nms_thr, void* ws, size_t ws_bytes, float* out_boxes, float* out_scores, int* out_classes,
                         int* n_out, hipStream_t s);
int launch_gdino_post(const float* logits, int nq, int ld, const float* cxcywh, const int* spans, int K, int img_h, int img_w,
                      float box_thr, float nms_thr, float* out_boxes, float* out_scores, int* out_classes, int* n_out, hipStream_t s);

}  // namespace ovm
