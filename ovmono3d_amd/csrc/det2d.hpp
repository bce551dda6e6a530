// RPN inference + 2D box head + Fast R-CNN inference (mode "B" of the path), device side.
#pragma once
#include <vector>
#include "kernels.hpp"

namespace ovm {

struct SplitPtr { half_t* hi; half_t* lo; };

struct Det2dWorkspace {
  // RPN
  SplitPtr rpn_t[3];        // conv3x3+ReLU output per level, fp16 split [B*H*W][C]
  float* rpn_o[3];          // [B*H*W][16] fp32: 3 objectness logits + 12 deltas
  float* lvl_scores;        // [B][3][pre_topk]
  float* lvl_boxes;         // [B][3][pre_topk][4]
  int* lvl_count;           // [B][3]
  float* prop_boxes;        // [B][post_topk][4]
  float* prop_scores;       // [B][post_topk]
  int* prop_bidx;           // [B][post_topk]
  int* prop_count;          // [B]
  unsigned long long* nms_mask;   // scratch bit matrix
  unsigned int* hist;       // radix-select scratch
  int* sel_idx; float* sel_key;    // candidate buffers
  int* sort_idx; float* sort_key;
  // box head
  float* cand_boxes; float* cand_scores; int* cand_cls; int* cand_row; int* cand_count;
  int* keep_flags;
  int maxB, maxR, pre_topk, post_topk, topk, num_classes, cand_cap;
};

struct Det2dModel {
  int npass, B, G, C, F, roiK, num_classes;
  SplitPtr rpad[3];
  const half_t *rpn_conv_hi, *rpn_conv_lo; const float* rpn_conv_bias;
  const half_t *rpn_out_hi, *rpn_out_lo; const float* rpn_out_bias;
  const half_t *fc1_hi, *fc1_lo; const float* fc1_bias;
  const half_t *fc2_hi, *fc2_lo; const float* fc2_bias;
  const half_t *out_hi, *out_lo; const float* out_bias;
  float anchor_sizes[3], anchor_ratios[3];
  int pre_topk, post_topk; float rpn_nms;
  float score_thresh, nms_thresh; int topk;
  const ImageMeta* meta;
  RoiParams roi;
  SplitPtr RF, H1, H2; float* HO;
};

int det2d_alloc(Det2dWorkspace* w, int B, int G, int C, int num_classes, int maxR, int pre_topk, int post_topk, int topk,
                std::vector<void*>* allocs);
int det2d_forward(const Det2dModel& m, Det2dWorkspace& w, float* boxes, float* scores, int* classes, int* image_idx,
                  float* scores_full, int* out_counts, hipStream_t s);
int launch_nms_single(const float* boxes, const float* scores, int n, float thresh, int* keep_idx, int* n_keep, hipStream_t s);

}  // namespace ovm
