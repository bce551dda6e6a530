/*
 * libovm3d - C ABI of the MI355X-native OVMono3D-LIFT inference path.
 *
 * The reference (nightgoodl/ovmono3d) is pure Python and has no FFI; the entry points below are the
 * native equivalents of the Python plugin surface it exposes for this path. Each one cites the
 * reference interface it replaces. All pointers are plain pointers, all sizes plain integers; no
 * torch / C++ types cross the boundary. Device pointers are HIP device memory of the device the
 * handle was created on. Functions return 0 on success and a negative OVM_ERR_* code otherwise and
 * never throw; ovm_last_error() returns a message for the last failure on a handle.
 *
 * Threading: a handle is not thread-safe; use one handle per (device, stream). All work is
 * stream-ordered on the caller's stream; functions do not synchronise unless documented.
 * Ownership: the caller owns every input/output buffer; the handle owns packed weights + workspace
 * (sized at create for max_batch / max_rois) and allocates nothing on the hot path.
 */
#ifndef OVM3D_H
#define OVM3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OVM_OK 0
#define OVM_ERR_INVALID (-1)
#define OVM_ERR_HIP (-2)
#define OVM_ERR_MISSING_WEIGHT (-3)
#define OVM_ERR_SHAPE (-4)
#define OVM_ERR_CAPACITY (-5)
#define OVM_ERR_UNSUPPORTED (-6) /* a valid input outside the documented scope of the call (e.g. a CMYK JPEG) */

#define OVM_REC_FLOATS 48 /* detection record width, see OvmDet3D */

typedef struct OvmHandle OvmHandle;
typedef void* ovm_stream_t; /* hipStream_t */

/* Model / path configuration: the inference-relevant keys of the reference config tree
 * (cubercnn/config/config.py:80-92,118-239; configs/Base.yaml; configs/OVMono3D_dinov2_SFP.yaml). */
typedef struct OvmConfig {
  /* MODEL.DINO.MODEL_NAME table, reference cubercnn/modeling/backbone/dino.py:17-24 */
  int32_t embed_dim, depth, heads;
  int32_t pos_grid;         /* sqrt(#pretrained position embeddings) (37 for the 518px hub models) */
  int32_t canvas;           /* MODEL.FPN.SQUARE_PAD (OVMono3D_dinov2_SFP.yaml:37), multiple of 14 */
  int32_t fpn_channels;     /* MODEL.FPN.OUT_CHANNELS */
  int32_t use_depth_fusion; /* MODEL.DINO.USE_DEPTH_FUSION (config.py:92) */
  float pixel_mean[3];      /* MODEL.PIXEL_MEAN in tensor channel order */
  float pixel_std[3];
  /* ROI heads */
  int32_t num_classes;      /* MODEL.ROI_HEADS.NUM_CLASSES */
  int32_t fc_dim;           /* ROI_BOX_HEAD.FC_DIM / ROI_CUBE_HEAD.FC_DIM */
  int32_t pooler_res;       /* 7 */
  int32_t pooler_min_level, pooler_max_level; /* ROIPooler level clamp */
  float virtual_focal;      /* ROI_CUBE_HEAD.VIRTUAL_FOCAL */
  /* RPN + box head (reference configs/Base.yaml:45-66) */
  float anchor_sizes[4];    /* one size per pyramid level (3 levels: the 4th is unused) */
  float anchor_ratios[3];
  int32_t rpn_pre_topk, rpn_post_topk;
  float rpn_nms_thresh;
  float score_thresh, nms_thresh;
  int32_t detections_per_image;
  /* build-specific */
  int32_t precision;        /* 1 = fp16 operands, one MFMA pass; 3 = split fp16 (hi+lo), three passes */
  int32_t max_batch, max_rois;
  /* Which ViT feeds the simple feature pyramid (MODEL.BACKBONE.NAME):
   *   OVM_TOWER_DINOV2 (0)  build_dino_backbone, reference cubercnn/modeling/backbone/dino.py:17-153: hub DINOv2, patch 14,
   *                         LayerScale, erf-GELU, LN eps 1e-6, pos-embed bicubic with the +0.1 offset; pyramid scales (2, 1, 0.5)
   *                         -> p2..p4 at strides 7 / 14 / 28; checkpoint keys backbone.net.vit.*
   *   OVM_TOWER_CLIP   (1)  build_clip_backbone, reference cubercnn/modeling/backbone/clip.py:17-166: open_clip VisionTransformer
   *                         image tower (conv1 without bias, class_embedding, ln_pre, QuickGELU, LN eps 1e-5, no ln_post / proj),
   *                         patch 16, pos-embed resized with antialiased bicubic (:98-133); pyramid scales (4, 2, 1, 0.5)
   *                         -> p2..p5 at strides 4 / 8 / 16 / 32; checkpoint keys backbone.net.visual.*; prompt_depth refused
   *                         (detectron2's SimpleFeaturePyramid.forward takes no depth: SURVEY.md 0.4).
   *   OVM_TOWER_MAE    (2)  build_mae_backbone, reference cubercnn/modeling/backbone/mae.py:20-150: Hugging Face ViTMAE encoder
   *                         (patch 16 with bias, separate query / key / value linears, erf-GELU, LN eps 1e-12, no LayerScale),
   *                         position embeddings = fixed 2-D sin-cos table rebuilt for the canvas grid (:62-78,152-180), class token
   *                         + its (zero) position row, tap = hidden_states[depth] i.e. `depth` blocks are run - the reference taps
   *                         hidden_states[num_layers - 1], the state BEFORE the last block (:43-55,110-116), so depth = 11 for
   *                         vit-mae-base; same 4-level pyramid and prompt_depth rule as CLIP; keys backbone.net.vit.embeddings.*,
   *                         backbone.net.vit.encoder.layer.N.*
   *   OVM_TOWER_MIDAS  (3)  build_midas_backbone, reference cubercnn/modeling/backbone/midas_final.py:19-95: the ViT-L/16 of MiDaS
   *                         DPT_Large (timm vit_large_patch16_384: patch 16 with bias, class token, plain pre-norm blocks without
   *                         LayerScale, erf-GELU, LN eps 1e-6, norm_pre = identity, no final norm), position table of the 24 x 24
   *                         grid resized with the CLIP tower's antialiased bicubic (:64-66); checkpoint keys backbone.net.vit.*
   *                         in timm's naming (cls_token, pos_embed, patch_embed.proj, blocks.N.{norm1,attn.qkv,attn.proj,norm2,
   *                         mlp.fc1,mlp.fc2}); same 4-level pyramid and prompt_depth rule as CLIP
   *   OVM_TOWER_SAM    (4)  build_sam_backbone, reference cubercnn/modeling/backbone/sam.py:19-112: segment_anything's ImageEncoderViT
   *                         blocks (no class token; patch 16 with bias; absolute position table [grid][grid][D], bicubic-resized when
   *                         the canvas grid differs, :73-86; blocks with 14 x 14 windowed attention over the zero-padded grid except
   *                         the global ones; decomposed relative-position bias from the query content; erf-GELU, LN eps 1e-6), dense
   *                         output of the last block, the neck unused; keys backbone.net.vit.{pos_embed, patch_embed.proj,
   *                         blocks.N.{norm1, attn.qkv, attn.proj, attn.rel_pos_h, attn.rel_pos_w, norm2, mlp.lin1, mlp.lin2}};
   *                         pos_grid = the checkpoint's grid (64); sam_window / sam_global_mask below */
  int32_t tower;
  int32_t sam_window;        /* OVM_TOWER_SAM: window side of the windowed blocks (14) */
  uint32_t sam_global_mask;  /* OVM_TOWER_SAM: bit i set = block i attends globally (vit_b: blocks 2, 5, 8, 11) */
} OvmConfig;

enum { OVM_TOWER_DINOV2 = 0, OVM_TOWER_CLIP = 1, OVM_TOWER_MAE = 2, OVM_TOWER_MIDAS = 3, OVM_TOWER_SAM = 4 };

/* One host-resident fp32 tensor of a checkpoint, named with the reference state_dict key
 * (module tree printed at reference nohup.out:563-684; loaded at reference demo/demo.py:148). */
typedef struct OvmTensor {
  const char* name;
  const float* data;
  int32_t ndim;
  int64_t shape[4];
} OvmTensor;

/* One input image: the per-image dict of the reference (demo/demo.py:82-85, dataset_mapper.py:72):
 * 'image' uint8 at network resolution (any C/H/W element strides: CHW dict tensors and native NHWC both
 * work without a copy), 'height'/'width' (original), 'K'. */
typedef struct OvmImage {
  const uint8_t* data;      /* device pointer */
  int32_t height, width;    /* network resolution */
  int64_t stride_c, stride_h, stride_w;
  int32_t orig_height, orig_width;
  float K[9];
} OvmImage;

/* Detection record, 48 x 4 bytes (fields of detectron2 Instances as filled at reference
 * cubercnn/modeling/roi_heads/roi_heads.py:823-843 and consumed at omni3d_evaluation.py:1219-1249). */
typedef struct OvmDet3D {
  float box[4];        /* pred_boxes xyxy, original resolution (after detector_postprocess) */
  float score;         /* sqrt(score2d * exp(-uncertainty)) */
  int32_t category;    /* pred_classes */
  float bbox3D[24];    /* 8 corners x (x,y,z), camera space */
  float center_cam[3];
  float center_2D[2];  /* original-resolution pixels */
  float dimensions[3]; /* W, H, L */
  float pose[9];       /* row-major 3x3, egocentric */
  int32_t image;       /* index into the call's image array */
} OvmDet3D;

/* --- lifecycle ---------------------------------------------------------------------------------
 * Replaces build_model(cfg) + DetectionCheckpointer.resume_or_load (reference rcnn3d.py:252-276,
 * demo/demo.py:144-150): packs the named fp32 tensors into device-resident fp16(-split) GEMM layouts. */
int ovm_create(const OvmConfig* cfg, const OvmTensor* weights, int32_t n_weights, int32_t device, OvmHandle** out);
int ovm_destroy(OvmHandle* h);
const char* ovm_last_error(const OvmHandle* h);
const char* ovm_version(void);
/* sizeof() of a struct of this header as the library was compiled ("OvmConfig", "OvmTensor", "OvmImage", "OvmDet3D",
 * "OvmGdinoConfig"), -1 for an unknown name: lets a binding check its mirror of the layout before the first call. */
int ovm_abi_sizeof(const char* struct_name);

/* --- backbone: build_dino_backbone(...).forward(x, prompt_depth) -> {p2,p3,p4}
 * (reference dino.py:70-120,123-153,208-224; preprocess_image folded in, rcnn3d.py:88).
 * images: host array of B descriptors. prompt_depth: device fp32 [B][1][depth_h][depth_w] or NULL.
 * p2/p3/p4: device fp32 NHWC outputs [B][S/7][S/7][C], [B][S/14]..., [B][S/28]... or NULL to keep
 * the features only inside the handle (ovm_cube_forward / ovm_rpn_box_forward read them there). */
int ovm_backbone_forward(OvmHandle* h, const OvmImage* images, int32_t B, const float* prompt_depth,
                         int32_t depth_h, int32_t depth_w, float* p2, float* p3, float* p4, ovm_stream_t stream);

/* The pyramid the handle holds after ovm_backbone_forward, level by level (0 = finest, "p2"): number of levels (3 or 4 by
 * tower), and for one level its device pointer (fp32 NHWC [max_batch][side][side][fpn_channels], valid until the next forward
 * or ovm_destroy), side and stride in pixels. The 4-level towers' p5 is read this way. */
int ovm_backbone_num_levels(const OvmHandle* h);
int ovm_backbone_level(const OvmHandle* h, int32_t level, const float** data, int32_t* side, float* stride);

/* --- ROIHeads3D._forward_cube, eval branch (reference roi_heads.py:329-549,798-848) followed by
 * GeneralizedRCNN._postprocess (rcnn3d.py:115). Uses the features of the last ovm_backbone_forward.
 * boxes [n][4] xyxy at network resolution, scores [n], classes [n] int32, image_idx [n] int32 (sorted by
 * image), all device. out: device [n] records; out_counts: device int32 [B] kept detections per image
 * (records of empty post-processed boxes are dropped, order preserved). postprocess = 0 keeps every record
 * with its network-resolution box (RCNN3D.inference(do_postprocess=False), rcnn3d.py:113-117). */
int ovm_cube_forward(OvmHandle* h, const OvmImage* images, int32_t B, const float* boxes, const float* scores,
                     const int32_t* classes, const int32_t* image_idx, int32_t n, int32_t postprocess, OvmDet3D* out,
                     int32_t* out_counts, ovm_stream_t stream);

/* --- RPN inference + ROIHeads3D._forward_box + FastRCNNOutputs.inference
 * (reference rcnn3d.py:106, rpn.py:19-39 -> detectron2 RPN; roi_heads.py:252-296; fast_rcnn.py:57-143).
 * Outputs up to detections_per_image rows per image, image-major: boxes [B*topk][4] network res,
 * scores, classes, image_idx, and scores_full [B*topk][num_classes] (may be NULL); out_counts int32 [B]. */
int ovm_rpn_box_forward(OvmHandle* h, const OvmImage* images, int32_t B, float* boxes, float* scores,
                        int32_t* classes, int32_t* image_idx, float* scores_full, int32_t* out_counts,
                        ovm_stream_t stream);

/* --- detection gather over RCCL: replaces comm.gather(inference_json, dst=0)
 * (reference omni3d_evaluation.py:717-720). comm is an ncclComm_t. counts_all (host, world ints) is
 * filled on every rank; recv (device) must hold sum(counts_all) records on rank 0. */
int ovm_gather_records(void* comm, int32_t rank, int32_t world, const OvmDet3D* send, int32_t n_send,
                       OvmDet3D* recv, int32_t* counts_all, ovm_stream_t stream);

/* The counts exchange alone, so that rank 0 can size `recv` first: every rank calls it, then every rank calls
 * ovm_gather_records (which repeats the 4-byte exchange). counts_all: host, world ints, filled on every rank. */
int ovm_gather_counts(void* comm, int32_t rank, int32_t world, int32_t n_send, int32_t* counts_all, ovm_stream_t stream);

int ovm_comm_unique_id(uint8_t* id128);                                     /* ncclGetUniqueId */
int ovm_comm_init(const uint8_t* id128, int32_t rank, int32_t world, int32_t device, void** comm);
int ovm_comm_destroy(void* comm);

/* --- per-kernel timing with HIP events recorded on the stream the kernels are launched on (what
 * bench.py's roofline figure is computed from). Categories index the ms/launches arrays. */
#define OVM_PROF_ATTN 0
#define OVM_PROF_QKV 1
#define OVM_PROF_PROJ 2
#define OVM_PROF_FC1 3
#define OVM_PROF_FC2 4
#define OVM_PROF_LN 5
#define OVM_PROF_NCAT 6
/* Co-run mode: tells the handle that the caller runs other work on a second stream while ovm_backbone_forward executes (the
 * GroundingDINO detector of ROIHeads3DGDINO). The attention launches then keep to one workgroup per CU so that the other stream's
 * short kernels find free wave slots. Scheduling only: results are unchanged. */
int ovm_set_corun(OvmHandle* handle, int32_t on);

int ovm_profile_enable(OvmHandle* h, int32_t on); /* 0 off, 1 every category, else bit (c + 1) selects category c (attn, qkv, proj, fc1, fc2, ln) */
int ovm_profile_read(OvmHandle* h, float* ms /* [OVM_PROF_NCAT] */, int32_t* launches /* [OVM_PROF_NCAT] */);

/* --- host-side helpers (no GPU needed) -------------------------------------------------------- */
/* dinov2 interpolate_pos_encoding (hub: offset 0.1, bicubic, no antialias): pos [1+M*M][D] -> out [1+G*G][D] */
int ovm_host_interp_pos_embed(const float* pos, int32_t M, int32_t D, int32_t G, float* out);
/* resize_pos_embed of the CLIP tower (reference cubercnn/modeling/backbone/clip.py:98-133): F.interpolate(size=(G,G), bicubic,
 * align_corners=False, antialias=True) of the patch rows, class row kept: pos [1+M*M][D] -> out [1+G*G][D] */
int ovm_host_resize_pos_embed_aa(const float* pos, int32_t M, int32_t D, int32_t G, float* out);
/* get_2d_sincos_pos_embed(D, (G, G), add_cls_token=True) of the MAE tower (reference cubercnn/modeling/backbone/mae.py:152-180 over
 * transformers' get_2d_sincos_pos_embed_from_grid): out [1+G*G][D], row 0 zero; first D/2 columns encode the x coordinate, the
 * rest y, each as [sin | cos] over D/4 frequencies 10000^(-i/(D/4)); computed in double, stored fp32 */
int ovm_host_sincos_pos_embed(int32_t D, int32_t G, float* out);
/* InferenceSampler contiguous shard [begin,end) of rank (reference cubercnn/data/build.py:320) */
int ovm_host_shard_range(int64_t n_items, int32_t rank, int32_t world, int64_t* begin, int64_t* end);

/* --- kernel-level entry points (parity tests and micro-benchmarks call the same kernels the model
 * path launches). All pointers device; "split" fp16 tensors are a hi array and an optional lo array
 * (x ~= hi + lo). */
int ovm_op_split_f16(const float* x, int64_t n, uint16_t* hi, uint16_t* lo, ovm_stream_t stream);
int ovm_op_gemm(const uint16_t* a_hi, const uint16_t* a_lo, int32_t lda, const uint16_t* w_hi, const uint16_t* w_lo,
                int32_t M, int32_t N, int32_t K, const float* bias, int32_t relu, float* c, int32_t ldc,
                int32_t precision, ovm_stream_t stream);
/* hi, lo [rows][K] (K % 32 == 0) -> out [rows][K/32][hi 32 | lo 32]: the interleaved operand image of the split-precision GEMM.
 * In split mode ovm_op_gemm takes w_hi = such an image and w_lo = w_hi + 32; activations may be plain arrays or an image
 * (a_lo = a_hi + 32, lda = 2K). */
int ovm_op_interleave(const uint16_t* hi, const uint16_t* lo, int64_t rows, int32_t K, uint16_t* out, ovm_stream_t stream);
int ovm_op_layernorm(const float* x, int32_t M, int32_t D, const float* gamma, const float* beta, float eps,
                     float* y, ovm_stream_t stream);
int ovm_op_attention(const float* qkv, int32_t B, int32_t T, int32_t heads, float* out, int32_t precision,
                     ovm_stream_t stream);
int ovm_op_roi_align(const float* p2, const float* p3, const float* p4, const int32_t* hw /* [3][2] */,
                     const float* scales /* [3] */, int32_t C, int32_t out_res, int32_t min_level, int32_t max_level,
                     const float* boxes, const int32_t* image_idx, int32_t n, float* out /* [n][res*res*C] (ph,pw,c) */,
                     ovm_stream_t stream);
int ovm_op_cube_decode(const float* head13, int32_t ld, const float* boxes, const float* scores, const int32_t* classes,
                       const int32_t* image_idx, const OvmImage* images, int32_t B, int32_t n, float virtual_focal,
                       int32_t postprocess, OvmDet3D* rec, int32_t* keep, ovm_stream_t stream);
int ovm_op_nms(const float* boxes, const float* scores, int32_t n, float thresh, int32_t* keep_idx, int32_t* n_keep,
               ovm_stream_t stream);

/* --- GroundingDINO output glue of ROIHeads3DGDINO (reference roi_heads_gdino.py:186-202,236-263,266-294):
 * pred_logits [nq][ld] (pre-sigmoid token logits, ld = 256), pred_boxes [nq][4] cxcywh in [0,1] (device);
 * spans: host int32 [n_phrases][2] = [begin, end) token positions of each category phrase (walked from id 1,
 * +1 per separator, :277-291). Sigmoid, per-phrase SUM, max / first-argmax, strict `> box_threshold`, * [w,h,w,h],
 * cxcywh->xyxy, class-agnostic NMS. Outputs (device, capacity nq) in decreasing-score order; n_out device int32.
 * Synchronises the stream. */
int ovm_gdino_postprocess(const float* pred_logits, int32_t nq, int32_t ld, const float* pred_boxes, const int32_t* spans,
                          int32_t n_phrases, int32_t img_h, int32_t img_w, float box_threshold, float nms_threshold,
                          float* out_boxes, float* out_scores, int32_t* out_classes, int32_t* n_out, ovm_stream_t stream);

/* --- GroundingDINO network of ROIHeads3DGDINO as ONE call (ovm_gdino_forward). Replaces
 * `load_model("./configs/GroundingDINO_SwinB_cfg.py", "./checkpoints/groundingdino_swinb_cogcoor.pth")` and
 * `model(image[None], captions=[caption])` of reference cubercnn/modeling/roi_heads/roi_heads_gdino.py:16-23,186 (network:
 * IDEA-Research/GroundingDINO @856dde2, configs/GroundingDINO_SwinB_cfg.py:1-43). The handle owns the packed weights; per
 * (image size, caption) it builds a plan (index maps, masks, position tables, activation arena) once and replays the forward
 * as a single HIP graph afterwards. Weights are named as in the Hugging Face port (upstream checkpoints are renamed by the host,
 * ovmono3d_amd/gdino/detector.py:convert_upstream_state_dict). */
typedef struct OvmGdino OvmGdino;
typedef struct OvmGdinoConfig {
  int32_t d_model, enc_layers, dec_layers, heads, ffn_dim;        /* cfg:9-15: 256, 6, 6, 8, 2048 */
  int32_t n_levels, n_points, num_queries, max_text_len;          /* cfg:19-21,16,33: 4, 4, 900, 256 */
  float pe_temperature, eps;                                      /* cfg:4-6: 20; LayerNorm eps 1e-5 */
  int32_t bert_heads;                                             /* bert-base-uncased: 12 */
  int32_t swin_embed, swin_depths[4], swin_heads[4], swin_window; /* swin_B_384_22k (cfg:3): 128, 2/2/18/2, 4/8/16/32, 12 */
  float pixel_mean[3], pixel_std[3];                              /* MODEL.PIXEL_MEAN / STD in tensor channel order */
  int32_t flip_channels;                                          /* 1: the reference's images[0][[2,1,0]] (roi_heads_gdino.py:146) */
  int32_t precision;                                              /* 1 = fp16 operands, 3 = split fp16 (default) */
  int32_t use_graphs;                                             /* capture each plan's forward into a HIP graph */
  int32_t max_plans;                                              /* plans kept (LRU); 0 = 128 */
  int32_t plan_budget_mb;                                         /* device memory the kept plans may hold together (arenas, tables,
                                                                     split-K workspaces), MiB; least recently used plans go first;
                                                                     0 = 32768. A dataset has more aspect ratios than any fixed
                                                                     count: the bound that matters is bytes */
} OvmGdinoConfig;
int ovm_gdino_create(const OvmGdinoConfig* cfg, const OvmTensor* weights, int32_t n_weights, int32_t device, OvmGdino** out);
int ovm_gdino_destroy(OvmGdino* g);
const char* ovm_gdino_last_error(const OvmGdino* g);
/* image: uint8 at network resolution (device). token_ids: host int32 [ntok] = tokenizer(caption) incl. [CLS] / [SEP];
 * position_ids: host int32 [ntok] or NULL (upstream numbering: restart per phrase, delimiter included).
 * pred_logits: device fp32 [num_queries][max_text_len], pre-sigmoid, -inf beyond the caption; pred_boxes: device fp32
 * [num_queries][4] (cx, cy, w, h in [0, 1]). Either output may be NULL (results stay in the handle for ovm_gdino_detect). */
int ovm_gdino_forward(OvmGdino* g, const OvmImage* image, const int32_t* token_ids, int32_t ntok, const int32_t* position_ids,
                      float* pred_logits, float* pred_boxes, ovm_stream_t stream);
/* forward + the reference-owned output glue (ovm_gdino_postprocess below; roi_heads_gdino.py:186-202,236-263): outputs as there. */
int ovm_gdino_detect(OvmGdino* g, const OvmImage* image, const int32_t* token_ids, int32_t ntok, const int32_t* spans, int32_t n_phrases,
                     float box_threshold, float nms_threshold, float* out_boxes, float* out_scores, int32_t* out_classes, int32_t* n_out,
                     ovm_stream_t stream);
int32_t ovm_gdino_num_queries(const OvmGdino* g);
/* device pointers of the last forward's raw outputs (owned by the handle; valid until its next forward); logits_ld = max_text_len */
int ovm_gdino_last_outputs(OvmGdino* g, const float** pred_logits, const float** pred_boxes, int32_t* logits_ld);
/* tests: pin the two-stage top-k selection to the given device int32 [num_queries] (NULL: the network's own) */
int ovm_gdino_set_force_topk(OvmGdino* g, const int32_t* idx_device);
/* tests: copy an intermediate of the last forward ("bert_out", "text_features", "swin_stage1..3", "enc_vision", "enc_text",
 * "topk" (int32), "init_ref") into dst; returns the element count. name "launches": returns the kernel launches per forward. */
int64_t ovm_gdino_debug_copy(OvmGdino* g, const char* name, void* dst, int64_t capacity_elems, ovm_stream_t stream);

/* --- the whole path for one image as ONE call (SURVEY.md 8b `ovm_infer`): RCNN3D.inference with the text-prompted head, reference
 * cubercnn/modeling/meta_arch/rcnn3d.py:79-117 (batched_inputs = [{image, height, width, K, category_list}]) ->
 * roi_heads_gdino.py:93-171 -> roi_heads.py:329-549,798-848 -> detector_postprocess. `image` as for ovm_backbone_forward (orig
 * size and K filled); token_ids / spans as for ovm_gdino_detect (spans[k] = [begin, end) token positions of category k's phrase,
 * so the record's `category` is the index into the category list, roi_heads_gdino.py:162). out: device records, capacity
 * out_capacity (<= the detector's num_queries are produced); n_out: host. Synchronises the stream. */
int ovm_infer(OvmHandle* h, OvmGdino* g, const OvmImage* image, const int32_t* token_ids, int32_t ntok, const int32_t* spans,
              int32_t n_phrases, float box_threshold, float nms_threshold, OvmDet3D* out, int32_t out_capacity, int32_t* n_out,
              ovm_stream_t stream);

/* --- generic device ops the GroundingDINO branch (ROIHeads3DGDINO's network, reference roi_heads_gdino.py:186) is
 * was sequenced from in round 1 (still exported: unit tests and the Python-sequenced cross-check path use them): fp32
 * row-major tensors in HBM, one call per op, all arithmetic on the device. */
int ovm_g_pack_weight(const float* w, int32_t N, int32_t K, int32_t Kpad, uint16_t* hi, uint16_t* lo, ovm_stream_t stream);
int ovm_g_linear(const float* x, int32_t ldx, int32_t M, int32_t K, const uint16_t* w_hi, const uint16_t* w_lo, int32_t N, int32_t Kpad,
                 const float* bias, int32_t act /* 0 none, 1 relu, 2 gelu */, const float* residual, int32_t ldr, float* y, int32_t ldy,
                 int32_t precision, ovm_stream_t stream);
int ovm_g_layernorm(const float* x, const float* residual, int32_t M, int32_t D, const float* gamma, const float* beta, float eps, float* y,
                    ovm_stream_t stream);
int ovm_g_bmm(const float* a, const float* b, float* c, int32_t batch, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc,
              int64_t sA, int64_t sB, int64_t sC, int32_t transB, float alpha, ovm_stream_t stream);
int ovm_g_bmm2(const float* a, const float* b, float* c, int32_t nb1, int32_t nb2, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb,
               int32_t ldc, int64_t sA1, int64_t sB1, int64_t sC1, int64_t sA2, int64_t sB2, int64_t sC2, int32_t transB, float alpha,
               ovm_stream_t stream);
int ovm_g_softmax2(float* x, int32_t rows, int32_t cols, int32_t ld, const float* bias, int32_t bias_rows, int32_t bias_div, int32_t bias_ld,
                   const float* bias2, int32_t d2, int32_t m2, ovm_stream_t stream);
int ovm_g_softmax(float* x, int32_t rows, int32_t cols, int32_t ld, const float* bias, int32_t bias_rows, int32_t bias_div, int32_t bias_ld,
                  ovm_stream_t stream);
int ovm_g_eltwise(int32_t op, const float* a, const float* b, float* out, int64_t n, int64_t bmod, float alpha, float beta, ovm_stream_t stream);
int ovm_g_gather_rows(const float* src, int32_t ld_src, const int32_t* idx, int64_t n_out, int32_t nidx, int32_t cols, float* dst,
                      ovm_stream_t stream);
int ovm_g_groupnorm(const float* x, int32_t B, int32_t HW, int32_t C, int32_t groups, const float* gamma, const float* beta, float eps, float* y,
                    ovm_stream_t stream);
int ovm_g_msdeform(const float* value, const int32_t* shapes_hw, int32_t L, int32_t B, int32_t S, int32_t Q, int32_t H, int32_t dh, int32_t P,
                   const float* loc, const float* w, float* out, ovm_stream_t stream);
int ovm_g_sine_embed(const float* pos, int64_t n, int32_t nc, int32_t F, float temperature, float* out, ovm_stream_t stream);
int ovm_g_normalize_image(const OvmImage* image, const float* mean, const float* stdv, int32_t flip_channels, float* out_nhwc,
                          ovm_stream_t stream);
int ovm_g_rowmax(const float* x, int32_t rows, int32_t cols, int32_t ld, float* out, ovm_stream_t stream);
int ovm_g_topk(const float* scores, int32_t n, int32_t k, int32_t* out_idx, ovm_stream_t stream);

/* Process-global tuning knobs for experiments and tests (also settable as OVM_TUNE="key=value,..." when the host loads the
 * library). Defaults are the measured best; none changes results beyond fp32 summation order.
 *   gemm_bm 0|128|256, gemm_stages 0 (auto: wave-specialised kernel up to 512 tiles, symmetric 2-slot kernel above) |2|3|5|6,
 *   gemm_splitk 0|1, gemm_tail 0|1 (leftover rows as dot-product workgroups), attn_waves 0 (auto)|4|8, attn_lds_pad bytes,
 *   gemm256 0|1 (256 x 256 two-wave-group kernel for qkv / fc1), op_gemm256 n (ovm_op_gemm on that kernel, n = split-K hint),
 *   attn_tail 0|1, glin_small_max_tiles (-1 = heuristic), glin_target_blocks, glin_max_ksplit, glin_stages 1|2, gbmm_tiled 0|1,
 *   gemm256_n192 0|1 (qkv on 256 x 192 tiles where they fill the chip better; default 1), attn_q64 0|1 (the 4-wave x 64-query attention
 *   kernel; default 0: measured slower), attn_pp 0|1 (two-wave-group attention kernel), msdeform_vec 0|1 (vectorised deformable sampling;
 *   default 1), gdino_dec_chain 0|1 (GroundingDINO decoder layers as row-chain kernels; default 1; read when a plan is built),
 *   gdino_ffn_split 0|1 (decoder row chains: the FFN's 512-column chunks on separate workgroups + a finishing kernel; bit-identical, faster for the detector alone, not beside the ViT; default 0; read when a plan is built),
 *   gdino_swin_fused 0|1 (Swin blocks: qkv projection inside the window-attention kernel; default 1; read when a plan is built),
 *   gdino_gemm256 0|1 (the detector's wide K <= 256 contractions on the 256 x 256 kernel; default 1; read when a plan is built),
 *   attn_tail_split 0|1 (attention's leftover queries split over 16 key slices + combine kernel; default 1), attn_prio n (experiment:
 *   s_setprio inside the two-wave-group attention kernel; default 0), gemm256_ksplit n (experiment: split-K hint of the 256 x 256 kernel for
 *   proj / fc2; default 0 = off, measured slower), glin_wpe 2|4 (experiment: workgroups per CU the small fp32-A GEMM is compiled for; default 2),
 *   gdino_branches 0|1. Values that select timing-only ablations with wrong results exist in -DOVM_DIAG builds only. */
int ovm_tune_set(const char* key, int32_t value);
/* diagnostic hooks: device pointer for a named debug hook ("gemm256_stamps": u64 [8 waves][128] s_memtime stamps of workgroup 0;
 * "attn_stamps": -DOVM_DIAG builds only, OVM_ERR_UNSUPPORTED otherwise) */
int ovm_debug_set_ptr(const char* key, void* ptr);

/* --- introspection for tests: copy a named intermediate of the last forward into dst (device).
 * names: "tokens" [B*T][D] fp32, "p2" / "p3" / "p4" (/ "p5"); "rpn_boxes" [B][R][4], "rpn_scores" [B][R], "rpn_counts" [B] (int32 bits) =
 * the RPN's proposals after top-k / NMS of the last ovm_rpn_box_forward; "cube_head" [n][16] = the cube head's raw outputs of the
 * last ovm_cube_forward (deltas 2, dims 3, pose 6-D, depth 1, uncertainty 1). Returns the element count or a negative error. */
int64_t ovm_debug_copy(OvmHandle* h, const char* name, float* dst, int64_t capacity, ovm_stream_t stream);

/* ---- evaluation ("next" row 1 of SURVEY.md 8f) ---------------------------------------------------------------------------
 * Exact IoU of oriented 3D boxes, iou[i*M + j] for detection i and ground truth j; boxes are 8 corners x 3 floats in
 * pytorch3d's corner order. Replaces `box3d_overlap` -> pytorch3d `_C.iou_box3d` (cubercnn/evaluation/omni3d_evaluation.py:109-169)
 * including the screening of the detections: rows of non-coplanar (:68-87) or zero-area (:90-107) detections are 0.
 * `vol` (optional) receives the intersection volumes. */
int ovm_box3d_iou(const float* boxes_dt, const float* boxes_gt, int32_t N, int32_t M, float eps_coplanar, float eps_nonzero, float* iou,
                  float* vol, ovm_stream_t stream);

/* ---- data feeding ("next" row 2 of SURVEY.md 8f) ---------------------------------------------------------------------------
 * uint8 bilinear resize bit-identical to Pillow's Image.resize(size, BILINEAR), i.e. to detectron2's ResizeShortestEdge on
 * uint8 images (reference demo/demo.py:79-83, cubercnn/data/dataset_mapper.py:62-72). ovm_host_pil_bilinear_coeffs builds one
 * axis' window bounds and 22-bit integer weights on the host (call with null tables to get ksize); ovm_resize_bilinear_u8 runs
 * the horizontal then the vertical pass on the device. src strides are in elements over [H][W][C]. */
int ovm_host_pil_bilinear_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* coefs, int32_t coefs_capacity);
int ovm_resize_bilinear_u8(const uint8_t* src, int32_t H, int32_t W, int32_t C, int64_t sy, int64_t sx, int64_t sc, int32_t outH, int32_t outW,
                           const int32_t* xbounds, const int32_t* xcoefs, int32_t xksize, const int32_t* ybounds, const int32_t* ycoefs,
                           int32_t yksize, uint8_t* tmp, uint8_t* dst, ovm_stream_t stream);

/* fp32 bilinear resize, align_corners=False, no antialias: torch.nn.functional.interpolate(mode="bilinear") as the reference's
 * mapper applies it to a depth prompt (cubercnn/data/dataset_mapper.py:45-52 to the image size, :70-72 through ResizeShortestEdge
 * - detectron2's ResizeTransform takes this route for non-uint8 arrays). src [B][H][W] dense, dst [B][outH][outW], device. */
int ovm_resize_bilinear_f32(const float* src, int32_t B, int32_t H, int32_t W, int32_t outH, int32_t outW, float* dst, ovm_stream_t stream);

/* JPEG decode split at its only serial step. The reference reads images with cv2.imread (demo/demo.py:52) and detectron2's
 * read_image = Pillow (cubercnn/data/dataset_mapper.py:38), both libjpeg-turbo at its defaults (JDCT_ISLOW, fancy upsampling,
 * integer YCbCr -> RGB). ovm_host_jpeg_info walks the headers; ovm_host_jpeg_entropy_decode Huffman-decodes every scan on the
 * host into coefficient planes (int16 [coef_blocks][64] in natural order, component after component, each plane bw x bh blocks =
 * whole MCUs); ovm_jpeg_reconstruct dequantises, runs the 8 x 8 inverse DCT, upsamples the chroma and converts the colours on the
 * device into rgb [height][width][3] (planes: device scratch of coef_blocks * 64 bytes). Bit-identical to libjpeg-turbo / Pillow
 * `Image.open(f).convert("RGB")`. Scope: 8-bit Huffman JPEGs - baseline, extended-sequential and progressive (every scan of the
 * progression present) -, grey or 3 components with luma sampling 1x1 / 2x1 / 2x2; anything else (arithmetic coding, 12-bit, CMYK,
 * 4:4:0, an incomplete progression) -> OVM_ERR_UNSUPPORTED from the two host
 * calls, a corrupt stream -> OVM_ERR_INVALID. */
typedef struct OvmJpegInfo {
  int32_t width, height, ncomp;   /* ncomp 1 or 3 */
  int32_t hmax, vmax;             /* luma sampling factors (chroma is 1 x 1) */
  int32_t h[3], v[3];
  int32_t bw[3], bh[3];           /* coefficient plane of each component, in blocks */
  int32_t cw[3], ch[3];           /* component size in samples (ceil(width * h / hmax), ...) */
  int32_t qidx[3];
  int32_t colorspace;             /* 0 grey, 1 YCbCr, 2 RGB (no transform) */
  int32_t coef_blocks;            /* sum of bw * bh */
  uint16_t qt[4][64];             /* quantisation tables, natural order */
} OvmJpegInfo;
int ovm_host_jpeg_info(const uint8_t* data, size_t n, OvmJpegInfo* info);
int ovm_host_jpeg_entropy_decode(const uint8_t* data, size_t n, int16_t* coef, int64_t coef_capacity, OvmJpegInfo* info);
int ovm_jpeg_reconstruct(const int16_t* coef, const OvmJpegInfo* info, uint8_t* planes, uint8_t* rgb, ovm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OVM3D_H */
