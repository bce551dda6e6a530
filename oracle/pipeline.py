"""End-to-end RCNN3D.inference restatement (fp32, CPU).

Follows reference cubercnn/modeling/meta_arch/rcnn3d.py:79-117 (``RCNN3D.inference``),
detectron2 ``GeneralizedRCNN.preprocess_image`` / ``ImageList.from_tensors`` /
``detector_postprocess`` (not in the container), the oracle branch of
``ROIHeads3D.forward`` roi_heads.py:232-243, and ``instances_to_coco_json``
cubercnn/evaluation/omni3d_evaluation.py:1200-1252.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import clip_vit, heads, mae_vit, midas_vit, rpn, sam_vit, sfp, vit


@dataclass
class OracleParams:
    model_name: str = "vitl14"
    sam_window: int = 14                                            # tower 'sam': window side and the global-attention blocks
    sam_global: Sequence[int] = (2, 5, 8, 11)
    tower: str = "dinov2"                                           # 'dinov2' | 'clip' | 'mae' (build_*_backbone); for 'mae' depth = num_layers
    embed_dim: int = 1024
    depth: int = 24
    heads: int = 16
    square_pad: int = 896
    pixel_mean: Sequence[float] = (123.675, 116.280, 103.530)      # OVMono3D_dinov2_SFP.yaml:23
    pixel_std: Sequence[float] = (58.395, 57.120, 57.375)          # :24
    use_depth_fusion: bool = True
    strides: Sequence[int] = (7, 14, 28)
    anchor_sizes: Sequence[float] = (64.0, 256.0, 512.0)           # :35-36
    anchor_ratios: Sequence[float] = (0.5, 1.0, 2.0)               # Base.yaml:44
    rpn_pre_topk: int = 1000
    rpn_post_topk: int = 1000
    rpn_nms: float = 0.7
    score_thresh: float = 0.01                                      # Base.yaml:65
    nms_thresh: float = 0.5
    topk: int = 100                                                 # config.py:220
    virtual_focal: float = 512.0
    pooler_min_level: int = 2
    pooler_max_level: int = 4
    pooler_res: int = 7


def preprocess(batched_inputs: List[Dict], P: OracleParams) -> Tuple[torch.Tensor, List[Tuple[int, int]]]:
    """(x - mean)/std per channel in tensor channel order, zero-pad bottom/right to SxS."""
    mean = torch.tensor(P.pixel_mean, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(P.pixel_std, dtype=torch.float32).view(3, 1, 1)
    ims = [(b["image"].to(torch.float32) - mean) / std for b in batched_inputs]
    sizes = [(int(i.shape[1]), int(i.shape[2])) for i in ims]
    S = P.square_pad
    if S <= 0:
        mh, mw = max(s[0] for s in sizes), max(s[1] for s in sizes)
        d = P.strides[-1]
        S_h, S_w = (mh + d - 1) // d * d, (mw + d - 1) // d * d
    else:
        S_h = S_w = S
    out = torch.zeros(len(ims), 3, S_h, S_w, dtype=torch.float32)
    for i, im in enumerate(ims):
        assert im.shape[1] <= S_h and im.shape[2] <= S_w, "image larger than SQUARE_PAD canvas"
        out[i, :, : im.shape[1], : im.shape[2]] = im
    return out, sizes


def backbone(sd, images: torch.Tensor, P: OracleParams, prompt_depth=None) -> Dict[str, torch.Tensor]:
    if P.tower == "clip":
        # detectron2's SimpleFeaturePyramid.forward(x) has no depth argument: the fork's rcnn3d.py:97 call with one raises
        assert prompt_depth is None, "the CLIP backbone takes no prompt_depth (SURVEY.md 0.4)"
        return sfp.sfp4_forward(sd, clip_vit.clip_backbone_forward(sd, images, P.heads, P.depth))
    if P.tower == "sam":
        assert prompt_depth is None, "the SAM backbone takes no prompt_depth (SURVEY.md 0.4)"
        return sfp.sfp4_forward(sd, sam_vit.sam_backbone_forward(sd, images, P.heads, P.depth, P.sam_window, P.sam_global))
    if P.tower == "midas":
        assert prompt_depth is None, "the MiDaS backbone takes no prompt_depth (SURVEY.md 0.4)"
        return sfp.sfp4_forward(sd, midas_vit.midas_backbone_forward(sd, images, P.heads, P.depth))
    if P.tower == "mae":
        assert prompt_depth is None, "the MAE backbone takes no prompt_depth (SURVEY.md 0.4)"
        return sfp.sfp4_forward(sd, mae_vit.mae_backbone_forward(sd, images, P.heads, P.depth))
    dense = vit.dino_backbone_forward(sd, images, P.heads, P.depth, prompt_depth, P.use_depth_fusion)
    return sfp.sfp_forward(sd, dense)


def detector_postprocess(inst: Dict[str, torch.Tensor], image_size: Tuple[int, int], out_h: int, out_w: int):
    sx, sy = out_w / image_size[1], out_h / image_size[0]
    o = dict(inst)
    b = o["pred_boxes"].clone()
    b[:, 0::2] *= sx
    b[:, 1::2] *= sy
    b = rpn.clip_boxes(b, (out_h, out_w))
    o["pred_boxes"] = b
    keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
    return {k: v[keep] for k, v in o.items()}


def inference(sd, batched_inputs: List[Dict], P: OracleParams, prompt_depth: Optional[torch.Tensor] = None,
              given_boxes: Optional[List[Dict[str, torch.Tensor]]] = None, do_postprocess: bool = True,
              return_aux: bool = False):
    """rcnn3d.py:79-117. ``given_boxes`` stands for the 2D detections a GroundingDINO forward would
    supply to ``_forward_cube`` (roi_heads_gdino.py:155-170)."""
    images, sizes = preprocess(batched_inputs, P)
    ratios = [b["height"] / s[0] for b, s in zip(batched_inputs, sizes)]            # rcnn3d.py:92
    Ks = [torch.as_tensor(b["K"], dtype=torch.float32) for b in batched_inputs]      # :95
    feats_d = backbone(sd, images, P, prompt_depth)
    feats = [feats_d[k] for k in sorted(feats_d)]
    scales = [1.0 / s for s in P.strides]
    aux = {"features": feats_d}
    if given_boxes is not None:
        inst = [dict(g) for g in given_boxes]
    elif any("oracle2D" in b for b in batched_inputs):                               # :100-102
        inst = []
        for b, r in zip(batched_inputs, ratios):
            o = b["oracle2D"]                                                        # roi_heads.py:232-243
            d = dict(pred_boxes=o["gt_bbox2D"].to(torch.float32) / r, pred_classes=o["gt_classes"])
            d["scores"] = o["gt_scores"].to(torch.float32) if "gt_scores" in o else torch.ones_like(o["gt_classes"]).float()
            inst.append(d)
    else:                                                                            # :106-111
        props = rpn.rpn_inference(sd, feats, P.strides, P.anchor_sizes, P.anchor_ratios, sizes,
                                  P.rpn_pre_topk, P.rpn_post_topk, P.rpn_nms)
        aux["proposals"] = props
        inst = heads.forward_box(sd, feats, [p[0] for p in props], sizes, scales, P.score_thresh, P.nms_thresh,
                                 P.topk, min_level=P.pooler_min_level, max_level=P.pooler_max_level,
                                 pooler_res=P.pooler_res)
    aux["instances_2d"] = inst
    inst, cube_aux = heads.forward_cube(sd, feats, inst, Ks, sizes, ratios, scales, P.virtual_focal,
                                        P.pooler_min_level, P.pooler_max_level, P.pooler_res)
    aux["cube"] = cube_aux
    if do_postprocess:
        inst = [detector_postprocess(i, s, b.get("height", s[0]), b.get("width", s[1]))
                for i, s, b in zip(inst, sizes, batched_inputs)]
    return (inst, aux) if return_aux else inst


def instances_to_coco_json(inst: Dict[str, torch.Tensor], img_id) -> List[Dict]:
    """omni3d_evaluation.py:1200-1252 (XYXY->XYWH, per-detection dict)."""
    n = len(inst["scores"])
    if n == 0:
        return []
    b = inst["pred_boxes"].clone()
    b[:, 2] -= b[:, 0]
    b[:, 3] -= b[:, 1]
    res = []
    has3d = "pred_bbox3D" in inst
    for k in range(n):
        r = {"image_id": img_id, "category_id": int(inst["pred_classes"][k]), "bbox": b[k].tolist(),
             "score": float(inst["scores"][k])}
        if has3d:
            r["bbox3D"] = inst["pred_bbox3D"][k].tolist()
            r["center_cam"] = inst["pred_center_cam"][k].tolist()
            r["center_2D"] = inst["pred_center_2D"][k].tolist()
            r["dimensions"] = inst["pred_dimensions"][k].tolist()
            r["pose"] = inst["pred_pose"][k].tolist()
            r["depth"] = float(inst["pred_center_cam"][k][2])
        res.append(r)
    return res
