"""RPN inference (fp32, CPU).

The reference's ``RPNWithIgnore`` (cubercnn/modeling/proposal_generator/rpn.py:19-39) overrides only
training methods; inference is detectron2's stock ``RPN.forward`` -> ``predict_proposals`` ->
``find_top_rpn_proposals`` (not in the container; restated from the published algorithm, SURVEY.md
Appendix A4). Head structure: reference nohup.out:632-639. Config: reference configs/Base.yaml:45-58,
configs/OVMono3D_dinov2_SFP.yaml:38-41 (sizes 64/256/512 on p2/p3/p4, ratios .5/1/2, offset 0,
pre/post NMS top-k 1000, NMS 0.7, min box size 0).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

from .roi_ops import batched_nms

SCALE_CLAMP = math.log(1000.0 / 16)


def cell_anchors(size: float, ratios: Sequence[float]) -> torch.Tensor:
    out = []
    area = size ** 2.0
    for r in ratios:
        w = math.sqrt(area / r)
        h = r * w
        out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(out, dtype=torch.float32)


def grid_anchors(H: int, W: int, stride: int, base: torch.Tensor, offset: float = 0.0) -> torch.Tensor:
    sx = torch.arange(offset * stride, W * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(offset * stride, H * stride, step=stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
    return (shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4)


def apply_deltas(deltas: torch.Tensor, boxes: torch.Tensor, weights=(1.0, 1.0, 1.0, 1.0)) -> torch.Tensor:
    """detectron2 ``Box2BoxTransform.apply_deltas``; deltas [N, k*4], boxes [N,4]."""
    deltas = deltas.float()
    boxes = boxes.to(deltas.dtype)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = deltas[:, 2::4] / ww
    dh = deltas[:, 3::4] / wh
    dw = torch.clamp(dw, max=SCALE_CLAMP)
    dh = torch.clamp(dh, max=SCALE_CLAMP)
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = torch.exp(dw) * widths[:, None]
    ph = torch.exp(dh) * heights[:, None]
    x1 = pcx - 0.5 * pw
    y1 = pcy - 0.5 * ph
    x2 = pcx + 0.5 * pw
    y2 = pcy + 0.5 * ph
    return torch.stack((x1, y1, x2, y2), dim=-1).reshape(deltas.shape)


def clip_boxes(b: torch.Tensor, hw: Tuple[int, int]) -> torch.Tensor:
    h, w = hw
    x1 = b[..., 0].clamp(min=0, max=w)
    y1 = b[..., 1].clamp(min=0, max=h)
    x2 = b[..., 2].clamp(min=0, max=w)
    y2 = b[..., 3].clamp(min=0, max=h)
    return torch.stack((x1, y1, x2, y2), dim=-1)


def rpn_head(sd: Dict[str, torch.Tensor], feats: List[torch.Tensor], prefix="proposal_generator.rpn_head."):
    logits, deltas = [], []
    for x in feats:
        t = F.relu(F.conv2d(x, sd[prefix + "conv.weight"], sd[prefix + "conv.bias"], padding=1))
        logits.append(F.conv2d(t, sd[prefix + "objectness_logits.weight"], sd[prefix + "objectness_logits.bias"]))
        deltas.append(F.conv2d(t, sd[prefix + "anchor_deltas.weight"], sd[prefix + "anchor_deltas.bias"]))
    return logits, deltas


def rpn_inference(sd, feats: List[torch.Tensor], strides: Sequence[int], sizes: Sequence[float],
                  ratios: Sequence[float], image_sizes: List[Tuple[int, int]], pre_topk=1000, post_topk=1000,
                  nms_thresh=0.7, min_box_size=0.0):
    """Returns per image (proposal_boxes [R,4], objectness_logits [R])."""
    logits, deltas = rpn_head(sd, feats)
    B = feats[0].shape[0]
    A = len(ratios)
    lvl_scores, lvl_boxes, lvl_ids = [], [], []
    for li, (lg, dl) in enumerate(zip(logits, deltas)):
        _, _, H, W = lg.shape
        anchors = grid_anchors(H, W, strides[li], cell_anchors(sizes[li], ratios))
        lg = lg.permute(0, 2, 3, 1).flatten(1)                                   # [B, HWA]
        dl = dl.view(B, A, 4, H, W).permute(0, 3, 4, 1, 2).flatten(1, -2)        # [B, HWA, 4]
        k = min(lg.shape[1], pre_topk)
        sc, idx = torch.sort(lg, dim=1, descending=True, stable=True)
        sc, idx = sc[:, :k], idx[:, :k]
        boxes = []
        for b in range(B):
            boxes.append(apply_deltas(dl[b][idx[b]], anchors[idx[b]]))
        lvl_scores.append(sc)
        lvl_boxes.append(torch.stack(boxes))
        lvl_ids.append(torch.full((k,), li, dtype=torch.int64))
    scores = torch.cat(lvl_scores, dim=1)
    boxes = torch.cat(lvl_boxes, dim=1)
    lvl = torch.cat(lvl_ids)
    results = []
    for b in range(B):
        bx, sc, lv = boxes[b], scores[b], lvl
        valid = torch.isfinite(bx).all(dim=1) & torch.isfinite(sc)
        if not valid.all():
            bx, sc, lv = bx[valid], sc[valid], lv[valid]
        bx = clip_boxes(bx, image_sizes[b])
        keep = ((bx[:, 2] - bx[:, 0]) > min_box_size) & ((bx[:, 3] - bx[:, 1]) > min_box_size)
        if int(keep.sum()) != len(bx):
            bx, sc, lv = bx[keep], sc[keep], lv[keep]
        keep = batched_nms(bx, sc, lv, nms_thresh)[:post_topk]
        results.append((bx[keep], sc[keep]))
    return results
