"""ROIAlign (aligned=True, sampling_ratio=0), ROIPooler level assignment, NMS (fp32, CPU).

Third-party arithmetic reached from reference cubercnn/modeling/roi_heads/roi_heads.py:270,366
(``box_pooler`` / ``cube_pooler`` = detectron2 ``ROIPooler`` over torchvision ``roi_align``),
fast_rcnn.py:105 (``batched_nms``) and roi_heads_gdino.py:254 (``nms``). torchvision 0.19.1 and
detectron2 are not in the container; restated from their published algorithms (SURVEY.md
Appendix A5, A10). Pooler parameters: reference nohup.out:645-650 (7x7, scales 1/7,1/14,1/28,
sampling_ratio 0, aligned True).
"""
from __future__ import annotations

import math
from typing import List, Sequence

import torch


def _bilinear_axis(coord: torch.Tensor, size: int):
    """torchvision bilinear_interpolate index/weight rules along one axis.
    Returns (valid, low, high, w_low, w_high)."""
    valid = ~((coord < -1.0) | (coord > float(size)))
    c = coord.clamp(min=0.0)
    low = c.floor().to(torch.int64)
    at_edge = low >= size - 1
    low = torch.where(at_edge, torch.full_like(low, size - 1), low)
    high = torch.where(at_edge, low, low + 1)
    c = torch.where(at_edge, low.to(c.dtype), c)
    l = c - low.to(c.dtype)
    return valid, low, high, 1.0 - l, l


def roi_align_single(feat: torch.Tensor, box: torch.Tensor, scale: float, out: int = 7) -> torch.Tensor:
    """feat [C,H,W], box xyxy (image coords) -> [C,out,out]; aligned=True, sampling_ratio=0."""
    C, H, W = feat.shape
    f32 = torch.float32
    x1 = box[0].to(f32) * scale - 0.5
    y1 = box[1].to(f32) * scale - 0.5
    x2 = box[2].to(f32) * scale - 0.5
    y2 = box[3].to(f32) * scale - 0.5
    roi_w = x2 - x1
    roi_h = y2 - y1
    bin_h = roi_h / out
    bin_w = roi_w / out
    gh = int(math.ceil(float(roi_h) / out))
    gw = int(math.ceil(float(roi_w) / out))
    if gh <= 0 or gw <= 0:
        return torch.zeros(C, out, out, dtype=feat.dtype)
    count = max(gh * gw, 1)
    ph = torch.arange(out, dtype=f32).view(out, 1)
    iy = torch.arange(gh, dtype=f32).view(1, gh)
    ys = (y1 + ph * bin_h + (iy + 0.5) * bin_h / gh).reshape(-1)         # [out*gh]
    pw = torch.arange(out, dtype=f32).view(out, 1)
    ix = torch.arange(gw, dtype=f32).view(1, gw)
    xs = (x1 + pw * bin_w + (ix + 0.5) * bin_w / gw).reshape(-1)         # [out*gw]
    vy, yl, yh, hy, ly = _bilinear_axis(ys, H)
    vx, xl, xh, hx, lx = _bilinear_axis(xs, W)
    v = (feat[:, yl][:, :, xl] * (hy[:, None] * hx[None, :])
         + feat[:, yl][:, :, xh] * (hy[:, None] * lx[None, :])
         + feat[:, yh][:, :, xl] * (ly[:, None] * hx[None, :])
         + feat[:, yh][:, :, xh] * (ly[:, None] * lx[None, :]))
    v = v * (vy[:, None] & vx[None, :]).to(v.dtype)
    v = v.view(C, out, gh, out, gw).sum(dim=(2, 4)) / count
    return v


def assign_boxes_to_levels(boxes: torch.Tensor, min_level: int, max_level: int,
                           canonical_box_size: int = 224, canonical_level: int = 4) -> torch.Tensor:
    """detectron2 ``assign_boxes_to_levels``: floor(4 + log2(sqrt(area)/224 + 1e-8)) clamped."""
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    sizes = torch.sqrt(area)
    lvl = torch.floor(canonical_level + torch.log2(sizes / canonical_box_size + 1e-8))
    lvl = torch.clamp(lvl, min=min_level, max=max_level)
    return lvl.to(torch.int64) - min_level


def roi_pooler(features: Sequence[torch.Tensor], box_lists: List[torch.Tensor], scales: Sequence[float],
               out: int = 7, min_level: int = 2, max_level: int = 4) -> torch.Tensor:
    """detectron2 ``ROIPooler.forward``: features = per-level [B,C,H,W]; box_lists = per-image [n_i,4].
    Returns [sum n_i, C, out, out] in (image 0 boxes, image 1 boxes, ...) order."""
    C = features[0].shape[1]
    outs = []
    for b, boxes in enumerate(box_lists):
        if boxes.numel() == 0:
            continue
        boxes = boxes.to(torch.float32)
        if len(features) == 1:
            lv = torch.zeros(len(boxes), dtype=torch.int64)
        else:
            lv = assign_boxes_to_levels(boxes, min_level, max_level)
        for i in range(len(boxes)):
            l = int(lv[i])
            outs.append(roi_align_single(features[l][b], boxes[i], scales[l], out))
    if not outs:
        return torch.zeros(0, C, out, out)
    return torch.stack(outs)


def nms(boxes: torch.Tensor, scores: torch.Tensor, thr: float) -> torch.Tensor:
    """torchvision.ops.nms: greedy, score-descending, suppress IoU > thr (strict). Returns kept
    indices in decreasing-score order."""
    if boxes.numel() == 0:
        return torch.zeros(0, dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True)[1]
    b = boxes[order].to(torch.float32)
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    n = len(b)
    suppressed = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 >= n:
            break
        xx1 = torch.maximum(b[i, 0], b[i + 1:, 0])
        yy1 = torch.maximum(b[i, 1], b[i + 1:, 1])
        xx2 = torch.minimum(b[i, 2], b[i + 1:, 2])
        yy2 = torch.minimum(b[i, 3], b[i + 1:, 3])
        inter = (xx2 - xx1).clamp(min=0) * (yy2 - yy1).clamp(min=0)
        ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= ovr > thr
    return order[torch.as_tensor(keep, dtype=torch.int64)]


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, thr: float) -> torch.Tensor:
    """torchvision ``batched_nms`` (per-category NMS); kept indices sorted by decreasing score.
    Implemented as the per-class loop (``_batched_nms_vanilla``), equivalent to the coordinate-offset
    form up to ulp-level IoU ties."""
    if boxes.numel() == 0:
        return torch.zeros(0, dtype=torch.int64)
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for cid in torch.unique(idxs):
        cur = torch.where(idxs == cid)[0]
        k = nms(boxes[cur], scores[cur], thr)
        keep_mask[cur[k]] = True
    keep = torch.where(keep_mask)[0]
    return keep[torch.sort(scores[keep], descending=True, stable=True)[1]]
