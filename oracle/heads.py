"""2D box head, Fast R-CNN inference, cube head and cube decode (fp32, CPU).

Reference-owned arithmetic, restated line by line:
  * ``ROIHeads3D._forward_box``            cubercnn/modeling/roi_heads/roi_heads.py:252-296
  * ``fast_rcnn_inference_single_image``   cubercnn/modeling/roi_heads/fast_rcnn.py:57-116
  * ``CubeHead.forward``                   cubercnn/modeling/roi_heads/cube_head.py:148-204
  * ``ROIHeads3D._forward_cube`` (eval)    cubercnn/modeling/roi_heads/roi_heads.py:329-549, :798-848
  * ``compute_virtual_scale_from_focal_spaces`` cubercnn/util/math_util.py:581-592
  * ``R_from_allocentric`` (tensor branch) cubercnn/util/math_util.py:651-679
  * ``get_cuboid_verts_faces``             cubercnn/util/math_util.py:116-219
Third-party (pytorch3d @055ab3a, not in the container; restated from the published functions):
``rotation_6d_to_matrix`` (cube_head.py:177), ``axis_angle_to_matrix`` (math_util.py:676).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

from .roi_ops import batched_nms, roi_pooler
from .rpn import apply_deltas, clip_boxes


# ----------------------------------------------------------------------------- pytorch3d bits
def rotation_6d_to_matrix(d6: torch.Tensor) -> torch.Tensor:
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def axis_angle_to_matrix(axis_angle: torch.Tensor) -> torch.Tensor:
    """pytorch3d quaternion route: axis_angle_to_quaternion -> quaternion_to_matrix."""
    angles = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    small = angles.abs() < 1e-6
    s = torch.empty_like(angles)
    s[~small] = torch.sin(half[~small]) / angles[~small]
    s[small] = 0.5 - (angles[small] * angles[small]) / 48
    q = torch.cat([torch.cos(half), axis_angle * s], dim=-1)
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


# ----------------------------------------------------------------------------- math_util bits
def compute_virtual_scale_from_focal_spaces(f, H, f0, H0):
    return (H0 * f) / (f0 * H)                                   # math_util.py:592


def R_from_allocentric(K: torch.Tensor, R_view: torch.Tensor, u: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    fx, fy, sx, sy = K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]       # math_util.py:658-661
    oray = torch.stack(((u - sx) / fx, (v - sy) / fy, torch.ones_like(u))).T
    oray = oray / torch.linalg.norm(oray, dim=1).unsqueeze(1)
    angle = torch.acos(oray[:, -1])
    axis = torch.zeros_like(oray)
    axis[:, 0] = axis[:, 0] - oray[:, 1]
    axis[:, 1] = axis[:, 1] + oray[:, 0]
    norms = torch.linalg.norm(axis, dim=1)
    valid = angle > 0
    M = axis_angle_to_matrix(angle.unsqueeze(1) * axis / norms.unsqueeze(1))
    R = R_view.clone()
    R[valid] = torch.bmm(M[valid], R_view[valid])                          # math_util.py:678-679
    return R


def get_cuboid_verts(box3d: torch.Tensor, R: torch.Tensor) -> torch.Tensor:
    """box3d [n,6] = X Y Z W H L; returns [n,8,3] (math_util.py:116-196)."""
    n = len(box3d)
    x3d, y3d, z3d, w3d, h3d, l3d = [box3d[:, i].unsqueeze(1) for i in range(6)]
    verts = torch.zeros(n, 3, 8, dtype=torch.float32)
    verts[:, 0, [0, 3, 4, 7]] = -l3d / 2
    verts[:, 0, [1, 2, 5, 6]] = l3d / 2
    verts[:, 1, [0, 1, 4, 5]] = -h3d / 2
    verts[:, 1, [2, 3, 6, 7]] = h3d / 2
    verts[:, 2, [0, 1, 2, 3]] = -w3d / 2
    verts[:, 2, [4, 5, 6, 7]] = w3d / 2
    verts = R @ verts
    verts[:, 0, :] += x3d
    verts[:, 1, :] += y3d
    verts[:, 2, :] += z3d
    return verts.transpose(1, 2)


# ----------------------------------------------------------------------------- 2D box branch
def fc_head(x: torch.Tensor, sd, prefix: str) -> torch.Tensor:
    x = F.relu(F.linear(x, sd[prefix + "fc1.weight"], sd[prefix + "fc1.bias"]))
    x = F.relu(F.linear(x, sd[prefix + "fc2.weight"], sd[prefix + "fc2.bias"]))
    return x


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk):
    """fast_rcnn.py:57-116. boxes [R, C*4], scores [R, C+1]."""
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if not valid.all():
        boxes, scores = boxes[valid], scores[valid]
    scores = scores[:, :-1]
    nreg = boxes.shape[1] // 4
    boxes = clip_boxes(boxes.reshape(-1, 4), image_shape).view(-1, nreg, 4)
    mask = scores > score_thresh
    inds = mask.nonzero()
    boxes = boxes[inds[:, 0], 0] if nreg == 1 else boxes[mask]
    scores_full = scores[inds[:, 0]]
    scores = scores[mask]
    keep = batched_nms(boxes, scores, inds[:, 1], nms_thresh)
    if topk >= 0:
        keep = keep[:topk]
    boxes, scores, inds, scores_full = boxes[keep], scores[keep], inds[keep], scores_full[keep]
    return dict(pred_boxes=boxes, scores=scores, scores_full=scores_full, pred_classes=inds[:, 1]), inds[:, 0]


def forward_box(sd, feats: List[torch.Tensor], proposals: List[torch.Tensor], image_sizes, scales,
                score_thresh=0.01, nms_thresh=0.5, topk=100, bbox_weights=(10.0, 10.0, 5.0, 5.0),
                min_level=2, max_level=4, pooler_res=7) -> List[Dict[str, torch.Tensor]]:
    """roi_heads.py:252-296 inference branch. proposals: per-image [R_i,4]."""
    x = roi_pooler(feats, proposals, scales, pooler_res, min_level, max_level).flatten(1)
    x = fc_head(x, sd, "roi_heads.box_head.")
    cls = F.linear(x, sd["roi_heads.box_predictor.cls_score.weight"], sd["roi_heads.box_predictor.cls_score.bias"])
    dlt = F.linear(x, sd["roi_heads.box_predictor.bbox_pred.weight"], sd["roi_heads.box_predictor.bbox_pred.bias"])
    nums = [len(p) for p in proposals]
    allp = torch.cat(proposals, dim=0)
    pred = apply_deltas(dlt, allp, bbox_weights)
    probs = F.softmax(cls, dim=-1)
    out = []
    for pb, sc, shp in zip(pred.split(nums), probs.split(nums), image_sizes):
        r, _ = fast_rcnn_inference_single_image(pb, sc, shp, score_thresh, nms_thresh, topk)
        out.append(r)
    return out


# ----------------------------------------------------------------------------- cube branch
def cube_head_forward(x: torch.Tensor, sd, prefix="roi_heads.cube_head."):
    """cube_head.py:148-204, shared FC, class-agnostic (DIMS_PRIORS_ENABLED False), 6d pose, 1 bin."""
    f = fc_head(x, sd, prefix + "feature_generator.")
    lin = lambda n: F.linear(f, sd[prefix + n + ".weight"], sd[prefix + n + ".bias"])
    deltas = lin("bbox_3D_center_deltas")
    dims = lin("bbox_3D_dims")
    pose6 = lin("bbox_3D_pose").view(-1, 6)
    pose = rotation_6d_to_matrix(pose6)
    z = lin("bbox_3D_center_depth")
    uncert = lin("bbox_3D_uncertainty").clip(0.01)
    return deltas, z, dims, pose, uncert, f, pose6


def forward_cube(sd, feats: List[torch.Tensor], instances: List[Dict[str, torch.Tensor]], Ks: List[torch.Tensor],
                 im_dims: List[Tuple[int, int]], im_scales_ratio: List[float], scales: Sequence[float],
                 virtual_focal=512.0, min_level=2, max_level=4, pooler_res=7):
    """roi_heads.py:329-549 + :798-848, eval branch with Base.yaml:71-86 settings.
    instances: per-image dict with pred_boxes [n,4] (network res), scores [n], pred_classes [n].
    Adds pred_bbox3D, pred_center_cam, pred_center_2D, pred_dimensions, pred_pose; fuses scores."""
    boxes = [i["pred_boxes"] for i in instances]
    cube_features = roi_pooler(feats, boxes, scales, pooler_res, min_level, max_level).flatten(1)
    n = cube_features.shape[0]
    if n == 0:
        return instances, None                                           # roi_heads.py:371-372
    nums = [len(b) for b in boxes]
    Ks_box = torch.cat([(Ks[i] / im_scales_ratio[i]).unsqueeze(0).repeat([num, 1, 1]) for i, num in enumerate(nums)])
    Ks_box[:, -1, -1] = 1
    focal = torch.cat([Ks[i][1, 1].unsqueeze(0).repeat([num]) for i, num in enumerate(nums)])
    ratios = torch.cat([torch.FloatTensor([im_scales_ratio[i]]).repeat(num) for i, num in enumerate(nums)])
    im_scales = torch.cat([torch.FloatTensor([im_dims[i][0]]).repeat(num) for i, num in enumerate(nums)])
    im_scales_orig = im_scales * ratios
    virtual_to_real = compute_virtual_scale_from_focal_spaces(focal, im_scales_orig, virtual_focal, im_scales)

    src = torch.cat(boxes, dim=0)
    src_w = src[:, 2] - src[:, 0]
    src_h = src[:, 3] - src[:, 1]
    ctr_x = src[:, 0] + 0.5 * src_w
    ctr_y = src[:, 1] + 0.5 * src_h

    deltas, z, dims, pose, uncert, fc_feat, pose6 = cube_head_forward(cube_features, sd)
    uncert = uncert[:, 0]
    cube_x = ctr_x + src_w * deltas[:, 0]                                # roi_heads.py:480-481
    cube_y = ctr_y + src_h * deltas[:, 1]
    cube_xy = torch.stack((cube_x, cube_y), dim=1)
    dims = torch.exp(dims.clip(max=5))                                    # :507
    pose = R_from_allocentric(Ks_box, pose, u=cube_x, v=cube_y)          # :513
    z = z.squeeze()                                                       # :515
    z = z * virtual_to_real                                               # :548-549
    if z.dim() == 0:
        z = z.unsqueeze(0)                                                # :798-799
    x3d = z * (cube_x - Ks_box[:, 0, 2]) / Ks_box[:, 0, 0]                # :802-803
    y3d = z * (cube_y - Ks_box[:, 1, 2]) / Ks_box[:, 1, 1]
    cube_3D = torch.cat((torch.stack((x3d, y3d, z)).T, dims, cube_xy * ratios.unsqueeze(1)), dim=1)
    conf = torch.exp(-uncert)                                             # :807
    cube_3D = torch.cat((cube_3D, conf.unsqueeze(1)), dim=1)
    out = []
    for c3, ps, p6, inst in zip(cube_3D.split(nums), pose.split(nums), pose6.split(nums), instances):
        o = dict(inst)
        o["_pose6d"] = p6                                                 # diagnostic only (tests/parity.py: conditioning of the 6-D -> R map)
        if "scores" in o:
            o["scores"] = (o["scores"] * c3[:, -1]) ** (1 / 2)            # :825
        else:
            o["scores"] = c3[:, -1]
        o["pred_bbox3D"] = get_cuboid_verts(c3[:, :6], ps)                # :839
        o["pred_center_cam"] = c3[:, :3]
        o["pred_center_2D"] = c3[:, 6:8]
        o["pred_dimensions"] = c3[:, 3:6]
        o["pred_pose"] = ps
        out.append(o)
    aux = dict(cube_features=cube_features, fc_feat=fc_feat, deltas=deltas, z=z, dims=dims, uncert=uncert,
               virtual_to_real=virtual_to_real)
    return out, aux
