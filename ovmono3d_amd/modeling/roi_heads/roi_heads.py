"""ROIHeads3D plugin (reference cubercnn/modeling/roi_heads/roi_heads.py:39-848, inference branches).

``forward(images, features, proposals, Ks, im_scales_ratio, targets=None) -> (List[Instances], {})``
(:207-249). The oracle-2D branch (:232-243) builds the 2D instances on the host (a division by the
image scale ratio), everything else - box pooler + box head + Fast R-CNN inference (:252-296,
fast_rcnn.py:57-143) and the cube branch (:329-549,:798-848) - runs in libovm3d.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from ...native import Engine, records_to_fields
from ...registry import ROI_HEADS_REGISTRY
from ...structures import Boxes, Instances
from ..proposal_generator.rpn import NativeProposals
from .cube_head import build_cube_head


def build_roi_heads(cfg, input_shape=None, priors=None, engine: Optional[Engine] = None):
    name = cfg.MODEL.ROI_HEADS.NAME
    return ROI_HEADS_REGISTRY.get(name)(cfg, input_shape, priors=priors, engine=engine)


@ROI_HEADS_REGISTRY.register()
class ROIHeads3D:
    def __init__(self, cfg, input_shape=None, priors=None, engine: Optional[Engine] = None):
        self.cfg = cfg
        self.in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.loss_w_3d = cfg.MODEL.ROI_CUBE_HEAD.LOSS_W_3D
        self.cube_head = build_cube_head(cfg, input_shape)
        self.engine = engine if engine is not None else Engine(cfg)
        self.training = False

    # -- helpers -----------------------------------------------------------------------------
    def _instances_from_records(self, rec, counts, image_sizes, extra_full=None) -> List[Instances]:
        out, ofs = [], 0
        fields = records_to_fields(rec)
        for i, c in enumerate(counts):
            inst = Instances(image_sizes[i])
            sl = slice(ofs, ofs + c)
            inst.pred_boxes = Boxes(fields["pred_boxes"][sl])
            inst.scores = fields["scores"][sl]
            inst.pred_classes = fields["pred_classes"][sl]
            for k in ("pred_bbox3D", "pred_center_cam", "pred_center_2D", "pred_dimensions", "pred_pose"):
                inst.set(k, fields[k][sl])
            out.append(inst)
            ofs += c
        return out

    def _forward_cube(self, features, instances: List[Instances], Ks, im_current_dims, im_scales_ratio,
                      images=None, postprocess=False):
        """reference roi_heads.py:329-848 (eval). ``instances`` carry pred_boxes (network res), scores,
        pred_classes. Returns new Instances with the 3D fields; images with no box keep their 2D instance
        untouched when the whole batch is empty (:371-372)."""
        B = len(instances)
        nums = [len(i) for i in instances]
        n = sum(nums)
        if n == 0:
            return instances
        dev = self.engine.device
        boxes = torch.cat([i.pred_boxes.tensor.to(dev) for i in instances])
        scores = torch.cat([i.scores.to(dev, torch.float32) for i in instances])
        classes = torch.cat([i.pred_classes.to(dev) for i in instances])
        idx = torch.cat([torch.full((k,), b, dtype=torch.int32) for b, k in enumerate(nums)]).to(dev)
        rec, counts = self.engine.cube_forward(images.native, B, boxes, scores, classes, idx, postprocess=postprocess)
        if postprocess:
            sizes = [(int(images.native[b].orig_height), int(images.native[b].orig_width)) for b in range(B)]
        else:
            sizes = [tuple(d) for d in im_current_dims]
        out = self._instances_from_records(rec, counts, sizes)
        for o in out:
            o._postprocessed = bool(postprocess)
        return out

    def _forward_box(self, features, proposals):
        """reference roi_heads.py:252-296 inference branch (+ the RPN that produced ``proposals``)."""
        if not isinstance(proposals, NativeProposals):
            raise NotImplementedError("native _forward_box consumes the deferred proposals of RPNWithIgnore.forward")
        images = proposals.images
        B = len(images)
        boxes, scores, classes, idx, full, counts = self.engine.rpn_box_forward(images.native, B)
        out, ofs = [], 0
        for b, c in enumerate(counts):
            inst = Instances(images.image_sizes[b])
            sl = slice(ofs, ofs + c)
            inst.pred_boxes = Boxes(boxes[sl])
            inst.scores = scores[sl]
            inst.scores_full = full[sl]
            inst.pred_classes = classes[sl].to(torch.int64)
            out.append(inst)
            ofs += c
        return out

    def forward(self, images, features, proposals, Ks, im_scales_ratio, targets=None):
        assert not self.training, "training is out of scope of the native inference path"
        im_dims = list(images.image_sizes)
        fuse = bool(getattr(images, "fuse_postprocess", False))
        if isinstance(proposals, list) and not isinstance(proposals, NativeProposals) and \
                not np.any([isinstance(p, Instances) for p in proposals]):
            pred_instances = []                                            # oracle branch, :232-243
            for proposal, im_dim, r in zip(proposals, im_dims, im_scales_ratio):
                inst = Instances(im_dim)
                inst.pred_boxes = Boxes(torch.as_tensor(proposal["gt_bbox2D"], dtype=torch.float32) / r)
                inst.pred_classes = torch.as_tensor(proposal["gt_classes"])
                if "gt_scores" in proposal.keys():
                    inst.scores = torch.as_tensor(proposal["gt_scores"], dtype=torch.float32)
                else:
                    inst.scores = torch.ones_like(inst.pred_classes).float()
                pred_instances.append(inst)
        else:
            pred_instances = self._forward_box(features, proposals)
        if self.loss_w_3d > 0:
            pred_instances = self._forward_cube(features, pred_instances, Ks, im_dims, im_scales_ratio,
                                                images=images, postprocess=fuse)
        return pred_instances, {}

    __call__ = forward
