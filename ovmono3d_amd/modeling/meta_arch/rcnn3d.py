"""RCNN3D meta-architecture plugin (reference cubercnn/modeling/meta_arch/rcnn3d.py:25-276).

``RCNN3D.forward(batched_inputs, prompt_depth=None)`` -> ``inference`` (:41,:79-117): preprocess,
backbone, (oracle-2D | RPN -> box head | category_list -> GDINO head), cube head, postprocess.
Host code only sequences calls into libovm3d; tensors stay in HBM.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from ...native import Engine
from ...registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY
from ...structures import ImageList, Instances
from ..roi_heads import build_roi_heads


def build_backbone(cfg, input_shape=None, priors=None, engine=None):
    """reference rcnn3d.py:264-276"""
    name = cfg.MODEL.BACKBONE.NAME
    return BACKBONE_REGISTRY.get(name)(cfg, input_shape, priors, engine=engine)


def build_proposal_generator(cfg, input_shape=None, engine=None):
    name = cfg.MODEL.PROPOSAL_GENERATOR.NAME
    if name == "PrecomputedProposals":
        return None
    return PROPOSAL_GENERATOR_REGISTRY.get(name)(cfg, input_shape, engine=engine)


@META_ARCH_REGISTRY.register()
class RCNN3D:
    def __init__(self, cfg, priors=None, device=None):
        self.cfg = cfg
        self.engine = Engine(cfg, device)
        self.backbone = build_backbone(cfg, priors=priors, engine=self.engine)
        self.proposal_generator = build_proposal_generator(cfg, self.backbone.output_shape(), engine=self.engine)
        self.roi_heads = build_roi_heads(cfg, self.backbone.output_shape(), priors=priors, engine=self.engine)
        self.input_format = cfg.INPUT.FORMAT
        self.pixel_mean = list(cfg.MODEL.PIXEL_MEAN)
        self.pixel_std = list(cfg.MODEL.PIXEL_STD)
        self.training = False

    # nn.Module-like surface used by the entry points
    @property
    def device(self):
        return self.engine.device

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("training is out of scope of the native inference path")
        return self

    def to(self, *a, **k):
        return self

    def load_state_dict(self, state_dict, strict=True):
        self.engine.load_state_dict(state_dict)

    def __call__(self, batched_inputs, prompt_depth=None):
        return self.forward(batched_inputs, prompt_depth=prompt_depth)

    def forward(self, batched_inputs: List[Dict], prompt_depth=None):
        assert not self.training
        return self.inference(batched_inputs, prompt_depth=prompt_depth)

    def preprocess_image(self, batched_inputs) -> ImageList:
        """detectron2 ``preprocess_image`` (called at rcnn3d.py:88). Normalisation and the zero pad to the
        SQUARE_PAD canvas are fused into the patch-gather kernel; this only describes the images."""
        native, keep = self.engine.make_images(batched_inputs)
        sizes = [(int(native[i].height), int(native[i].width)) for i in range(len(batched_inputs))]
        il = ImageList(None, sizes)
        il.native, il.raw = native, keep
        return il

    def inference(self, batched_inputs: List[Dict], detected_instances=None, do_postprocess: bool = True,
                  prompt_depth: Optional[torch.Tensor] = None):
        assert not self.training
        images = self.preprocess_image(batched_inputs)
        im_scales_ratio = [info["height"] / s[0] for info, s in zip(batched_inputs, images.image_sizes)]   # :92
        Ks = [torch.FloatTensor(np.asarray(info["K"], dtype=np.float32)) for info in batched_inputs]      # :95
        images.fuse_postprocess = bool(do_postprocess)
        fused = self._infer_fused(batched_inputs, images, prompt_depth, do_postprocess)
        if fused is not None:
            return fused
        if hasattr(self.roi_heads, "prefetch") and "oracle2D" not in batched_inputs[0] and "category_list" in batched_inputs[0]:
            self.roi_heads.prefetch(images, batched_inputs[0]["category_list"])       # side stream, overlaps the backbone
        features = self.backbone(images, prompt_depth=prompt_depth)                                        # :97
        if isinstance(batched_inputs, list) and np.any(["oracle2D" in b for b in batched_inputs]):         # :100-102
            oracles = [b["oracle2D"] for b in batched_inputs]
            results, _ = self.roi_heads(images, features, oracles, Ks, im_scales_ratio, None)
        else:                                                                                              # :105-111
            proposals, _ = self.proposal_generator(images, features, None)
            if np.any(["category_list" in b for b in batched_inputs]):
                results, _ = self.roi_heads(images, features, proposals, Ks, im_scales_ratio, None,
                                            category_list=batched_inputs[0]["category_list"])
            else:
                results, _ = self.roi_heads(images, features, proposals, Ks, im_scales_ratio, None)
        if do_postprocess:
            return self._postprocess(results, batched_inputs, images.image_sizes)
        return results

    def _infer_fused(self, batched_inputs, images, prompt_depth, do_postprocess):
        """The text-prompted path as ONE C call (``ovm_infer``) when nothing needs the stages separately: ROIHeads3DGDINO with
        the native engine as its detector, one image with a category_list, no depth prompt, post-processing on. Same kernels in
        the same order as the staged path below (the detector beside the backbone on a side stream); returns None otherwise."""
        rh = self.roi_heads
        if not (do_postprocess and prompt_depth is None and len(batched_inputs) == 1 and hasattr(rh, "load_detector")
                and bool(self.cfg.MODEL.AMD.get("FUSED_INFER", True)) and bool(self.cfg.MODEL.AMD.GDINO_OVERLAP)
                and "oracle2D" not in batched_inputs[0] and batched_inputs[0].get("category_list")):
            return None
        if rh.detector is None:
            rh.load_detector()
        eng = getattr(rh.detector, "engine", None)
        if eng is None or rh.loss_w_3d <= 0 or getattr(self.backbone, "export_features", False):
            return None
        from ..roi_heads.gdino_glue import build_caption, phrase_spans
        cats = list(batched_inputs[0]["category_list"])
        caption, cap_list = build_caption(cats)
        ids, phrase_ids, _ = rh.detector._tokens(caption)
        spans = phrase_spans(ids, phrase_ids)
        rec, n = self.engine.infer_gdino(images.native, eng._h, ids, spans, 0.001, 0.5, eng.cfg.num_queries)   # roi_heads_gdino.py:148,254
        # phrase k -> class index: filtered_texts.index([name]) of the reference (:162) = first occurrence of the caption's name
        remap = [cats.index(c) if c in cats else -1 for c in cap_list]
        if any(r < 0 for r in remap):
            raise ValueError("category names must be lower-case / stripped for ROIHeads3DGDINO (reference roi_heads_gdino.py:162: "
                             "filtered_texts.index([class_name]) raises for names that build_caption normalised)")
        if remap != list(range(len(cap_list))) and n > 0:
            cls = rec.view(torch.int32)[:, 5]                      # rec: row-contiguous block of records
            cls.copy_(torch.tensor(remap, dtype=torch.int32, device=rec.device)[cls.long()])
        size = (int(images.native[0].orig_height), int(images.native[0].orig_width))
        if n == 0:
            from ...structures import Boxes, Instances
            inst = Instances(images.image_sizes[0])             # roi_heads.py:371-372: no 3D fields; 2D instance untouched
            inst.pred_boxes = Boxes(torch.zeros((0, 4), dtype=torch.float32))
            inst.scores = torch.zeros(0)
            inst.pred_classes = torch.zeros(0, dtype=torch.int64)
            return [{"instances": inst}]
        out = rh._instances_from_records(rec, [n], [size])
        out[0]._postprocessed = True
        return self._postprocess(out, batched_inputs, images.image_sizes)

    @staticmethod
    def _postprocess(instances, batched_inputs, image_sizes):
        """detectron2 ``GeneralizedRCNN._postprocess``. The rescale/clip/non-empty filter already ran fused in
        the cube-decode kernel; instances that never reached it (empty batches, roi_heads.py:371-372) carry no
        boxes to rescale."""
        out = []
        for r, inp, size in zip(instances, batched_inputs, image_sizes):
            if not getattr(r, "_postprocessed", False) and len(r) > 0:
                raise RuntimeError("instances were not post-processed by the native path")
            out.append({"instances": r})
        return out


def build_model(cfg, priors=None, device=None):
    """reference rcnn3d.py:252-262"""
    meta_arch = cfg.MODEL.META_ARCHITECTURE
    model = META_ARCH_REGISTRY.get(meta_arch)(cfg, priors=priors, device=device)
    return model
