"""ROIHeads3DGDINO plugin (reference cubercnn/modeling/roi_heads/roi_heads_gdino.py:26-171).

``forward(..., category_list=None)``: a text-prompted 2D detector supplies the boxes that enter
``_forward_cube``; as in the reference the RPN/box-head result is not used for the boxes (:118-171)
and batch size is 1 (:130-134, rcnn3d.py:108-109).

The network is pluggable: ``self.detector(image_u8_chw, caption) -> dict`` returning EITHER the raw
GroundingDINO outputs ``{"pred_logits": [nq,256], "pred_boxes": [nq,4] cxcywh, "input_ids": caption token ids,
"phrase_ids": per-category token ids}`` - the reference-owned glue (caption building :176-181, phrase-logit
reduction :273-294, threshold :197, cxcywh->xyxy :266-270, NMS :254, class index :162) then runs natively
(``gdino_glue`` + ``ovm_gdino_postprocess``) - OR already post-processed ``{"bboxes", "scores", "labels"}``.
By default the detector is the native GroundingDINO network (``ovmono3d_amd.gdino``: Swin-B + BERT + deformable
transformer on libovm3d ops), built on first use from ``MODEL.AMD.GDINO_WEIGHTS``; if neither a checkpoint nor a detector
is available ``forward`` raises instead of silently falling back.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from ...registry import ROI_HEADS_REGISTRY
from ...structures import Boxes, Instances
from .roi_heads import ROIHeads3D


@ROI_HEADS_REGISTRY.register()
class ROIHeads3DGDINO(ROIHeads3D):
    def __init__(self, cfg, input_shape=None, priors=None, engine=None, detector: Optional[Callable] = None):
        super().__init__(cfg, input_shape, priors=priors, engine=engine)
        self.detector = detector
        self._gdino_cfg = cfg
        self._side = None            # HIP stream the detector runs on while the DINOv2 backbone runs on the main one
        self._pending = None

    def load_detector(self):
        """Builds the native GroundingDINO (reference: load_model(...) in __init__, roi_heads_gdino.py:87-91). Deferred to
        first use so that a model without the checkpoint can still serve the oracle-2D / RPN paths."""
        import os
        from ...checkpoint import load_state_dict_file
        from ...gdino.detector import HashTokenizer, NativeGroundingDino
        from .gdino_glue import WordPieceTokenizer
        cfg = self._gdino_cfg
        path = cfg.MODEL.AMD.GDINO_WEIGHTS
        if path.startswith("synthetic://"):
            from ...util.synth_gdino_weights import synth_gdino_state_dict
            seed = int(path.split("seed=")[1]) if "seed=" in path else 0
            sd, tok = synth_gdino_state_dict(seed), HashTokenizer()
        else:
            if not os.path.isfile(path):
                raise NotImplementedError(
                    f"GroundingDINO checkpoint {path!r} not found (MODEL.AMD.GDINO_WEIGHTS) and no detector attached: "
                    "ROIHeads3DGDINO cannot produce 2D boxes. There is no fallback.")
            if not cfg.MODEL.AMD.BERT_VOCAB:
                raise NotImplementedError("MODEL.AMD.BERT_VOCAB (bert-base-uncased vocab.txt) is required to tokenise the caption")
            sd, tok = load_state_dict_file(path), WordPieceTokenizer(cfg.MODEL.AMD.BERT_VOCAB)
        prec = 3 if cfg.MODEL.AMD.GEMM_PRECISION == "f16x3" else 1
        self.detector = NativeGroundingDino(self.engine.device, sd, tok, cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, precision=prec,
                                            use_graphs=bool(cfg.MODEL.AMD.GDINO_GRAPHS))

    def prefetch(self, images, category_list):
        """Called by the meta-architecture before the backbone: starts the text-prompted detector on a side stream. It
        reads only the input image, so it overlaps the DINOv2 ViT (independent work, under-filled chip: ~1,300 short
        kernels) instead of running after it as in the reference's serial order (rcnn3d.py:97-111)."""
        self._pending = None
        if not category_list or len(images.image_sizes) != 1 or not bool(self._gdino_cfg.MODEL.AMD.GDINO_OVERLAP):
            self.engine.set_corun(False)
            return
        self.engine.set_corun(bool(self._gdino_cfg.MODEL.AMD.get("GDINO_CORUN", False)))   # optional: attention leaves room for the detector
        if self.detector is None:
            self.load_detector()
        from .gdino_glue import build_caption
        dev = self.engine.device
        caption, cap_list = build_caption(list(category_list))
        if self._side is None:
            self._side = torch.cuda.Stream(dev, priority=-1)      # short kernels: let them jump the ViT's queue
        main = torch.cuda.current_stream(dev)
        self._side.wait_stream(main)                               # the uploaded image is ready
        with torch.cuda.stream(self._side):
            det = self.detector(images.raw[0], caption)
        done = torch.cuda.Event()
        done.record(self._side)
        for v in det.values():
            if isinstance(v, torch.Tensor) and v.is_cuda:
                v.record_stream(main)
        self._pending = (images, caption, cap_list, det, done)

    def forward(self, images, features, proposals, Ks, im_scales_ratio, targets=None, category_list=None):
        assert not self.training, "training is out of scope of the native inference path"
        im_dims = list(images.image_sizes)
        fuse = bool(getattr(images, "fuse_postprocess", False))
        if category_list:
            filtered_texts = [[cat] for cat in category_list]
        else:
            # the reference leaves ``filtered_texts`` unbound here (NameError, :130-134)
            raise NameError("ROIHeads3DGDINO.forward requires category_list (reference roi_heads_gdino.py:130-134)")
        if len(im_dims) != 1:
            raise ValueError("GroundingDINO inference supports one image per batch (reference rcnn3d.py:108)")
        if self.detector is None:
            self.load_detector()
        from .gdino_glue import build_caption, gdino_postprocess, phrase_spans
        caption, cap_list = build_caption([t[0] for t in filtered_texts])
        pend, self._pending = self._pending, None
        if pend is not None and pend[0] is images and pend[1] == caption:
            det = pend[3]
            torch.cuda.current_stream(self.engine.device).wait_event(pend[4])
        else:
            det = self.detector(images.raw[0], caption)
        target = Instances(im_dims[0])
        if "pred_logits" in det:
            spans = phrase_spans(det["input_ids"], det["phrase_ids"])
            dev = self.engine.device
            boxes, scores, cls = gdino_postprocess(det["pred_logits"].to(dev), det["pred_boxes"].to(dev), spans, im_dims[0],
                                                   box_threshold=0.001, nms_threshold=0.5)          # :148, :254
            class_names = [cap_list[int(i)] for i in cls.cpu()]
        else:
            boxes = torch.as_tensor(det["bboxes"], dtype=torch.float32).reshape(-1, 4)
            scores = torch.as_tensor(det["scores"], dtype=torch.float32)
            class_names = det["labels"]
        target.pred_classes = torch.tensor([filtered_texts.index([c]) for c in class_names], dtype=torch.int64)  # :162
        target.pred_boxes = Boxes(boxes)
        target.scores = scores
        pred = [target]
        if self.loss_w_3d > 0:
            pred = self._forward_cube(features, pred, Ks, im_dims, im_scales_ratio, images=images, postprocess=fuse)
        return pred, {}

    __call__ = forward
