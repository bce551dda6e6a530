"""RPN plugin. The reference's ``RPNWithIgnore`` (cubercnn/modeling/proposal_generator/rpn.py:19-39) changes
only training; inference is detectron2's RPN. In the native path the RPN head, anchor decode, top-k and
NMS run fused with the 2D box head inside ``ovm_rpn_box_forward``; ``forward`` therefore returns a
deferred ``NativeProposals`` token that ``ROIHeads3D._forward_box`` resolves."""
from __future__ import annotations

from ...registry import PROPOSAL_GENERATOR_REGISTRY


class NativeProposals(list):
    """Per-image placeholders for proposals that stay on the device inside the native workspace."""

    def __init__(self, images):
        super().__init__([None] * len(images))
        self.images = images


@PROPOSAL_GENERATOR_REGISTRY.register()
class RPNWithIgnore:
    def __init__(self, cfg, input_shape=None, engine=None):
        self.in_features = cfg.MODEL.RPN.IN_FEATURES
        self.pre_nms_topk = cfg.MODEL.RPN.PRE_NMS_TOPK_TEST
        self.post_nms_topk = cfg.MODEL.RPN.POST_NMS_TOPK_TEST
        self.nms_thresh = cfg.MODEL.RPN.NMS_THRESH
        self.engine = engine
        self.training = False

    def forward(self, images, features, gt_instances=None):
        assert not self.training, "training is out of scope of the native inference path"
        return NativeProposals(images), {}

    __call__ = forward
