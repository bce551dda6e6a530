"""SAM ViT-B image encoder + Simple Feature Pyramid backbone plugin (native).

Mirrors the reference plugin surface cubercnn/modeling/backbone/sam.py: ``build_sam_backbone(cfg, input_shape, priors=None)`` (:115-140) =
``SAMBackbone`` (:19-112: segment_anything ``sam_model_registry['vit_b']``'s ``image_encoder`` - patch embedding, absolute position table
(bicubic-resized for another input size), the 12 blocks with windowed / global attention and decomposed relative positions, dense output of
the last block, neck unused) inside detectron2's ``SimpleFeaturePyramid`` with scale factors (4, 2, 1, 0.5). All arithmetic runs in libovm3d
(``OvmConfig.tower = OVM_TOWER_SAM``). No ``prompt_depth`` (see backbone/clip.py).
"""
from __future__ import annotations

from typing import Optional

from ...native import Engine
from ...registry import BACKBONE_REGISTRY
from ...util.synth_weights import SAM_ARCH
from .clip import SimpleFeaturePyramid
from .dino import ShapeSpec


class SAMBackbone:
    """Configuration holder for the tower (reference SAMBackbone.__init__, sam.py:19-71)."""

    def __init__(self, cfg, input_shape=None, checkpoint="facebook/vit-mae-base", output="dense", layer=-1, return_multilayer=False,
                 out_feature="last_feat", arch="vit_b"):
        assert output in ["cls", "gap", "dense", "dense-cls"]
        if arch not in SAM_ARCH:
            raise ValueError(f"unknown SAM arch {arch}")
        if output != "dense" or return_multilayer:
            raise NotImplementedError("native path: MODEL.SAM.OUTPUT 'dense', single layer only")
        self.output = output
        self.feat_dim, num_layers, _, self.patch_size, grid, _, _ = SAM_ARCH[arch]
        assert self.patch_size == 16
        self.image_size = (grid * self.patch_size, grid * self.patch_size)
        self.multilayers = [num_layers - 1 if layer == -1 else layer]
        if self.multilayers != [num_layers - 1]:
            raise NotImplementedError("native path: MODEL.SAM.LAYER -1 (last block)")
        self.layer = "-".join(str(x) for x in self.multilayers)
        self.use_depth_fusion = False
        self._out_feature_channels = {out_feature: self.feat_dim}
        self._out_feature_strides = {out_feature: self.patch_size}
        self._out_features = [out_feature]

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}


@BACKBONE_REGISTRY.register()
def build_sam_backbone(cfg, input_shape=None, priors=None, engine: Optional[Engine] = None):
    bottom_up = SAMBackbone(cfg, input_shape, output=cfg.MODEL.SAM.OUTPUT, layer=cfg.MODEL.SAM.LAYER,
                            return_multilayer=cfg.MODEL.SAM.RETURN_MULTILAYER, arch=cfg.MODEL.SAM.ARCH)
    return SimpleFeaturePyramid(net=bottom_up, in_feature=cfg.MODEL.FPN.IN_FEATURE, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
                                scale_factors=(4.0, 2.0, 1.0, 0.5), norm=cfg.MODEL.FPN.NORM, top_block=None,
                                square_pad=cfg.MODEL.FPN.SQUARE_PAD, engine=engine, cfg=cfg)
