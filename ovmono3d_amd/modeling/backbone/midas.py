"""MiDaS DPT_Large ViT-L/16 + Simple Feature Pyramid backbone plugin (native).

Mirrors the reference plugin surface cubercnn/modeling/backbone/midas_final.py: ``build_midas_backbone(cfg, input_shape, priors=None)``
(:98-127) = ``MIDASBackbone`` (:19-95: ``torch.hub`` MiDaS ``DPT_Large``'s ``pretrained.model`` - timm's vit_large_patch16_384 -, position
table resized with the CLIP backbone's antialiased bicubic, class token, ``norm_pre`` (identity), all 24 blocks, dense output of the last
one) inside detectron2's ``SimpleFeaturePyramid`` with scale factors (4, 2, 1, 0.5). All arithmetic runs in libovm3d
(``OvmConfig.tower = OVM_TOWER_MIDAS``). No ``prompt_depth`` (see backbone/clip.py).
"""
from __future__ import annotations

from typing import Optional

from ...native import Engine
from ...registry import BACKBONE_REGISTRY
from ...util.synth_weights import MIDAS_ARCH
from .clip import SimpleFeaturePyramid
from .dino import ShapeSpec


class MIDASBackbone:
    """Configuration holder for the tower (reference MIDASBackbone.__init__, midas_final.py:19-55)."""

    def __init__(self, cfg, input_shape=None, output="dense", layer=-1, return_multilayer=False, out_feature="last_feat", arch="DPT_Large"):
        if arch not in MIDAS_ARCH:
            raise ValueError(f"unknown MiDaS arch {arch}")
        if output != "dense" or return_multilayer:
            raise NotImplementedError("native path: MODEL.MIDAS.OUTPUT 'dense', single layer only")
        self.output = output
        self.feat_dim, num_layers, _, self.patch_size, grid = MIDAS_ARCH[arch]
        self.image_size = (grid * self.patch_size, grid * self.patch_size)
        self.multilayers = [num_layers - 1 if layer == -1 else layer]
        if self.multilayers != [num_layers - 1]:
            raise NotImplementedError("native path: MODEL.MIDAS.LAYER -1 (last block)")
        self.layer = "-".join(str(x) for x in self.multilayers)
        self.use_depth_fusion = False
        self._out_feature_channels = {out_feature: self.feat_dim}
        self._out_feature_strides = {out_feature: self.patch_size}
        self._out_features = [out_feature]

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}


@BACKBONE_REGISTRY.register()
def build_midas_backbone(cfg, input_shape=None, priors=None, engine: Optional[Engine] = None):
    bottom_up = MIDASBackbone(cfg, input_shape, output=cfg.MODEL.MIDAS.OUTPUT, layer=cfg.MODEL.MIDAS.LAYER,
                              return_multilayer=cfg.MODEL.MIDAS.RETURN_MULTILAYER, arch=cfg.MODEL.MIDAS.ARCH)
    return SimpleFeaturePyramid(net=bottom_up, in_feature=cfg.MODEL.FPN.IN_FEATURE, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
                                scale_factors=(4.0, 2.0, 1.0, 0.5), norm=cfg.MODEL.FPN.NORM, top_block=None,
                                square_pad=cfg.MODEL.FPN.SQUARE_PAD, engine=engine, cfg=cfg)
