"""MAE ViT encoder + Simple Feature Pyramid backbone plugin (native).

Mirrors the reference plugin surface cubercnn/modeling/backbone/mae.py: ``build_mae_backbone(cfg, input_shape, priors=None)`` (:120-150)
= ``MAEBackbone`` (:20-118: Hugging Face ``ViTMAEForPreTraining(...).vit`` without masking, position table rebuilt as a 2-D sin-cos
table for the input grid, dense output of ``hidden_states[num_layers - 1]``) inside detectron2's ``SimpleFeaturePyramid`` with scale
factors (4, 2, 1, 0.5) -> ``{"p2","p3","p4","p5"}``. All arithmetic runs in libovm3d (``OvmConfig.tower = OVM_TOWER_MAE``). Like every
non-DINO backbone it takes no ``prompt_depth`` (see backbone/clip.py).
"""
from __future__ import annotations

from typing import Optional

from ...native import Engine
from ...registry import BACKBONE_REGISTRY
from ...util.synth_weights import MAE_ARCH
from .clip import SimpleFeaturePyramid
from .dino import ShapeSpec


class MAEBackbone:
    """Configuration holder for the tower (reference MAEBackbone.__init__, mae.py:20-60)."""

    def __init__(self, cfg, input_shape=None, checkpoint="facebook/vit-mae-base", output="dense", layer=-1, return_multilayer=False,
                 out_feature="last_feat"):
        assert output in ["cls", "gap", "dense", "dense-cls"]
        if checkpoint not in MAE_ARCH:
            raise ValueError(f"unknown MAE checkpoint {checkpoint}")
        if output != "dense" or return_multilayer:
            raise NotImplementedError("native path: MODEL.MAE.OUTPUT 'dense', single layer only")
        self.checkpoint_name = checkpoint.split("/")[1]
        self.output = output
        self.feat_dim, num_layers, _, self.patch_size = MAE_ARCH[checkpoint]
        self.multilayers = [num_layers - 1 if layer == -1 else layer]        # an index into hidden_states (0 = embeddings), :43-55
        if self.multilayers != [num_layers - 1]:
            raise NotImplementedError("native path: MODEL.MAE.LAYER -1")
        self.layer = "-".join(str(x) for x in self.multilayers)
        self.use_depth_fusion = False
        self._out_feature_channels = {out_feature: self.feat_dim}
        self._out_feature_strides = {out_feature: self.patch_size}
        self._out_features = [out_feature]

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}


@BACKBONE_REGISTRY.register()
def build_mae_backbone(cfg, input_shape=None, priors=None, engine: Optional[Engine] = None):
    bottom_up = MAEBackbone(cfg, input_shape, checkpoint=cfg.MODEL.MAE.CHECKPOINT, output=cfg.MODEL.MAE.OUTPUT, layer=cfg.MODEL.MAE.LAYER,
                            return_multilayer=cfg.MODEL.MAE.RETURN_MULTILAYER)
    return SimpleFeaturePyramid(net=bottom_up, in_feature=cfg.MODEL.FPN.IN_FEATURE, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
                                scale_factors=(4.0, 2.0, 1.0, 0.5), norm=cfg.MODEL.FPN.NORM, top_block=None,
                                square_pad=cfg.MODEL.FPN.SQUARE_PAD, engine=engine, cfg=cfg)
