"""CLIP ViT image tower + Simple Feature Pyramid backbone plugin (native).

Mirrors the reference plugin surface cubercnn/modeling/backbone/clip.py: ``build_clip_backbone(cfg, input_shape, priors=None)``
(:135-166) = ``CLIPBackbone`` (:17-96: open_clip ``ViT-B-16`` visual tower, dense output of the last block, no ln_post) inside
detectron2's ``SimpleFeaturePyramid`` with scale factors (4, 2, 1, 0.5) -> ``{"p2","p3","p4","p5"}`` at strides 4 / 8 / 16 / 32.
All arithmetic runs in libovm3d (``ovm_backbone_forward`` with ``OvmConfig.tower = OVM_TOWER_CLIP``).

The fork's ``RCNN3D`` passes ``prompt_depth`` to every backbone (rcnn3d.py:97) and detectron2's ``SimpleFeaturePyramid.forward``
does not take one, so the reference's CLIP config cannot run with a depth prompt; here the argument is accepted
and must be ``None``.
"""
from __future__ import annotations

from typing import Optional

from ...native import Engine
from ...registry import BACKBONE_REGISTRY
from ...util.synth_weights import CLIP_ARCH
from .dino import ShapeSpec, SimpleFeaturePyramidWithDepth


class CLIPBackbone:
    """Configuration holder for the tower (reference CLIPBackbone.__init__, clip.py:17-60)."""

    def __init__(self, cfg, input_shape=None, arch="ViT-B-16", checkpoint="openai", output="dense", layer=-1,
                 return_multilayer=False, out_feature="last_feat"):
        assert output in ["dense-cls", "cls", "gap", "dense"]
        if arch not in CLIP_ARCH:
            raise ValueError(f"unknown CLIP arch {arch}")
        if output != "dense" or return_multilayer:
            raise NotImplementedError("native path: MODEL.CLIP.OUTPUT 'dense', single layer only")
        self.output = output
        self.checkpoint_name = f"clip_{arch}_{checkpoint}"
        self.feat_dim, n_layers, _, self.patch_size, _ = CLIP_ARCH[arch]
        self.multilayers = [n_layers - 1 if layer == -1 else layer]
        if self.multilayers != [n_layers - 1]:
            raise NotImplementedError("native path: MODEL.CLIP.LAYER -1 (last block)")
        self.layer = "-".join(str(x) for x in self.multilayers)
        self.use_depth_fusion = False
        self._out_feature_channels = {out_feature: self.feat_dim}
        self._out_feature_strides = {out_feature: self.patch_size}
        self._out_features = [out_feature]

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}


class SimpleFeaturePyramid(SimpleFeaturePyramidWithDepth):
    """detectron2's SimpleFeaturePyramid as the reference's CLIP / MAE / SAM configs build it: no depth input."""

    def forward(self, x, prompt_depth=None):
        if prompt_depth is not None:
            raise TypeError("this backbone takes no prompt_depth (only the DINOv2 tower has the depth-fusion conv)")
        return super().forward(x, None)

    __call__ = forward


@BACKBONE_REGISTRY.register()
def build_clip_backbone(cfg, input_shape=None, priors=None, engine: Optional[Engine] = None):
    bottom_up = CLIPBackbone(cfg, input_shape, arch=cfg.MODEL.CLIP.ARCH, checkpoint=cfg.MODEL.CLIP.CHECKPOINT,
                             output=cfg.MODEL.CLIP.OUTPUT, layer=cfg.MODEL.CLIP.LAYER,
                             return_multilayer=cfg.MODEL.CLIP.RETURN_MULTILAYER)
    return SimpleFeaturePyramid(net=bottom_up, in_feature=cfg.MODEL.FPN.IN_FEATURE, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
                                scale_factors=(4.0, 2.0, 1.0, 0.5), norm=cfg.MODEL.FPN.NORM, top_block=None,
                                square_pad=cfg.MODEL.FPN.SQUARE_PAD, engine=engine, cfg=cfg)
