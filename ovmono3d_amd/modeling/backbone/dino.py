"""DINOv2 ViT + Simple Feature Pyramid backbone plugin (native).

Mirrors the reference plugin surface cubercnn/modeling/backbone/dino.py:
``build_dino_backbone(cfg, input_shape, priors=None)`` (:123-153) returning a backbone whose
``forward(x, prompt_depth=None)`` yields ``{"p2","p3","p4"}`` (:208-224) and which exposes
``output_shape()``, ``size_divisibility`` and ``padding_constraints`` to the meta-architecture.
All arithmetic (normalise+pad, patch embed, ViT blocks, SFP) runs in libovm3d (ovm_backbone_forward).
"""
from __future__ import annotations

from collections import namedtuple
from typing import Dict, Optional

import torch

from ...native import Engine
from ...registry import BACKBONE_REGISTRY
from ...structures import ImageList
from ...util.synth_weights import VIT_ARCH

ShapeSpec = namedtuple("ShapeSpec", ["channels", "height", "width", "stride"], defaults=(None, None, None, None))


class FeatureRef:
    """Handle to a pyramid level kept inside the native workspace (avoids a D2D copy per call).
    ``.tensor()`` materialises it as a logical-NCHW torch tensor (NHWC storage)."""

    def __init__(self, engine: Engine, name: str, B: int, hw: int, channels: int):
        self.engine, self.name, self.B, self.hw, self.channels = engine, name, B, hw, channels

    @property
    def shape(self):
        return (self.B, self.channels, self.hw, self.hw)

    @property
    def device(self):
        return self.engine.device

    def tensor(self) -> torch.Tensor:
        n = self.B * self.hw * self.hw * self.channels
        t = self.engine.debug_tensor(self.name, n)
        return t.view(self.B, self.hw, self.hw, self.channels).permute(0, 3, 1, 2)


class DINOBackbone:
    """Configuration holder for the ViT tower (reference DINOBackbone.__init__, dino.py:15-68)."""

    def __init__(self, cfg, input_shape=None, dino_name="dinov2", model_name="vitb14", output="dense", layer=-1,
                 return_multilayer=False, out_feature="last_feat"):
        if model_name not in VIT_ARCH:
            raise ValueError(f"unknown DINOv2 arch {model_name}")
        assert output in ["cls", "gap", "dense", "dense-cls"]
        if output != "dense" or return_multilayer:
            raise NotImplementedError("native path: MODEL.DINO.OUTPUT 'dense', single layer only")
        self.model_name = dino_name
        self.checkpoint_name = f"{dino_name}_{model_name}"
        self.has_registers = "_reg" in model_name
        self.use_depth_fusion = cfg.MODEL.DINO.USE_DEPTH_FUSION
        self.output = output
        self.patch_size = 14
        self.feat_dim = VIT_ARCH[model_name][0]
        num_layers = VIT_ARCH[model_name][1]
        self.multilayers = [num_layers - 1 if layer == -1 else layer]
        self.layer = "-".join(str(x) for x in self.multilayers)
        self._out_feature_channels = {out_feature: self.feat_dim}
        self._out_feature_strides = {out_feature: self.patch_size}
        self._out_features = [out_feature]

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}


class SimpleFeaturePyramidWithDepth:
    """reference dino.py:208-224 on top of detectron2's SimpleFeaturePyramid (scale factors (2,1,.5))."""

    def __init__(self, net: DINOBackbone, in_feature, out_channels, scale_factors, norm="LN", top_block=None,
                 square_pad=0, engine: Optional[Engine] = None, cfg=None):
        if tuple(scale_factors) not in ((2.0, 1.0, 0.5), (4.0, 2.0, 1.0, 0.5)) or norm != "LN" or top_block is not None:
            raise NotImplementedError("native SFP: scale_factors (2, 1, 0.5) or (4, 2, 1, 0.5), norm 'LN', no top block")
        self.net = net
        self.in_feature = in_feature
        self.scale_factors = scale_factors
        stride = net.patch_size
        strides = [int(stride / s) for s in scale_factors]
        self._out_feature_strides = {"p{}".format(int(__import__("math").log2(s))): s for s in strides}
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = strides[-1]
        self._square_pad = square_pad
        self.engine = engine if engine is not None else Engine(cfg)
        self.export_features = False
        self.training = False

    @property
    def size_divisibility(self):
        return self._size_divisibility

    @property
    def padding_constraints(self):
        return {"size_divisiblity": self._size_divisibility, "square_size": self._square_pad}

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}

    def forward(self, x, prompt_depth=None) -> Dict[str, object]:
        """x: ``ImageList`` produced by ``RCNN3D.preprocess_image`` (uint8 images + sizes; normalisation and
        the zero pad to the square canvas happen inside the patch-gather kernel) or a list of per-image
        dicts. Returns {"p2","p3","p4"} (torch NCHW views when ``export_features`` else ``FeatureRef``)."""
        if isinstance(x, torch.Tensor):
            raise TypeError("native backbone consumes the uint8 image batch (ImageList from preprocess_image), "
                            "not a pre-normalised float tensor: preprocessing is fused into the patch-embed load")
        if isinstance(x, ImageList):
            native, B = x.native, len(x)
        else:
            native, keep = self.engine.make_images(x)
            B = len(x)
        feats = self.engine.backbone_forward(native, B, prompt_depth, export=self.export_features)
        if feats is not None:
            return feats
        return {k: FeatureRef(self.engine, k, B, g, self.engine.C) for k, g, _ in self.engine.levels}

    __call__ = forward


@BACKBONE_REGISTRY.register()
def build_dino_backbone(cfg, input_shape=None, priors=None, engine: Optional[Engine] = None):
    bottom_up = DINOBackbone(cfg, input_shape, dino_name=cfg.MODEL.DINO.NAME, model_name=cfg.MODEL.DINO.MODEL_NAME,
                             output=cfg.MODEL.DINO.OUTPUT, layer=cfg.MODEL.DINO.LAYER,
                             return_multilayer=cfg.MODEL.DINO.RETURN_MULTILAYER)
    return SimpleFeaturePyramidWithDepth(net=bottom_up, in_feature=cfg.MODEL.FPN.IN_FEATURE,
                                         out_channels=cfg.MODEL.FPN.OUT_CHANNELS, scale_factors=(2.0, 1.0, 0.5),
                                         norm=cfg.MODEL.FPN.NORM, top_block=None, square_pad=cfg.MODEL.FPN.SQUARE_PAD,
                                         engine=engine, cfg=cfg)
