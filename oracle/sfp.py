"""Simple Feature Pyramid neck (fp32, CPU).

Follows reference cubercnn/modeling/backbone/dino.py:124-153 (scale_factors (2,1,0.5), norm 'LN')
and :208-224 (SimpleFeaturePyramidWithDepth.forward); the stages are detectron2's
``SimpleFeaturePyramid`` (yaojin17 fork, not in the container) - restated from the published
ViTDet definition and the module printout at reference nohup.out:565-596.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def channel_layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """detectron2.layers.LayerNorm on NCHW: normalise over channels at each pixel (biased variance)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def _conv_ln(x, sd, prefix, pad):
    x = F.conv2d(x, sd[prefix + ".weight"], None, padding=pad)     # bias=False with norm (nohup.out:567-574)
    return channel_layer_norm(x, sd[prefix + ".norm.weight"], sd[prefix + ".norm.bias"])


def sfp_forward(sd: Dict[str, torch.Tensor], feat: torch.Tensor, prefix: str = "backbone.") -> Dict[str, torch.Tensor]:
    """feat [B,D,G,G] -> {'p2' stride 7, 'p3' stride 14, 'p4' stride 28}, each [B,256,*,*]."""
    p2 = F.conv_transpose2d(feat, sd[prefix + "simfp_2.0.weight"], sd[prefix + "simfp_2.0.bias"], stride=2)
    p2 = _conv_ln(p2, sd, prefix + "simfp_2.1", 0)
    p2 = _conv_ln(p2, sd, prefix + "simfp_2.2", 1)
    p3 = _conv_ln(feat, sd, prefix + "simfp_3.0", 0)
    p3 = _conv_ln(p3, sd, prefix + "simfp_3.1", 1)
    p4 = F.max_pool2d(feat, kernel_size=2, stride=2)
    p4 = _conv_ln(p4, sd, prefix + "simfp_4.1", 0)
    p4 = _conv_ln(p4, sd, prefix + "simfp_4.2", 1)
    return {"p2": p2, "p3": p3, "p4": p4}


def sfp4_forward(sd: Dict[str, torch.Tensor], feat: torch.Tensor, prefix: str = "backbone.") -> Dict[str, torch.Tensor]:
    """detectron2 SimpleFeaturePyramid with scale_factors (4, 2, 1, 0.5) as the reference's CLIP / MAE / SAM builders use it
    (cubercnn/modeling/backbone/clip.py:155-166; published ViTDet definition): feat [B,D,G,G] at stride P ->
    {'p2' P/4, 'p3' P/2, 'p4' P, 'p5' 2P}. Scale 4 is ConvT . LN . GELU . ConvT before the two convs."""
    p2 = F.conv_transpose2d(feat, sd[prefix + "simfp_2.0.weight"], sd[prefix + "simfp_2.0.bias"], stride=2)
    p2 = F.gelu(channel_layer_norm(p2, sd[prefix + "simfp_2.1.weight"], sd[prefix + "simfp_2.1.bias"]))
    p2 = F.conv_transpose2d(p2, sd[prefix + "simfp_2.3.weight"], sd[prefix + "simfp_2.3.bias"], stride=2)
    p2 = _conv_ln(p2, sd, prefix + "simfp_2.4", 0)
    p2 = _conv_ln(p2, sd, prefix + "simfp_2.5", 1)
    p3 = F.conv_transpose2d(feat, sd[prefix + "simfp_3.0.weight"], sd[prefix + "simfp_3.0.bias"], stride=2)
    p3 = _conv_ln(p3, sd, prefix + "simfp_3.1", 0)
    p3 = _conv_ln(p3, sd, prefix + "simfp_3.2", 1)
    p4 = _conv_ln(feat, sd, prefix + "simfp_4.0", 0)
    p4 = _conv_ln(p4, sd, prefix + "simfp_4.1", 1)
    p5 = F.max_pool2d(feat, kernel_size=2, stride=2)
    p5 = _conv_ln(p5, sd, prefix + "simfp_5.1", 0)
    p5 = _conv_ln(p5, sd, prefix + "simfp_5.2", 1)
    return {"p2": p2, "p3": p3, "p4": p4, "p5": p5}
