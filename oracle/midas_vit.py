"""MiDaS DPT_Large's ViT-L/16 forward as the reference's MIDASBackbone runs it (fp32, CPU). TEST INFRASTRUCTURE: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg only.

Follows reference cubercnn/modeling/backbone/midas_final.py:57-95: patch embedding, class token, position table resized with the CLIP
backbone's ``resize_pos_embed`` (antialiased bicubic, :64-66), ``norm_pre`` (identity for this model), every block, dense output of the
last one without the final norm. The tower is timm's ``vit_large_patch16_384`` as MiDaS (torch.hub intel-isl/MiDaS, ``DPT_Large``) holds
it - source absent from the container, restated from the published timm ``VisionTransformer`` / ``Block`` definition: x + attn(norm1 x),
x + mlp(norm2 x), fused qkv linear, no LayerScale, erf-GELU, LayerNorm eps 1e-6. Cross-checked against Hugging Face ``ViTModel`` layers
(same architecture) in tests/test_oracle_crosscheck.py. Parity unpinned vs the reference itself.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .clip_vit import resize_pos_embed


def timm_block(x: torch.Tensor, sd, p: str, heads: int, eps: float = 1e-6) -> torch.Tensor:
    B, T, D = x.shape
    dh = D // heads
    h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, T, 3, heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (dh ** -0.5), qkv[1], qkv[2]
    a = ((q @ k.transpose(-2, -1)).softmax(dim=-1) @ v).transpose(1, 2).reshape(B, T, D)
    x = x + F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def midas_backbone_forward(sd: Dict[str, torch.Tensor], images: torch.Tensor, heads: int, depth: int,
                           prefix: str = "backbone.net.vit.") -> torch.Tensor:
    w = sd[prefix + "patch_embed.proj.weight"]
    P = w.shape[-1]
    x = F.conv2d(images, w, sd[prefix + "patch_embed.proj.bias"], stride=P)
    gh, gw = x.shape[-2:]
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd[prefix + "cls_token"].expand(x.shape[0], -1, -1), x], dim=1)       # :70
    x = x + resize_pos_embed(sd[prefix + "pos_embed"][0], (gh, gw))[None]                # :64-66,71
    for i in range(depth):                                                                # norm_pre is Identity (:73)
        x = timm_block(x, sd, prefix + f"blocks.{i}.", heads)
    return x[:, 1:].reshape(x.shape[0], gh, gw, -1).permute(0, 3, 1, 2).contiguous()
