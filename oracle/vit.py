"""DINOv2 ViT forward as the reference's DINOBackbone runs it (fp32, CPU).

Follows reference cubercnn/modeling/backbone/dino.py:70-120 (forward), :155-174 (tokens_to_output)
and the facebookresearch/dinov2 @ main hub model it loads at dino.py:29 (source not in the container;
restated from the published model definition: PatchEmbed conv 14x14/14, cls token, bicubic pos-embed
interpolation with interpolate_offset=0.1 / antialias=False, pre-norm blocks with LayerScale,
exact-erf GELU; module structure cross-checked against reference nohup.out:598-627).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

PATCH = 14


def interpolate_pos_encoding(pos_embed: torch.Tensor, gh: int, gw: int,
                             interpolate_offset: float = 0.1) -> torch.Tensor:
    """dinov2 ``interpolate_pos_encoding`` (hub models: offset 0.1, no antialias).
    pos_embed [1, 1+M*M, D] -> [1, 1+gh*gw, D]."""
    N = pos_embed.shape[1] - 1
    M = int(math.sqrt(N))
    assert M * M == N
    D = pos_embed.shape[-1]
    if gh * gw == N and gh == gw:
        return pos_embed
    cls_pos = pos_embed[:, :1]
    patch_pos = pos_embed[:, 1:].reshape(1, M, M, D).permute(0, 3, 1, 2)
    if interpolate_offset:
        # dinov2 names these (sx, sy) from (w0, h0); the tensor is [1, D, M(h), M(w)] and
        # scale_factor applies to (H, W) in that order - the hub code passes (sx, sy) as is.
        sx = float(gw + interpolate_offset) / M
        sy = float(gh + interpolate_offset) / M
        patch_pos = F.interpolate(patch_pos, scale_factor=(sx, sy), mode="bicubic", antialias=False)
    else:
        patch_pos = F.interpolate(patch_pos, size=(gw, gh), mode="bicubic", antialias=False)
    assert patch_pos.shape[-2:] == (gw, gh) or patch_pos.shape[-2:] == (gh, gw)
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, D)
    return torch.cat([cls_pos, patch_pos], dim=1)


def prepare_tokens(sd: Dict[str, torch.Tensor], images: torch.Tensor, prefix: str) -> torch.Tensor:
    """dinov2 ``prepare_tokens_with_masks(x, None)`` (called at dino.py:75)."""
    B, _, H, W = images.shape
    x = F.conv2d(images, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"],
                 stride=PATCH)                       # [B, D, gh, gw]
    gh, gw = x.shape[-2:]
    x = x.flatten(2).transpose(1, 2)                # [B, gh*gw, D]
    cls = sd[prefix + "cls_token"].expand(B, -1, -1)
    x = torch.cat([cls, x], dim=1)
    x = x + interpolate_pos_encoding(sd[prefix + "pos_embed"], gh, gw)
    return x


def attention(x: torch.Tensor, wqkv, bqkv, wproj, bproj, heads: int) -> torch.Tensor:
    """dinov2 ``Attention.forward`` (MemEffAttention computes the same function)."""
    B, T, D = x.shape
    qkv = F.linear(x, wqkv, bqkv).reshape(B, T, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * ((D // heads) ** -0.5), qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, T, D)
    return F.linear(out, wproj, bproj)


def block(x: torch.Tensor, sd, p: str, heads: int) -> torch.Tensor:
    """dinov2 ``Block.forward`` in eval mode: x + ls1(attn(norm1 x)); x + ls2(mlp(norm2 x)).
    LayerNorm eps 1e-6 (nohup.out:604), GELU approximate='none' (nohup.out:617)."""
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    h = attention(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"],
                  sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], heads)
    x = x + h * sd[p + "ls1.gamma"]
    h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    h = F.gelu(h)
    h = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    x = x + h * sd[p + "ls2.gamma"]
    return x


def dino_backbone_forward(sd: Dict[str, torch.Tensor], images: torch.Tensor, heads: int, depth: int,
                          prompt_depth: Optional[torch.Tensor] = None,
                          use_depth_fusion: bool = True,
                          net_prefix: str = "backbone.net.") -> torch.Tensor:
    """DINOBackbone.forward (dino.py:70-120) with output='dense', layer=-1, single layer.
    images [B,3,S,S] normalised+padded -> dense feature [B, D, S/14, S/14]. The final ViT
    LayerNorm is NOT applied (dino.py:88-110 taps the raw block output)."""
    vp = net_prefix + "vit."
    B = images.shape[0]
    gh, gw = images.shape[-2] // PATCH, images.shape[-1] // PATCH
    x = prepare_tokens(sd, images, vp)
    depth_tokens = None
    if use_depth_fusion and prompt_depth is not None:                  # dino.py:83-86
        d = F.interpolate(prompt_depth, size=(gh, gw), mode="bilinear")
        depth_tokens = d.flatten(2).permute(0, 2, 1)                   # [B, gh*gw, 1]
    for i in range(depth):
        x = block(x, sd, vp + f"blocks.{i}.", heads)
        if use_depth_fusion and depth_tokens is not None and i == depth - 1:   # dino.py:91-105
            cls_tok, patch = x[:, :1], x[:, 1:]
            fused = torch.cat([patch.permute(0, 2, 1), depth_tokens.permute(0, 2, 1)], dim=1)
            fused = fused.view(B, -1, gh, gw)
            fused = F.conv2d(fused, sd[net_prefix + "depth_fusion.weight"], sd[net_prefix + "depth_fusion.bias"])
            x = torch.cat([cls_tok, fused.flatten(2).permute(0, 2, 1)], dim=1)
    spatial = x[:, -gh * gw:]                                           # dino.py:112-117
    dense = spatial.reshape(B, gh, gw, -1).permute(0, 3, 1, 2).contiguous()   # tokens_to_output :168-170
    return dense
