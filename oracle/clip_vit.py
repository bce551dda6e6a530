"""open_clip ViT image tower forward as the reference's CLIPBackbone runs it (fp32, CPU). TEST INFRASTRUCTURE: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.

Follows reference cubercnn/modeling/backbone/clip.py:62-96 (forward: conv1, class embedding, resized positional embedding,
ln_pre, the residual blocks up to the tapped one, ``tokens_to_output('dense')``, no ln_post / proj) and :98-133
(``resize_pos_embed``: antialiased bicubic). The tower itself is open_clip_torch==2.30.0 (requirements.txt:80; source not in
the container): ``VisionTransformer`` with ``ResidualAttentionBlock``s - x + attn(ln_1 x), x + mlp(ln_2 x), nn.MultiheadAttention
(in_proj_weight / in_proj_bias / out_proj), LayerNorm eps 1e-5, no LayerScale, and QuickGELU ``x * sigmoid(1.702 x)`` for the
OpenAI weights the reference loads (``checkpoint='openai'``, clip.py:19). Restated from the published model definition and
cross-checked against Hugging Face ``CLIPVisionModel`` in tests/test_oracle_crosscheck.py. Parity unpinned vs the reference
itself: it holds no fixture or output for this config.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F


def resize_pos_embed(pos_embed: torch.Tensor, hw: Tuple[int, int]) -> torch.Tensor:
    """clip.py:98-133 with has_cls_token=True: [1 + M*M, D] -> [1 + h*w, D]; unchanged when the grid already matches."""
    n_grid = pos_embed.shape[0] - 1
    if n_grid == hw[0] * hw[1]:
        return pos_embed
    cls_embed, grid = pos_embed[:1], pos_embed[1:]
    m = int(grid.shape[0] ** 0.5)
    grid = grid.reshape(m, m, -1).permute(2, 0, 1)[None]                      # "(h w) c -> 1 c h w"
    grid = F.interpolate(grid, hw, mode="bicubic", align_corners=False, antialias=True)
    grid = grid[0].permute(1, 2, 0).reshape(hw[0] * hw[1], -1)
    return torch.cat([cls_embed, grid], dim=0)


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(1.702 * x)


def residual_block(x: torch.Tensor, sd, p: str, heads: int) -> torch.Tensor:
    """open_clip ResidualAttentionBlock.forward (ls_1 / ls_2 are Identity)."""
    B, T, D = x.shape
    dh = D // heads
    h = F.layer_norm(x, (D,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-5)
    q, k, v = F.linear(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"]).chunk(3, dim=-1)
    q = q.reshape(B, T, heads, dh).transpose(1, 2) * (dh ** -0.5)
    k = k.reshape(B, T, heads, dh).transpose(1, 2)
    v = v.reshape(B, T, heads, dh).transpose(1, 2)
    a = (q @ k.transpose(-2, -1)).softmax(dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, T, D)
    x = x + F.linear(a, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
    h = F.layer_norm(x, (D,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], 1e-5)
    h = quick_gelu(F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]))
    return x + F.linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])


def clip_backbone_forward(sd: Dict[str, torch.Tensor], images: torch.Tensor, heads: int, depth: int,
                          prefix: str = "backbone.net.visual.") -> torch.Tensor:
    """CLIPBackbone.forward with output='dense', layer=-1: images [B,3,S,S] normalised + padded -> [B, D, S/P, S/P]."""
    w = sd[prefix + "conv1.weight"]
    P = w.shape[-1]
    x = F.conv2d(images, w, None, stride=P)                                   # clip.py:66 (conv1 has no bias)
    gh, gw = x.shape[-2:]
    x = x.flatten(2).transpose(1, 2)
    cls = sd[prefix + "class_embedding"].reshape(1, 1, -1).expand(x.shape[0], -1, -1)
    x = torch.cat([cls, x], dim=1)                                            # :71-73
    x = x + resize_pos_embed(sd[prefix + "positional_embedding"], (gh, gw))   # :76-77
    x = F.layer_norm(x, (x.shape[-1],), sd[prefix + "ln_pre.weight"], sd[prefix + "ln_pre.bias"], 1e-5)
    for i in range(depth):                                                    # :81-87, tap = last block
        x = residual_block(x, sd, prefix + f"transformer.resblocks.{i}.", heads)
    dense = x[:, 1:]                                                          # :91 tokens_to_output('dense', x[:, 1:], ...)
    return dense.reshape(x.shape[0], gh, gw, -1).permute(0, 3, 1, 2).contiguous()
