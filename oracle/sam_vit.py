"""SAM image-encoder forward as the reference's SAMBackbone runs it (fp32, CPU). TEST INFRASTRUCTURE: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg only.

Follows reference cubercnn/modeling/backbone/sam.py:73-112: ``resize_pos_embed`` (plain bicubic to the input grid), patch embedding + position
table, every block, dense output of the last one (NHWC -> NCHW), neck unused. The blocks are segment_anything's ``ImageEncoderViT`` blocks
(``sam_model_registry['vit_b']``; source absent from the container, restated from the published definition): norm1 -> window partition with
zero padding at the bottom / right (windowed blocks) -> attention -> un-partition -> + shortcut; + mlp(norm2 x). Attention: fused qkv linear,
scores (q * dh^-0.5) k^T plus the decomposed relative-position bias rel_h[q, kh] + rel_w[q, kw] computed from the UNSCALED query and the
(linearly resized) tables, softmax, proj. LayerNorm eps 1e-6, erf-GELU. Cross-checked against Hugging Face ``SamVisionLayer`` in
tests/test_oracle_crosscheck.py. Parity unpinned vs the reference itself.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F


def get_rel_pos(q_size: int, k_size: int, rel_pos: torch.Tensor) -> torch.Tensor:
    """[L, C] table -> [q_size, k_size, C]; the table is linearly resized to 2 max(q, k) - 1 entries when its length differs."""
    max_rel = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel:
        rel_pos = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel, mode="linear")
        rel_pos = rel_pos.reshape(-1, max_rel).permute(1, 0)
    q = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    return rel_pos[((q - k) + (k_size - 1) * max(q_size / k_size, 1.0)).long()]


def attention(x: torch.Tensor, sd, p: str, heads: int) -> torch.Tensor:
    """x [B, H, W, C] -> [B, H, W, C]"""
    B, H, W, C = x.shape
    dh = C // heads
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(B, H * W, 3, heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * heads, H * W, dh).unbind(0)
    attn = (q * dh ** -0.5) @ k.transpose(-2, -1)
    rq = q.reshape(B * heads, H, W, dh)
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, get_rel_pos(H, H, sd[p + "rel_pos_h"]))
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, get_rel_pos(W, W, sd[p + "rel_pos_w"]))
    attn = (attn.view(B * heads, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(B * heads, H * W, H * W)
    out = (attn.softmax(dim=-1) @ v).view(B, heads, H, W, dh).permute(0, 2, 3, 1, 4).reshape(B, H, W, C)
    return F.linear(out, sd[p + "proj.weight"], sd[p + "proj.bias"])


def block(x: torch.Tensor, sd, p: str, heads: int, window: int) -> torch.Tensor:
    B, H, W, C = x.shape
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    if window > 0:
        ph, pw = (window - H % window) % window, (window - W % window) % window
        h = F.pad(h, (0, 0, 0, pw, 0, ph))
        Hp, Wp = H + ph, W + pw
        h = h.view(B, Hp // window, window, Wp // window, window, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, window, window, C)
    h = attention(h, sd, p + "attn.", heads)
    if window > 0:
        h = h.view(B, Hp // window, Wp // window, window, window, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)[:, :H, :W]
    x = x + h
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"])), sd[p + "mlp.lin2.weight"], sd[p + "mlp.lin2.bias"])
    return x + h


def sam_backbone_forward(sd: Dict[str, torch.Tensor], images: torch.Tensor, heads: int, depth: int, window: int, global_blocks: Sequence[int],
                         prefix: str = "backbone.net.vit.") -> torch.Tensor:
    w = sd[prefix + "patch_embed.proj.weight"]
    P = w.shape[-1]
    x = F.conv2d(images, w, sd[prefix + "patch_embed.proj.bias"], stride=P).permute(0, 2, 3, 1)     # [B, gh, gw, C]
    gh, gw = x.shape[1:3]
    pos = sd[prefix + "pos_embed"]
    if tuple(pos.shape[1:3]) != (gh, gw):                                        # sam.py:73-86 (plain bicubic)
        pos = F.interpolate(pos.permute(0, 3, 1, 2), size=(gh, gw), mode="bicubic").permute(0, 2, 3, 1)
    x = x + pos
    for i in range(depth):
        x = block(x, sd, prefix + f"blocks.{i}.", heads, 0 if i in global_blocks else window)
    return x.permute(0, 3, 1, 2).contiguous()
