"""Hugging Face ViTMAE encoder forward as the reference's MAEBackbone runs it (fp32, CPU). TEST INFRASTRUCTURE: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg only.

Follows reference cubercnn/modeling/backbone/mae.py:62-118 (``resize_pos_embed``: the position table is REPLACED by a 2-D sin-cos table
for the input grid; ``embed_forward``: patch projection + table rows 1.., class token + row 0, no masking; the encoder with
``output_hidden_states=True``; tap ``hidden_states[num_layers - 1]`` = the state after num_layers - 1 blocks, :43-55,110-116) and
:152-180 (``get_2d_sincos_pos_embed`` over transformers' ``get_2d_sincos_pos_embed_from_grid`` - transformers==4.46.3 in
requirements.txt, the numpy form, restated from the published MAE definition because the installed 5.x no longer ships it). The encoder
layer is HF ``ViTLayer``: x + attn(layernorm_before x), x + mlp(layernorm_after x), separate query / key / value linears, erf-GELU, LayerNorm
eps 1e-12 (ViTMAEConfig). Cross-checked against ``ViTMAEModel.encoder`` in tests/test_oracle_crosscheck.py. Parity unpinned vs the
reference itself (no fixture for this config).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    omega = np.arange(embed_dim // 2, dtype=float)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim: int, gh: int, gw: int, add_cls_token: bool = True) -> np.ndarray:
    """mae.py:152-180: meshgrid(grid_w, grid_h) - the x coordinate comes first and feeds the first half of the channels."""
    grid = np.stack(np.meshgrid(np.arange(gw, dtype=np.float32), np.arange(gh, dtype=np.float32)), axis=0).reshape(2, 1, gh, gw)
    emb = np.concatenate([sincos_1d(embed_dim // 2, grid[0]), sincos_1d(embed_dim // 2, grid[1])], axis=1)
    return np.concatenate([np.zeros([1, embed_dim]), emb], axis=0) if add_cls_token else emb


def vit_layer(x: torch.Tensor, sd, p: str, heads: int, eps: float = 1e-12) -> torch.Tensor:
    B, T, D = x.shape
    dh = D // heads
    h = F.layer_norm(x, (D,), sd[p + "layernorm_before.weight"], sd[p + "layernorm_before.bias"], eps)
    a = p + "attention.attention."
    q = F.linear(h, sd[a + "query.weight"], sd[a + "query.bias"]).reshape(B, T, heads, dh).transpose(1, 2)
    k = F.linear(h, sd[a + "key.weight"], sd[a + "key.bias"]).reshape(B, T, heads, dh).transpose(1, 2)
    v = F.linear(h, sd[a + "value.weight"], sd[a + "value.bias"]).reshape(B, T, heads, dh).transpose(1, 2)
    ctx = ((q @ k.transpose(-2, -1)) / (dh ** 0.5)).softmax(dim=-1) @ v
    ctx = ctx.transpose(1, 2).reshape(B, T, D)
    x = x + F.linear(ctx, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
    h = F.layer_norm(x, (D,), sd[p + "layernorm_after.weight"], sd[p + "layernorm_after.bias"], eps)
    h = F.gelu(F.linear(h, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
    return x + F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])


def mae_backbone_forward(sd: Dict[str, torch.Tensor], images: torch.Tensor, heads: int, num_layers: int,
                         prefix: str = "backbone.net.vit.") -> torch.Tensor:
    """MAEBackbone.forward with output='dense', layer=-1: images [B,3,S,S] -> [B, D, S/P, S/P] from hidden_states[num_layers - 1]."""
    w = sd[prefix + "embeddings.patch_embeddings.projection.weight"]
    P = w.shape[-1]
    x = F.conv2d(images, w, sd[prefix + "embeddings.patch_embeddings.projection.bias"], stride=P)
    gh, gw = x.shape[-2:]
    x = x.flatten(2).transpose(1, 2)
    pos = torch.from_numpy(sincos_2d(w.shape[0], gh, gw)).float()[None]           # :62-78
    x = x + pos[:, 1:]
    cls = (sd[prefix + "embeddings.cls_token"].reshape(1, 1, -1) + pos[:, :1]).expand(x.shape[0], -1, -1)
    x = torch.cat([cls, x], dim=1)
    for i in range(num_layers - 1):                                               # hidden_states[num_layers - 1]
        x = vit_layer(x, sd, prefix + f"encoder.layer.{i}.", heads)
    return x[:, 1:].reshape(x.shape[0], gh, gw, -1).permute(0, 3, 1, 2).contiguous()
