"""Architecture record of the GroundingDINO network the engine runs (defaults = reference configs/GroundingDINO_SwinB_cfg.py:
Swin-B, BERT-base, 6 / 6 layers, 900 queries; IDEA-Research/GroundingDINO @856dde2)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence


@dataclass
class GDinoConfig:
    d_model: int = 256
    enc_layers: int = 6
    dec_layers: int = 6
    heads: int = 8
    ffn_dim: int = 2048
    n_levels: int = 4
    n_points: int = 4
    num_queries: int = 900
    max_text_len: int = 256
    pe_temperature: float = 20.0
    eps: float = 1e-5
    bert_heads: int = 12
    swin_embed: int = 128
    swin_depths: Sequence[int] = (2, 2, 18, 2)
    swin_heads: Sequence[int] = (4, 8, 16, 32)
    swin_window: int = 12
