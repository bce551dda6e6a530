"""BERT-base text encoder of GroundingDINO (text_encoder_type 'bert-base-uncased', reference
configs/GroundingDINO_SwinB_cfg.py:34) with the sub-sentence attention mask and per-phrase position ids
(cfg:43). Post-LN transformer, eps 1e-12, exact GELU; HF ``BertModel`` parameter names under
``model.text_backbone.``. Sequenced here, computed by libovm3d ops."""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .ops import ACT_GELU, Ops

SPECIAL_TOKENS = [101, 102, 1012, 1029]          # [CLS] [SEP] . ?


def masks_and_position_ids(input_ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """GroundingDINO ``generate_masks_with_special_tokens_and_transfer_map`` (host, integer index work):
    tokens attend inside their own sub-sentence (delimited by special tokens); position ids restart per phrase.
    input_ids [T] -> (mask bool [T,T], position_ids int64 [T])."""
    ids = input_ids.tolist()
    T = len(ids)
    mask = torch.eye(T, dtype=torch.bool)
    pos = torch.zeros(T, dtype=torch.int64)
    special = [i for i, t in enumerate(ids) if t in SPECIAL_TOKENS]
    prev = 0
    for n, col in enumerate(special):
        if col == 0 or col == T - 1:
            mask[col, col] = True
            pos[col] = 0
        else:
            mask[prev + 1: col + 1, prev + 1: col + 1] = True
            pos[prev + 1: col + 1] = torch.arange(0, col - prev)
        prev = col
    return mask, pos


class BertEncoder:
    def __init__(self, ops: Ops, sd: Dict[str, torch.Tensor], prefix: str = "model.text_backbone.", heads: int = 12):
        self.ops, self.heads = ops, heads
        f = ops.f32
        e = prefix + "embeddings."
        self.word, self.pos, self.typ = f(sd[e + "word_embeddings.weight"]), f(sd[e + "position_embeddings.weight"]), f(sd[e + "token_type_embeddings.weight"])
        self.eg, self.eb = f(sd[e + "LayerNorm.weight"]), f(sd[e + "LayerNorm.bias"])
        self.D = self.word.shape[1]
        self.layers = []
        i = 0
        while f"{prefix}encoder.layer.{i}.attention.self.query.weight" in sd:
            p = f"{prefix}encoder.layer.{i}."
            wq, wk, wv = (sd[p + f"attention.self.{n}.weight"] for n in ("query", "key", "value"))
            bq, bk, bv = (sd[p + f"attention.self.{n}.bias"] for n in ("query", "key", "value"))
            self.layers.append(dict(
                qkv=ops.pack(torch.cat([wq, wk, wv], 0), torch.cat([bq, bk, bv], 0)),
                ao=ops.pack(sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"]),
                ag=f(sd[p + "attention.output.LayerNorm.weight"]), ab=f(sd[p + "attention.output.LayerNorm.bias"]),
                fi=ops.pack(sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]),
                fo=ops.pack(sd[p + "output.dense.weight"], sd[p + "output.dense.bias"]),
                og=f(sd[p + "output.LayerNorm.weight"]), ob=f(sd[p + "output.LayerNorm.bias"])))
            i += 1

    def forward(self, input_ids: torch.Tensor, attn_mask: torch.Tensor, position_ids: torch.Tensor,
                token_type_ids: torch.Tensor = None, bias: torch.Tensor = None) -> torch.Tensor:
        """input_ids [T] int64; attn_mask bool [T,T]; returns last hidden state [T, D] (device fp32)."""
        o = self.ops
        T, D, H = int(input_ids.shape[0]), self.D, self.heads
        dh = D // H
        ids = input_ids.to(torch.int32).view(T, 1)
        tt = (token_type_ids if token_type_ids is not None else torch.zeros_like(ids)).to(torch.int32).view(T, 1)
        x = o.gather_rows(self.word, ids)
        x = o.add(x, o.gather_rows(self.pos, position_ids.to(torch.int32).view(T, 1)))
        x = o.add(x, o.gather_rows(self.typ, tt))
        x = o.layernorm(x, self.eg, self.eb, 1e-12)
        if bias is None:
            bias = torch.where(attn_mask, 0.0, torch.finfo(torch.float32).min).to(o.dev, torch.float32).contiguous()   # mask -> additive
        for ly in self.layers:
            qkv = o.linear(x, ly["qkv"])                                             # [T, 3D]
            s = o.empty(H, T, T)
            o.bmm_raw(qkv, 0, qkv, D, s, 0, H, T, T, dh, 3 * D, 3 * D, T, dh, dh, T * T, True, dh ** -0.5)
            o.softmax_(s, bias, bias_rows=T, bias_div=1)
            ctx = o.empty(T, D)
            o.bmm_raw(s, 0, qkv, 2 * D, ctx, 0, H, T, dh, T, T, 3 * D, D, T * T, dh, dh, False, 1.0)
            a = o.linear(ctx, ly["ao"])
            x = o.layernorm(a, ly["ag"], ly["ab"], 1e-12, residual=x)
            h = o.linear(x, ly["fi"], act=ACT_GELU)
            f2 = o.linear(h, ly["fo"])
            x = o.layernorm(f2, ly["og"], ly["ob"], 1e-12, residual=x)
        return x
