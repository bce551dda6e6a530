"""Native GroundingDINO forward (batch 1, one caption): BERT text encoder -> Swin backbone -> input projections ->
6 x (image<->text fusion, text enhancer, multi-scale deformable self-attention) -> two-stage query selection ->
6 x decoder (self-attn, text cross-attn, deformable cross-attn, FFN, iterative box refinement) -> contrastive
class logits + boxes of the last decoder layer.

This is the network ``ROIHeads3DGDINO`` calls at reference cubercnn/modeling/roi_heads/roi_heads_gdino.py:186 with the
configuration of reference configs/GroundingDINO_SwinB_cfg.py (IDEA-Research/GroundingDINO @856dde2; source not in the
reference tree). Module structure and parameter names follow the Hugging Face port (``GroundingDinoForObjectDetection``),
the independent CPU implementation the parity test compares against. The host sequences ops; all arithmetic is in libovm3d.

TEST INFRASTRUCTURE (tests/pyref_gdino): the round-1 Python-sequenced form of the detector, kept as an independent cross-check of the
C++ engine behind ovm_gdino_forward (ovmono3d_amd/gdino/engine.py). Not part of the product package; nothing in ovmono3d_amd imports it.
Constant tables that depend only on tensor shapes (sine position embeddings, reference grids, proposal grids, index maps)
are built once per input size on the host.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ovmono3d_amd.gdino.config import GDinoConfig  # noqa: F401

from . import ops as O
from .bert import BertEncoder, masks_and_position_ids
from .ops import ACT_RELU, Ops
from .swin import SwinBackbone


def _sine_pos(h: int, w: int, d_half: int, temperature: float) -> torch.Tensor:
    """GroundingDINO PositionEmbeddingSineHW with an all-valid mask -> [h*w, 2*d_half] (pos_y | pos_x)."""
    ones = torch.ones(1, h, w, dtype=torch.float32)
    y_embed, x_embed = ones.cumsum(1), ones.cumsum(2)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(d_half, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / d_half)
    pos_x, pos_y = x_embed[:, :, :, None] / dim_t, y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).reshape(h * w, 2 * d_half)


class _MHA:
    """GroundingDinoMultiheadAttention (query/key/value/out_proj Linear, softmax(qk/sqrt(dh)+mask)v)."""

    def __init__(self, ops: Ops, sd, p: str, heads: int):
        self.o, self.h = ops, heads
        self.q = ops.pack(sd[p + "query.weight"], sd[p + "query.bias"])
        self.k = ops.pack(sd[p + "key.weight"], sd[p + "key.bias"])
        self.v = ops.pack(sd[p + "value.weight"], sd[p + "value.bias"])
        self.out = ops.pack(sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])

    def __call__(self, q_in, k_in, v_in, bias=None, residual=None):
        o, H = self.o, self.h
        q, k, v = o.linear(q_in, self.q), o.linear(k_in, self.k), o.linear(v_in, self.v)
        Tq, Tk, D = q.shape[0], k.shape[0], q.shape[1]
        dh = D // H
        s = o.empty(H, Tq, Tk)
        o.bmm_raw(q, 0, k, 0, s, 0, H, Tq, Tk, dh, D, D, Tk, dh, dh, Tq * Tk, True, 1.0 / math.sqrt(dh))
        o.softmax_(s, bias, bias_rows=Tq if bias is not None else 1, bias_div=1)
        ctx = o.empty(Tq, D)
        o.bmm_raw(s, 0, v, 0, ctx, 0, H, Tq, dh, Tk, Tk, D, D, Tq * Tk, dh, dh, False, 1.0)
        return o.linear(ctx, self.out, residual=residual)


class _MSDeform:
    """GroundingDinoMultiscaleDeformableAttention."""

    def __init__(self, ops: Ops, sd, p: str, heads: int, levels: int, points: int):
        self.o, self.h, self.L, self.P = ops, heads, levels, points
        w = torch.cat([sd[p + "sampling_offsets.weight"], sd[p + "attention_weights.weight"]], 0)
        b = torch.cat([sd[p + "sampling_offsets.bias"], sd[p + "attention_weights.bias"]], 0)
        self.offw = ops.pack(w, b)                                  # one projection for offsets | attention logits
        self.value = ops.pack(sd[p + "value_proj.weight"], sd[p + "value_proj.bias"])
        self.out = ops.pack(sd[p + "output_proj.weight"], sd[p + "output_proj.bias"])

    def __call__(self, query, value_src, shapes, loc_fn, residual):
        """query [Q,D] (position embedding already added); value_src [S,D]; loc_fn(offsets [Q, H*L*P*2]) -> locations."""
        o, H, L, P = self.o, self.h, self.L, self.P
        D = query.shape[1]
        Q, S = query.shape[0], value_src.shape[0]
        val = o.linear(value_src, self.value)                        # [S, D] = [S, H, dh]
        ow = o.linear(query, self.offw)                              # [Q, H*L*P*2 + H*L*P]
        n_off = H * L * P * 2
        off = ow[:, :n_off].contiguous()
        aw = ow[:, n_off:].contiguous().view(Q * H, L * P)
        o.softmax_(aw)
        loc = loc_fn(off)
        out = o.msdeform(val.view(1, S, H, D // H), shapes, loc.view(1, Q, H, L, P, 2), aw.view(1, Q, H, L, P))
        return o.linear(out.view(Q, D), self.out, residual=residual)


class GroundingDinoNative:
    def __init__(self, ops: Ops, sd: Dict[str, torch.Tensor], cfg: GDinoConfig = GDinoConfig()):
        self.o, self.cfg = ops, cfg
        o, f, c = ops, ops.f32, cfg
        M = "model."
        self.bert = BertEncoder(o, sd, M + "text_backbone.", heads=c.bert_heads)
        self.text_proj = o.pack(sd[M + "text_projection.weight"], sd[M + "text_projection.bias"])
        self.swin = SwinBackbone(o, sd, M + "backbone.conv_encoder.model.", c.swin_embed, c.swin_depths, c.swin_heads, c.swin_window)
        self.in_proj = []
        for l in range(c.n_levels):
            w = sd[M + f"input_proj_vision.{l}.0.weight"]
            if w.shape[-1] == 3:
                w = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)     # [256, (ky,kx,c)]
            self.in_proj.append(dict(w=o.pack(w, sd[M + f"input_proj_vision.{l}.0.bias"]), k=int(sd[M + f"input_proj_vision.{l}.0.weight"].shape[-1]),
                                     g=f(sd[M + f"input_proj_vision.{l}.1.weight"]), b=f(sd[M + f"input_proj_vision.{l}.1.bias"])))
        self.level_embed = sd[M + "level_embed"].float()
        self.enc = []
        for i in range(c.enc_layers):
            p = M + f"encoder.layers.{i}."
            fu, te, de = p + "fusion_layer.", p + "text_enhancer_layer.", p + "deformable_layer."
            gv, gt = sd[fu + "vision_param"].float(), sd[fu + "text_param"].float()
            self.enc.append(dict(
                lnv=(f(sd[fu + "layer_norm_vision.weight"]), f(sd[fu + "layer_norm_vision.bias"])),
                lnt=(f(sd[fu + "layer_norm_text.weight"]), f(sd[fu + "layer_norm_text.bias"])),
                vq=o.pack(sd[fu + "attn.vision_proj.weight"], sd[fu + "attn.vision_proj.bias"]),
                tk=o.pack(sd[fu + "attn.text_proj.weight"], sd[fu + "attn.text_proj.bias"]),
                vv=o.pack(sd[fu + "attn.values_vision_proj.weight"], sd[fu + "attn.values_vision_proj.bias"]),
                tv=o.pack(sd[fu + "attn.values_text_proj.weight"], sd[fu + "attn.values_text_proj.bias"]),
                # layer-scale (vision_param / text_param) folded into the output projections
                ov=o.pack(sd[fu + "attn.out_vision_proj.weight"].float() * gv[:, None], sd[fu + "attn.out_vision_proj.bias"].float() * gv),
                ot=o.pack(sd[fu + "attn.out_text_proj.weight"].float() * gt[:, None], sd[fu + "attn.out_text_proj.bias"].float() * gt),
                te_attn=_MHA(o, sd, te + "self_attn.", c.heads // 2),
                te_ln1=(f(sd[te + "layer_norm_before.weight"]), f(sd[te + "layer_norm_before.bias"])),
                te_ln2=(f(sd[te + "layer_norm_after.weight"]), f(sd[te + "layer_norm_after.bias"])),
                te_fc1=o.pack(sd[te + "fc1.weight"], sd[te + "fc1.bias"]), te_fc2=o.pack(sd[te + "fc2.weight"], sd[te + "fc2.bias"]),
                msda=_MSDeform(o, sd, de + "self_attn.", c.heads, c.n_levels, c.n_points),
                de_ln1=(f(sd[de + "self_attn_layer_norm.weight"]), f(sd[de + "self_attn_layer_norm.bias"])),
                de_ln2=(f(sd[de + "final_layer_norm.weight"]), f(sd[de + "final_layer_norm.bias"])),
                de_fc1=o.pack(sd[de + "fc1.weight"], sd[de + "fc1.bias"]), de_fc2=o.pack(sd[de + "fc2.weight"], sd[de + "fc2.bias"])))
        self.enc_output = o.pack(sd[M + "enc_output.weight"], sd[M + "enc_output.bias"])
        self.enc_output_ln = (f(sd[M + "enc_output_norm.weight"]), f(sd[M + "enc_output_norm.bias"]))
        self.enc_bbox = [o.pack(sd[M + f"encoder_output_bbox_embed.layers.{k}.weight"], sd[M + f"encoder_output_bbox_embed.layers.{k}.bias"]) for k in range(3)]
        self.tgt = f(sd[M + "query_position_embeddings.weight"])
        self.dec = []
        for i in range(c.dec_layers):
            p = M + f"decoder.layers.{i}."
            self.dec.append(dict(
                sa=_MHA(o, sd, p + "self_attn.", c.heads), ln1=(f(sd[p + "self_attn_layer_norm.weight"]), f(sd[p + "self_attn_layer_norm.bias"])),
                ca=_MHA(o, sd, p + "encoder_attn_text.", c.heads),
                ln2=(f(sd[p + "encoder_attn_text_layer_norm.weight"]), f(sd[p + "encoder_attn_text_layer_norm.bias"])),
                msda=_MSDeform(o, sd, p + "encoder_attn.", c.heads, c.n_levels, c.n_points),
                ln3=(f(sd[p + "encoder_attn_layer_norm.weight"]), f(sd[p + "encoder_attn_layer_norm.bias"])),
                fc1=o.pack(sd[p + "fc1.weight"], sd[p + "fc1.bias"]), fc2=o.pack(sd[p + "fc2.weight"], sd[p + "fc2.bias"]),
                ln4=(f(sd[p + "final_layer_norm.weight"]), f(sd[p + "final_layer_norm.bias"]))))
        self.dec_ln = (f(sd[M + "decoder.layer_norm.weight"]), f(sd[M + "decoder.layer_norm.bias"]))
        self.ref_head = [o.pack(sd[M + f"decoder.reference_points_head.layers.{k}.weight"], sd[M + f"decoder.reference_points_head.layers.{k}.bias"]) for k in range(2)]
        self.bbox = [[o.pack(sd[f"bbox_embed.{i}.layers.{k}.weight"], sd[f"bbox_embed.{i}.layers.{k}.bias"]) for k in range(3)] for i in range(c.dec_layers)]
        # constant selection matrices for the 4-d reference -> per-(head, level, point, xy) expansion of the decoder
        n = c.heads * c.n_levels * c.n_points * 2
        e_ctr, e_wh = torch.zeros(4, n), torch.zeros(4, n)
        for j in range(n):
            e_ctr[j % 2, j] = 1.0
            e_wh[2 + j % 2, j] = 0.5 / c.n_points
        self.e_ctr, self.e_wh = f(e_ctr), f(e_wh)
        self._shape_cache = {}
        self._caption_cache = {}

    # ---- helpers --------------------------------------------------------------------------------
    def _mlp(self, x, layers):
        for i, w in enumerate(layers):
            x = self.o.linear(x, w, act=ACT_RELU if i < len(layers) - 1 else 0)
        return x

    def _shape_tables(self, shapes):
        key = tuple(shapes)
        if key in self._shape_cache:
            return self._shape_cache[key]
        c, o = self.cfg, self.o
        pos, ref, prop = [], [], []
        for l, (h, w) in enumerate(shapes):
            pos.append(_sine_pos(h, w, c.d_model // 2, c.pe_temperature) + self.level_embed[l].view(1, -1))
            ry, rx = torch.meshgrid(torch.linspace(0.5, h - 0.5, h, dtype=torch.float32), torch.linspace(0.5, w - 0.5, w, dtype=torch.float32), indexing="ij")
            ref.append(torch.stack((rx.reshape(-1) / w, ry.reshape(-1) / h), -1))                        # valid ratios are 1 (no padding)
            gy, gx = torch.meshgrid(torch.linspace(0, h - 1, h, dtype=torch.float32), torch.linspace(0, w - 1, w, dtype=torch.float32), indexing="ij")
            grid = (torch.stack((gx, gy), -1) + 0.5) / torch.tensor([w, h], dtype=torch.float32)
            wh = torch.ones_like(grid) * 0.05 * (2.0 ** l)
            prop.append(torch.cat((grid, wh), -1).view(-1, 4))
        ref = torch.cat(ref, 0)                                                                          # [S, 2]
        S = ref.shape[0]
        H, L, P = c.heads, c.n_levels, c.n_points
        ref_exp = ref.view(S, 1, 1, 1, 2).expand(S, H, L, P, 2).reshape(S, -1)                            # same point for every level
        norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)                            # offset normaliser (w, h) per level
        inv_norm = (1.0 / norm).view(1, L, 1, 2).expand(H, L, P, 2).reshape(-1)
        prop = torch.cat(prop, 0)
        valid = ((prop > 0.01) & (prop < 0.99)).all(-1)
        prop_logit = torch.log(prop / (1 - prop))
        prop_logit[~valid] = float("inf")
        valid_idx = torch.where(valid, torch.arange(S), torch.full((S,), -1)).to(torch.int32).view(S, 1)
        t = dict(pos=o.f32(torch.cat(pos, 0)), ref_exp=o.f32(ref_exp), inv_norm=o.f32(inv_norm), prop_logit=o.f32(prop_logit),
                 valid_idx=valid_idx.to(o.dev), S=S)
        self._shape_cache[key] = t
        return t

    # ---- forward ----------------------------------------------------------------------------------
    def forward(self, img_nhwc: torch.Tensor, H: int, W: int, input_ids: torch.Tensor, position_ids: Optional[torch.Tensor] = None,
                return_aux: bool = False, force_topk: Optional[torch.Tensor] = None):
        # force_topk (tests only): use this two-stage selection instead of the network's own, to compare decoders across near-ties
        """img_nhwc: device fp32 [H*W, 3] (normalised image as the reference hands it over, roi_heads_gdino.py:146);
        input_ids: int64 [T] = tokenizer(caption). Returns pred_logits [Q, max_text_len] (pre-sigmoid; -inf beyond the
        caption) and pred_boxes [Q, 4] (cx, cy, w, h in [0, 1])."""
        o, c = self.o, self.cfg
        D = c.d_model
        aux = {}
        # -- text. Everything derived from the caption alone is uploaded once per caption (no host->device traffic on
        # later calls, which also keeps the forward capturable into a HIP graph)
        T = int(input_ids.shape[0])
        ck = (tuple(int(i) for i in input_ids.tolist()), None if position_ids is None else tuple(int(i) for i in position_ids.tolist()))
        cap = self._caption_cache.get(ck)
        if cap is None:
            mask, pos_ids = masks_and_position_ids(input_ids)
            if position_ids is not None:
                pos_ids = position_ids
            cap = dict(ids=input_ids.to(o.dev, torch.int32).view(T, 1), mask=mask.to(o.dev), pos=pos_ids.to(o.dev, torch.int32).view(T, 1),
                       bias=torch.where(mask, 0.0, torch.finfo(torch.float32).min).to(o.dev, torch.float32).contiguous(),
                       pos_f=pos_ids.to(torch.float32).view(T, 1).to(o.dev))
            if len(self._caption_cache) > 64:
                self._caption_cache.clear()
            self._caption_cache[ck] = cap
        text = o.linear(self.bert.forward(cap["ids"], cap["mask"], cap["pos"], bias=cap["bias"]), self.text_proj)       # [T, D]
        text_bias = cap["bias"]
        text_pos = o.sine_embed(cap["pos_f"], D, 10000.0)                                                # [T, D]
        # -- image features, 4 levels
        feats = self.swin.forward(img_nhwc, H, W)
        srcs, shapes = [], []
        for l, (fm, h, w) in enumerate(feats):
            ip = self.in_proj[l]
            srcs.append(o.groupnorm(o.linear(fm, ip["w"]).view(1, h * w, D), 32, ip["g"], ip["b"], 1e-5).view(h * w, D))
            shapes.append((h, w))
        fm, h, w = feats[-1]
        for l in range(len(feats), c.n_levels):                                                           # 3x3 stride-2 conv levels
            ip = self.in_proj[l]
            h2, w2 = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
            key = ("conv3s2", h, w)
            if key not in self._shape_cache:
                oy, ox = np.meshgrid(np.arange(h2), np.arange(w2), indexing="ij")
                idx = np.full((h2 * w2, 9), -1, dtype=np.int32)
                for ky in range(3):
                    for kx in range(3):
                        y, x = 2 * oy + ky - 1, 2 * ox + kx - 1
                        idx[:, ky * 3 + kx] = np.where((y >= 0) & (y < h) & (x >= 0) & (x < w), y * w + x, -1).reshape(-1)
                self._shape_cache[key] = torch.from_numpy(idx).to(o.dev)
            src_in = fm if l == len(feats) else srcs[-1]
            cols = o.gather_rows(src_in, self._shape_cache[key])
            srcs.append(o.groupnorm(o.linear(cols, ip["w"]).view(1, h2 * w2, D), 32, ip["g"], ip["b"], 1e-5).view(h2 * w2, D))
            shapes.append((h2, w2))
            fm, h, w = srcs[-1], h2, w2
        tb = self._shape_tables(shapes)
        S = tb["S"]
        vis = torch.cat(srcs, 0)                                                                          # [S, D] (device concat = copies)
        if return_aux:
            aux["text_features"], aux["source_flatten"] = text, vis
        # -- encoder
        HF, dhf = c.heads // 2, (c.ffn_dim // 2) // (c.heads // 2)
        E = c.ffn_dim // 2
        for ly in self.enc:
            v = o.layernorm(vis, ly["lnv"][0], ly["lnv"][1], c.eps)
            t = o.layernorm(text, ly["lnt"][0], ly["lnt"][1], c.eps)
            q, k = o.linear(v, ly["vq"]), o.linear(t, ly["tk"])                                           # [S,E], [T,E]
            vv, tv = o.linear(v, ly["vv"]), o.linear(t, ly["tv"])
            a_v = o.empty(HF, S, T)
            o.bmm_raw(q, 0, k, 0, a_v, 0, HF, S, T, dhf, E, E, T, dhf, dhf, S * T, True, dhf ** -0.5)
            a_t = o.empty(HF, T, S)
            o.bmm_raw(k, 0, q, 0, a_t, 0, HF, T, S, dhf, E, E, S, dhf, dhf, T * S, True, dhf ** -0.5)
            o.softmax_(a_v)
            o.softmax_(a_t)
            cv, ct = o.empty(S, E), o.empty(T, E)
            o.bmm_raw(a_v, 0, tv, 0, cv, 0, HF, S, dhf, T, T, E, E, S * T, dhf, dhf, False, 1.0)
            o.bmm_raw(a_t, 0, vv, 0, ct, 0, HF, T, dhf, S, S, E, E, T * S, dhf, dhf, False, 1.0)
            vis = o.linear(cv, ly["ov"], residual=v)
            text = o.linear(ct, ly["ot"], residual=t)
            # text enhancer
            qk = o.add(text, text_pos)
            text = o.layernorm(ly["te_attn"](qk, qk, text, bias=text_bias, residual=text), ly["te_ln1"][0], ly["te_ln1"][1], c.eps)
            ff = o.linear(o.linear(text, ly["te_fc1"], act=ACT_RELU), ly["te_fc2"], residual=text)
            text = o.layernorm(ff, ly["te_ln2"][0], ly["te_ln2"][1], c.eps)
            # deformable self-attention over the image tokens
            qd = o.add(vis, tb["pos"])
            loc_fn = lambda off: o.add(o.elt(O.MUL, off, tb["inv_norm"]), tb["ref_exp"])
            vis = o.layernorm(ly["msda"](qd, vis, shapes, loc_fn, residual=vis), ly["de_ln1"][0], ly["de_ln1"][1], c.eps)
            ff = o.linear(o.linear(vis, ly["de_fc1"], act=ACT_RELU), ly["de_fc2"], residual=vis)
            vis = o.layernorm(ff, ly["de_ln2"][0], ly["de_ln2"][1], c.eps)
        if return_aux:
            aux["enc_vision"], aux["enc_text"] = vis, text
        # -- two-stage query selection
        oq = o.gather_rows(vis, tb["valid_idx"])                                                          # invalid proposals -> zero rows
        oq = o.layernorm(o.linear(oq, self.enc_output), self.enc_output_ln[0], self.enc_output_ln[1], c.eps)
        cls = o.bmm(oq.view(1, S, D), text.view(1, T, D), True)[0]                                        # [S, T]
        if S < c.num_queries:                         # torch.topk in the upstream two-stage selection raises the same way
            raise RuntimeError(f"selected index k out of range: {S} encoder tokens < {c.num_queries} queries (image too small)")
        topk = o.topk(o.rowmax(cls), c.num_queries) if force_topk is None else force_topk.to(o.dev, torch.int32)
        coord = o.add(self._mlp(oq, self.enc_bbox), tb["prop_logit"])
        ref = o.elt(O.SIGMOID, o.gather_rows(coord, topk.view(-1, 1)))                                    # [Q, 4]
        hs = self.tgt                                                                                     # embedding_init_target
        Q = c.num_queries
        if return_aux:
            aux["topk"], aux["init_ref"] = topk, ref
        # -- decoder
        n_off = c.heads * c.n_levels * c.n_points * 2
        for i, ly in enumerate(self.dec):
            qpos = self._mlp(o.sine_embed(ref, D // 2, 10000.0), self.ref_head)                           # [Q, D]
            qk = o.add(hs, qpos)
            hs = o.layernorm(ly["sa"](qk, qk, hs, residual=hs), ly["ln1"][0], ly["ln1"][1], c.eps)
            hs = o.layernorm(ly["ca"](o.add(hs, qpos), text, text, residual=hs), ly["ln2"][0], ly["ln2"][1], c.eps)
            ctr = o.bmm(ref.view(1, Q, 4), self.e_ctr.view(1, 4, n_off), False)[0]
            whs = o.bmm(ref.view(1, Q, 4), self.e_wh.view(1, 4, n_off), False)[0]
            loc_fn = lambda off: o.add(o.elt(O.MUL, off, whs), ctr)
            hs = o.layernorm(ly["msda"](o.add(hs, qpos), vis, shapes, loc_fn, residual=hs), ly["ln3"][0], ly["ln3"][1], c.eps)
            ff = o.linear(o.linear(hs, ly["fc1"], act=ACT_RELU), ly["fc2"], residual=hs)
            hs = o.layernorm(ff, ly["ln4"][0], ly["ln4"][1], c.eps)
            last_ref = ref
            if return_aux:
                aux.setdefault("dec_hs", []).append(hs)
                aux.setdefault("dec_ref", []).append(ref)
            ref = o.elt(O.SIGMOID, o.add(self._mlp(hs, self.bbox[i]), o.elt(O.INVSIG, ref, alpha=1e-5)))
        hn = o.layernorm(hs, self.dec_ln[0], self.dec_ln[1], c.eps)
        logits_t = o.bmm(hn.view(1, Q, D), text.view(1, T, D), True)[0]                                   # [Q, T]
        pred_logits = torch.full((Q, c.max_text_len), float("-inf"), dtype=torch.float32, device=o.dev)
        pred_logits[:, :T] = logits_t                                                                     # device copy into the padded buffer
        pred_boxes = o.elt(O.SIGMOID, o.add(self._mlp(hn, self.bbox[-1]), o.elt(O.INVSIG, last_ref, alpha=1e-5)))
        if return_aux:
            return pred_logits, pred_boxes, aux
        return pred_logits, pred_boxes
