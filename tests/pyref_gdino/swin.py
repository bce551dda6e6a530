"""Swin Transformer backbone of GroundingDINO (``swin_B_384_22k``: embed 128, depths 2/2/18/2, heads 4/8/16/32,
window 12, patch 4; reference configs/GroundingDINO_SwinB_cfg.py:3,7). Returns the LayerNorm-ed outputs of stages
2-4 (strides 8/16/32). Padding to the window size, the cyclic shift, window partition / reverse and patch merging are
index maps built once per input size on the host; the device does gathers, MFMA projections, batched window attention
with relative-position bias + shift mask, LayerNorm and the MLPs. HF ``SwinBackbone`` parameter names."""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from .ops import ACT_GELU, Ops


def _window_maps(H: int, W: int, ws: int, shift: int):
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    nwh, nww = Hp // ws, Wp // ws
    ys, xs = np.meshgrid(np.arange(Hp), np.arange(Wp), indexing="ij")           # shifted-map coordinates
    sy, sx = (ys + shift) % Hp, (xs + shift) % Wp                               # source (padded) coordinates
    src = np.where((sy < H) & (sx < W), sy * W + sx, -1)
    win = src.reshape(nwh, ws, nww, ws).transpose(0, 2, 1, 3).reshape(-1, 1)     # [nW*ws*ws, 1]
    ty, tx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    qy, qx = (ty - shift) % Hp, (tx - shift) % Wp
    rev = ((qy // ws) * nww + qx // ws) * ws * ws + (qy % ws) * ws + qx % ws
    mask = None
    if shift > 0:
        hr = (np.arange(Hp) >= Hp - ws).astype(np.int64) + (np.arange(Hp) >= Hp - shift)
        wr = (np.arange(Wp) >= Wp - ws).astype(np.int64) + (np.arange(Wp) >= Wp - shift)
        img = hr[:, None] * 3 + wr[None, :]
        mw = img.reshape(nwh, ws, nww, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)
        d = mw[:, None, :] - mw[:, :, None]
        mask = np.where(d != 0, -100.0, 0.0).astype(np.float32)                  # [nW, ws*ws, ws*ws]
    return win.astype(np.int32), rev.reshape(-1, 1).astype(np.int32), mask, nwh * nww


def _merge_map(H: int, W: int):
    H2, W2 = -(-H // 2), -(-W // 2)
    idx = np.full((H2 * W2, 4), -1, dtype=np.int32)
    oy, ox = np.meshgrid(np.arange(H2), np.arange(W2), indexing="ij")
    k = 0
    for col in range(2):                                                          # HF order: for col in 2: for row in 2
        for row in range(2):
            y, x = 2 * oy + row, 2 * ox + col
            idx[:, k] = np.where((y < H) & (x < W), y * W + x, -1).reshape(-1)
            k += 1
    return idx, (H2, W2)


class SwinBackbone:
    def __init__(self, ops: Ops, sd: Dict[str, torch.Tensor], prefix: str, embed_dim: int, depths, heads, window: int = 12,
                 patch: int = 4, out_stages=(1, 2, 3), eps: float = 1e-5):
        self.o, self.ws, self.patch, self.out_stages, self.eps = ops, window, patch, tuple(out_stages), eps
        self.depths, self.heads = list(depths), list(heads)
        f = ops.f32
        p = prefix + "swin."
        w = sd[p + "embeddings.patch_embeddings.projection.weight"]                 # [C,3,4,4] -> (py,px,c)
        self.pe = ops.pack(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1), sd[p + "embeddings.patch_embeddings.projection.bias"])
        self.pe_g, self.pe_b = f(sd[p + "embeddings.norm.weight"]), f(sd[p + "embeddings.norm.bias"])
        ws2 = window * window
        ch = torch.arange(window)
        coords = torch.stack(torch.meshgrid(ch, ch, indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += window - 1
        rel[:, :, 1] += window - 1
        rel[:, :, 0] *= 2 * window - 1
        rel_index = rel.sum(-1).view(-1)
        self.stages = []
        for s, (depth, nh) in enumerate(zip(self.depths, self.heads)):
            blocks = []
            for b in range(depth):
                q = f"{p}encoder.layers.{s}.blocks.{b}."
                wq, wk, wv = (sd[q + f"attention.{n}_proj.weight"] for n in "qkv")
                bq, bk, bv = (sd[q + f"attention.{n}_proj.bias"] for n in "qkv")
                table = sd[q + "attention.relative_position_bias.relative_position_bias_table"]
                bias = table[rel_index].view(ws2, ws2, nh).permute(2, 0, 1).contiguous()          # [nh, ws2, ws2]
                blocks.append(dict(
                    g1=f(sd[q + "layernorm_before.weight"]), b1=f(sd[q + "layernorm_before.bias"]),
                    g2=f(sd[q + "layernorm_after.weight"]), b2=f(sd[q + "layernorm_after.bias"]),
                    qkv=ops.pack(torch.cat([wq, wk, wv], 0), torch.cat([bq, bk, bv], 0)),
                    proj=ops.pack(sd[q + "attention.o_proj.weight"], sd[q + "attention.o_proj.bias"]),
                    fc1=ops.pack(sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"]),
                    fc2=ops.pack(sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"]),
                    relbias=f(bias)))
            st = dict(blocks=blocks, nh=nh)
            dk = f"{p}encoder.layers.{s}.downsample."
            if dk + "reduction.weight" in sd:
                st.update(red=ops.pack(sd[dk + "reduction.weight"]), dg=f(sd[dk + "norm.weight"]), db=f(sd[dk + "norm.bias"]))
            nk = f"{prefix}hidden_states_norms.stage{s + 1}."
            if nk + "weight" in sd:
                st.update(og=f(sd[nk + "weight"]), ob=f(sd[nk + "bias"]))
            self.stages.append(st)
        self._maps = {}

    def _get_maps(self, H, W, shift):
        key = (H, W, shift)
        if key not in self._maps:
            win, rev, mask, nW = _window_maps(H, W, self.ws, shift)
            dev = self.o.dev
            self._maps[key] = (torch.from_numpy(win).to(dev), torch.from_numpy(rev).to(dev),
                               torch.from_numpy(mask).to(dev).contiguous() if mask is not None else None, nW)
        return self._maps[key]

    def forward(self, img_nhwc: torch.Tensor, H: int, W: int) -> List[Tuple[torch.Tensor, int, int]]:
        """img_nhwc: device fp32 [H*W, 3] (normalised). Returns [(features [h*w, C], h, w)] for the output stages."""
        o, ws, P = self.o, self.ws, self.patch
        Hp, Wp = -(-H // P), -(-W // P)
        key = ("pe", H, W)
        if key not in self._maps:
            py, px = np.meshgrid(np.arange(P), np.arange(P), indexing="ij")
            oy, ox = np.meshgrid(np.arange(Hp), np.arange(Wp), indexing="ij")
            y = oy.reshape(-1, 1) * P + py.reshape(1, -1)
            x = ox.reshape(-1, 1) * P + px.reshape(1, -1)
            self._maps[key] = torch.from_numpy(np.where((y < H) & (x < W), y * W + x, -1).astype(np.int32)).to(o.dev)
        x = o.linear(o.gather_rows(img_nhwc, self._maps[key]), self.pe)
        x = o.layernorm(x, self.pe_g, self.pe_b, self.eps)
        h, w = Hp, Wp
        outs = []
        ws2 = ws * ws
        for s, st in enumerate(self.stages):
            nh = st["nh"]
            C = x.shape[1]
            dh = C // nh
            for b, blk in enumerate(st["blocks"]):
                shift = 0 if b % 2 == 0 else ws // 2
                win, rev, mask, nW = self._get_maps(h, w, shift)
                xn = o.layernorm(x, blk["g1"], blk["b1"], self.eps)
                xw = o.gather_rows(xn, win)                                          # [nW*ws2, C] (pad + shift + partition)
                qkv = o.linear(xw, blk["qkv"])                                       # [nW*ws2, 3C]
                sc = o.empty(nW, nh, ws2, ws2)
                o.bmm2_raw(qkv, 0, qkv, C, sc, 0, nW, nh, ws2, ws2, dh, 3 * C, 3 * C, ws2,
                           ws2 * 3 * C, ws2 * 3 * C, nh * ws2 * ws2, dh, dh, ws2 * ws2, True, dh ** -0.5)
                o.softmax2_(sc, blk["relbias"], nh * ws2, 1, mask, nh * ws2, ws2)
                ctx = o.empty(nW * ws2, C)
                o.bmm2_raw(sc, 0, qkv, 2 * C, ctx, 0, nW, nh, ws2, dh, ws2, ws2, 3 * C, C,
                           nh * ws2 * ws2, ws2 * 3 * C, ws2 * C, ws2 * ws2, dh, dh, False, 1.0)
                ao = o.linear(ctx, blk["proj"])
                x = o.add(x, o.gather_rows(ao, rev))                                 # reverse windows, un-shift, crop; + shortcut
                hdn = o.linear(o.layernorm(x, blk["g2"], blk["b2"], self.eps), blk["fc1"], act=ACT_GELU)
                x = o.linear(hdn, blk["fc2"], residual=x)
            if s in self.out_stages:
                outs.append((o.layernorm(x, st["og"], st["ob"], self.eps), h, w))
            if "red" in st:
                mk = ("merge", h, w)
                if mk not in self._maps:
                    idx, dims = _merge_map(h, w)
                    self._maps[mk] = (torch.from_numpy(idx).to(o.dev), dims)
                idx, (h2, w2) = self._maps[mk]
                xm = o.gather_rows(x, idx)                                           # [h2*w2, 4C]
                x = o.linear(o.layernorm(xm, st["dg"], st["db"], self.eps), st["red"])
                h, w = h2, w2
        return outs
