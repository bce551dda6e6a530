"""ctypes front end of libovm3d's GroundingDINO engine (``ovm_gdino_*``): the whole network - BERT, Swin, fusion encoder,
two-stage selection, decoder - is sequenced inside the library and replayed as one HIP graph per (image size, caption); Python
only hands over the checkpoint tensors once and, per call, the image descriptor and the caption's token ids.

Replaces ``load_model(...)`` + ``model(image[None], captions=[caption])`` of reference
cubercnn/modeling/roi_heads/roi_heads_gdino.py:16-23,186."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from .. import lib as _lib
from .config import GDinoConfig


class GdinoEngine:
    def __init__(self, device: torch.device, state_dict: Dict[str, torch.Tensor], cfg: GDinoConfig = GDinoConfig(),
                 pixel_mean: Sequence[float] = (0.0, 0.0, 0.0), pixel_std: Sequence[float] = (1.0, 1.0, 1.0), flip_channels: bool = True,
                 precision: int = 3, use_graphs: bool = True, max_plans: int = 0, plan_budget_mb: int = 0):
        if device.type != "cuda":
            raise RuntimeError("the GroundingDINO engine runs on the HIP device only (no CPU fallback)")
        self.dev, self.cfg, self.L = device, cfg, _lib.load()
        c = _lib.OvmGdinoConfig()
        c.d_model, c.enc_layers, c.dec_layers, c.heads, c.ffn_dim = cfg.d_model, cfg.enc_layers, cfg.dec_layers, cfg.heads, cfg.ffn_dim
        c.n_levels, c.n_points, c.num_queries, c.max_text_len = cfg.n_levels, cfg.n_points, cfg.num_queries, cfg.max_text_len
        c.pe_temperature, c.eps, c.bert_heads = float(cfg.pe_temperature), float(cfg.eps), cfg.bert_heads
        c.swin_embed, c.swin_window = cfg.swin_embed, cfg.swin_window
        for i in range(4):
            c.swin_depths[i] = int(cfg.swin_depths[i]) if i < len(cfg.swin_depths) else 0
            c.swin_heads[i] = int(cfg.swin_heads[i]) if i < len(cfg.swin_heads) else 0
        for i in range(3):
            c.pixel_mean[i], c.pixel_std[i] = float(pixel_mean[i]), float(pixel_std[i])
        c.flip_channels, c.precision, c.use_graphs, c.max_plans = int(flip_channels), int(precision), int(use_graphs), int(max_plans)
        c.plan_budget_mb = int(plan_budget_mb)          # 0 = library defaults: up to 128 plans within 32 GiB (LRU)
        # bool / integer buffers (relative_position_index ...) are not weights; tensors above 4-d do not occur
        host = {k: np.ascontiguousarray(v.detach().to(torch.float32).cpu().numpy()) for k, v in state_dict.items()
                if torch.is_tensor(v) and v.dtype.is_floating_point and v.dim() <= 4}
        table, keep = _lib.make_tensor_table(host)
        self._h = C.c_void_p()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        rc = self.L.ovm_gdino_create(C.byref(c), table, len(host), idx, C.byref(self._h))
        if rc != 0:
            msg = (self.L.ovm_gdino_last_error(self._h) or b"").decode() if self._h else ""
            if self._h:
                self.L.ovm_gdino_destroy(self._h)
                self._h = None
            raise _lib.OvmError(f"ovm_gdino_create failed with code {rc}: {msg}")
        del keep
        self._force = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self.L.ovm_gdino_destroy(h)
            except Exception:
                pass

    def _chk(self, rc, what):
        if rc != 0:
            raise _lib.OvmError(f"{what} failed with code {rc}: {(self.L.ovm_gdino_last_error(self._h) or b'').decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    @staticmethod
    def describe(image_u8_chw: torch.Tensor) -> "_lib.OvmImage":
        d = _lib.OvmImage()
        d.data = image_u8_chw.data_ptr()
        d.height, d.width = int(image_u8_chw.shape[1]), int(image_u8_chw.shape[2])
        d.stride_c, d.stride_h, d.stride_w = (int(s) for s in image_u8_chw.stride())
        d.orig_height, d.orig_width = d.height, d.width
        return d

    def forward(self, image_u8_chw: torch.Tensor, input_ids: Sequence[int], position_ids: Optional[Sequence[int]] = None):
        """image: uint8 [3,H,W] on the device (any strides). Returns (pred_logits [Q, max_text_len], pred_boxes [Q, 4])."""
        assert image_u8_chw.dtype == torch.uint8 and image_u8_chw.is_cuda and image_u8_chw.dim() == 3
        n = len(input_ids)
        ids = (C.c_int32 * n)(*[int(i) for i in input_ids])
        pids = (C.c_int32 * n)(*[int(i) for i in position_ids]) if position_ids is not None else None
        Q = self.cfg.num_queries
        logits = torch.empty((Q, self.cfg.max_text_len), dtype=torch.float32, device=self.dev)
        boxes = torch.empty((Q, 4), dtype=torch.float32, device=self.dev)
        d = self.describe(image_u8_chw)
        self._chk(self.L.ovm_gdino_forward(self._h, C.byref(d), ids, n, pids, logits.data_ptr(), boxes.data_ptr(), self._stream()),
                  "ovm_gdino_forward")
        return logits, boxes

    # ---- test hooks ----
    def set_force_topk(self, idx: Optional[torch.Tensor]):
        self._force = idx.to(self.dev, torch.int32).contiguous() if idx is not None else None
        self._chk(self.L.ovm_gdino_set_force_topk(self._h, self._force.data_ptr() if self._force is not None else None), "ovm_gdino_set_force_topk")

    def debug(self, name: str, shape, dtype=torch.float32) -> torch.Tensor:
        out = torch.empty(shape, dtype=dtype, device=self.dev)
        n = self.L.ovm_gdino_debug_copy(self._h, name.encode(), out.data_ptr(), out.numel(), self._stream())
        if n < 0:
            self._chk(int(n), f"ovm_gdino_debug_copy({name})")
        assert n == out.numel(), (name, n, tuple(shape))
        return out

    def launches(self) -> int:
        return int(self.L.ovm_gdino_debug_copy(self._h, b"launches", None, 0, self._stream()))

    def debug_scalar(self, name: str) -> int:
        """"plans": plans held by the cache; "plan_bytes": device memory they hold together; "launches": kernels of the last forward."""
        return int(self.L.ovm_gdino_debug_copy(self._h, name.encode(), None, 0, self._stream()))
