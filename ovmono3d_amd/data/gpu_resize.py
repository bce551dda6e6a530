"""Device-side ``ResizeShortestEdge`` for uint8 images ("next" row 2 of SURVEY.md 8f).

detectron2's ``T.ResizeShortestEdge`` (reference demo/demo.py:79-83, cubercnn/data/dataset_mapper.py:62-72) resizes uint8
images with Pillow's ``Image.resize(size, BILINEAR)``. ``ovm_resize_bilinear_u8`` reproduces Pillow's fixed-point separable
resampling bit for bit (window tables built on the host by ``ovm_host_pil_bilinear_coeffs``), so the host's PIL resize can be
replaced by a device kernel without changing a single pixel. No CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import numpy as np
import torch

from .. import lib as _lib


def pil_bilinear_tables(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """(bounds int32 [out,2] = first input index / count, coefs int32 [out,ksize]) exactly as Pillow's precompute_coeffs +
    normalize_coeffs_8bpc produce them. Host only."""
    L = _lib.load()
    ksize = L.ovm_host_pil_bilinear_coeffs(int(in_size), int(out_size), None, None, 0)
    if ksize < 0:
        raise ValueError(f"bad sizes {in_size} -> {out_size}")
    bounds = np.zeros((out_size, 2), np.int32)
    coefs = np.zeros((out_size, ksize), np.int32)
    rc = L.ovm_host_pil_bilinear_coeffs(int(in_size), int(out_size), bounds.ctypes.data_as(C.c_void_p), coefs.ctypes.data_as(C.c_void_p),
                                        int(coefs.size))
    if rc != ksize:
        raise RuntimeError(f"ovm_host_pil_bilinear_coeffs failed ({rc})")
    return bounds, coefs


_tables: Dict[tuple, tuple] = {}


def _device_tables(in_size: int, out_size: int, dev: torch.device):
    key = (in_size, out_size, dev.index)
    t = _tables.get(key)
    if t is None:
        b, c = pil_bilinear_tables(in_size, out_size)
        if len(_tables) > 256:
            _tables.clear()
        t = _tables[key] = (torch.from_numpy(b).to(dev), torch.from_numpy(c).to(dev), int(c.shape[1]))
    return t


def resize_bilinear_u8(img: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """img: uint8 [H, W, C] on the HIP device (any strides) -> dense uint8 [out_h, out_w, C], bit-identical to
    ``PIL.Image.fromarray(img).resize((out_w, out_h), Image.BILINEAR)``."""
    if img.device.type != "cuda" or img.dtype != torch.uint8 or img.dim() != 3:
        raise RuntimeError("resize_bilinear_u8 takes a uint8 [H, W, C] tensor on the HIP device (no CPU fallback)")
    L = _lib.load()
    H, W, Cc = (int(v) for v in img.shape)
    need_h, need_v = out_w != W, out_h != H
    if not (need_h and True) and not need_v:
        return img.contiguous().clone()
    if not need_h and not img.is_contiguous():
        img = img.contiguous()
    dev = img.device
    xb = xc = yb = yc = None
    xk = yk = 0
    if need_h:
        xb, xc, xk = _device_tables(W, out_w, dev)
    if need_v:
        yb, yc, yk = _device_tables(H, out_h, dev)
    dst = torch.empty((out_h, out_w, Cc), dtype=torch.uint8, device=dev)
    tmp = torch.empty((H, out_w, Cc), dtype=torch.uint8, device=dev) if (need_h and need_v) else None
    sy, sx, sc = (int(s) for s in img.stride())
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda t: t.data_ptr() if t is not None else None
    _lib.check(L.ovm_resize_bilinear_u8(img.data_ptr(), H, W, Cc, sy, sx, sc, int(out_h), int(out_w), p(xb), p(xc), xk, p(yb), p(yc), yk, p(tmp),
                                        dst.data_ptr(), stream), what="ovm_resize_bilinear_u8")
    return dst


class ResizeShortestEdgeGPU:
    """``T.ResizeShortestEdge(min_size, max_size)`` for inference on device-resident uint8 HWC images."""

    def __init__(self, short_edge_length: int, max_size: int):
        from .feeding import ResizeShortestEdge
        self._shape = ResizeShortestEdge(short_edge_length, max_size).output_shape

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        nh, nw = self._shape(int(img.shape[0]), int(img.shape[1]))
        return resize_bilinear_u8(img, nh, nw)


def resize_bilinear_f32(x: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """x: float32 [H, W] or [B, H, W] on the HIP device -> [out_h, out_w] / [B, out_h, out_w]:
    ``F.interpolate(x, (out_h, out_w), mode="bilinear", align_corners=False)`` (the depth-prompt resizes of the reference's mapper,
    dataset_mapper.py:45-52,70-72). No CPU fallback."""
    if x.device.type != "cuda" or x.dtype != torch.float32 or x.dim() not in (2, 3):
        raise RuntimeError("resize_bilinear_f32 takes a float32 [H, W] or [B, H, W] tensor on the HIP device (no CPU fallback)")
    L = _lib.load()
    squeeze = x.dim() == 2
    x = (x[None] if squeeze else x).contiguous()
    B, H, W = (int(v) for v in x.shape)
    if (H, W) == (int(out_h), int(out_w)):
        return x[0].clone() if squeeze else x.clone()
    out = torch.empty((B, int(out_h), int(out_w)), dtype=torch.float32, device=x.device)
    stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(L.ovm_resize_bilinear_f32(x.data_ptr(), B, H, W, int(out_h), int(out_w), out.data_ptr(), stream), what="ovm_resize_bilinear_f32")
    return out[0] if squeeze else out


class DepthPromptResizeGPU:
    """The two resizes the reference's mapper applies to a depth prompt, on the device: to the image's size when the stored map
    has another one (dataset_mapper.py:45-52), then ResizeShortestEdge's output size (:70-72)."""

    def __init__(self, short_edge_length: int, max_size: int):
        from .feeding import ResizeShortestEdge
        self._shape = ResizeShortestEdge(short_edge_length, max_size).output_shape

    def __call__(self, depth: torch.Tensor, image_hw) -> torch.Tensor:
        h, w = int(image_hw[0]), int(image_hw[1])
        if tuple(depth.shape[-2:]) != (h, w):
            depth = resize_bilinear_f32(depth, h, w)
        nh, nw = self._shape(h, w)
        return resize_bilinear_f32(depth, nh, nw) if (nh, nw) != (h, w) else depth
