"""JPEG decode with the data-parallel part on the device ("next" row 2 of SURVEY.md 8f).

The reference reads images with ``cv2.imread`` (demo/demo.py:52) and detectron2's ``read_image`` = Pillow
(cubercnn/data/dataset_mapper.py:38) - libjpeg-turbo at its defaults either way. Here the host only undoes the entropy coding
(``ovm_host_jpeg_entropy_decode``: Huffman streams are serial) into pinned coefficient planes; dequantisation, the inverse DCT,
chroma upsampling and the colour transform run in libovm3d on the device (``ovm_jpeg_reconstruct``), bit-identical to Pillow's
``Image.open(f).convert("RGB")``, and the image is born in HBM where the resize kernel and the patch gather read it.

Baseline and progressive files are covered; streams outside the scope (arithmetic-coded, CMYK, 4:4:0, an incomplete progression ...) raise :class:`UnsupportedJpeg`;
:func:`read_image_device` then hands that file, like a PNG, to the host reader and uploads the pixels."""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np
import torch

from .. import lib as _lib

OVM_ERR_UNSUPPORTED = -6


class UnsupportedJpeg(ValueError):
    """A valid JPEG the device decoder does not cover (see include/ovm3d.h)."""


def jpeg_info(data: bytes) -> "_lib.OvmJpegInfo":
    """Header walk on the host: size, sampling layout, quantisation tables."""
    L = _lib.load()
    info = _lib.OvmJpegInfo()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    rc = L.ovm_host_jpeg_info(C.addressof(buf), len(data), C.byref(info))
    if rc == OVM_ERR_UNSUPPORTED:
        raise UnsupportedJpeg("JPEG outside the device decoder's scope (arithmetic / 12-bit / CMYK / unusual sampling)")
    _lib.check(rc, what="ovm_host_jpeg_info")
    return info


def entropy_decode(data: bytes, pin: bool = False) -> Tuple[torch.Tensor, "_lib.OvmJpegInfo"]:
    """Huffman-decodes every scan on the host: (coefficients int16 [coef_blocks, 64] in natural order, header record)."""
    L = _lib.load()
    info = jpeg_info(data)
    coef = torch.empty((int(info.coef_blocks), 64), dtype=torch.int16, pin_memory=pin)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    rc = L.ovm_host_jpeg_entropy_decode(C.addressof(buf), len(data), coef.data_ptr(), coef.numel(), C.byref(info))
    if rc == OVM_ERR_UNSUPPORTED:
        raise UnsupportedJpeg("JPEG outside the device decoder's scope")
    _lib.check(rc, what="ovm_host_jpeg_entropy_decode")
    return coef, info


def decode_jpeg(data: bytes, device: torch.device) -> torch.Tensor:
    """JPEG bytes -> uint8 RGB [H, W, 3] on the HIP device, bit-identical to Pillow's decode."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("decode_jpeg reconstructs on the HIP device only (no CPU fallback)")
    L = _lib.load()
    coef, info = entropy_decode(data, pin=True)
    with torch.cuda.device(device):
        d_coef = coef.to(device, non_blocking=True)
        planes = torch.empty(int(info.coef_blocks) * 64, dtype=torch.uint8, device=device)
        rgb = torch.empty((int(info.height), int(info.width), 3), dtype=torch.uint8, device=device)
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        rc = L.ovm_jpeg_reconstruct(d_coef.data_ptr(), C.byref(info), planes.data_ptr(), rgb.data_ptr(), stream)
        _lib.check(rc, what="ovm_jpeg_reconstruct")
        d_coef.record_stream(torch.cuda.current_stream(device))
    return rgb


def read_image_device(path: str, fmt: str, device: torch.device) -> torch.Tensor:
    """uint8 [H, W, 3] on the device in RGB or BGR channel order (``fmt``; BGR = cv2.imread's order, demo/demo.py:52). Baseline JPEGs are
    reconstructed on the device; other formats (PNG, CMYK JPEG ...) are read by the host reader and uploaded."""
    with open(path, "rb") as f:
        data = f.read()
    img = None
    if data[:2] == b"\xff\xd8":
        try:
            img = decode_jpeg(data, device)
        except UnsupportedJpeg:
            img = None
    if img is None:
        from .feeding import read_image
        img = torch.from_numpy(np.array(read_image(path, "RGB"))).to(device)
    return img.flip(-1) if fmt == "BGR" else img
