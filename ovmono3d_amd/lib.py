"""ctypes binding of libovm3d.so (the C ABI declared in include/ovm3d.h).

The library is built in-tree by ``ovmono3d_amd/csrc/build.sh`` (``__graft_entry__.build()``).
There is no CPU or PyTorch fallback: if the shared object is missing or fails to load,
importing the native path raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libovm3d.so")

OVM_REC_FLOATS = 48


class OvmConfig(C.Structure):
    _fields_ = [
        ("embed_dim", C.c_int32), ("depth", C.c_int32), ("heads", C.c_int32),
        ("pos_grid", C.c_int32), ("canvas", C.c_int32), ("fpn_channels", C.c_int32),
        ("use_depth_fusion", C.c_int32),
        ("pixel_mean", C.c_float * 3), ("pixel_std", C.c_float * 3),
        ("num_classes", C.c_int32), ("fc_dim", C.c_int32), ("pooler_res", C.c_int32),
        ("pooler_min_level", C.c_int32), ("pooler_max_level", C.c_int32),
        ("virtual_focal", C.c_float),
        ("anchor_sizes", C.c_float * 4), ("anchor_ratios", C.c_float * 3),
        ("rpn_pre_topk", C.c_int32), ("rpn_post_topk", C.c_int32), ("rpn_nms_thresh", C.c_float),
        ("score_thresh", C.c_float), ("nms_thresh", C.c_float), ("detections_per_image", C.c_int32),
        ("precision", C.c_int32), ("max_batch", C.c_int32), ("max_rois", C.c_int32),
        ("tower", C.c_int32), ("sam_window", C.c_int32), ("sam_global_mask", C.c_uint32),
    ]


OVM_TOWER_DINOV2, OVM_TOWER_CLIP, OVM_TOWER_MAE, OVM_TOWER_MIDAS, OVM_TOWER_SAM = 0, 1, 2, 3, 4


class OvmTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class OvmImage(C.Structure):
    _fields_ = [("data", C.c_void_p), ("height", C.c_int32), ("width", C.c_int32),
                ("stride_c", C.c_int64), ("stride_h", C.c_int64), ("stride_w", C.c_int64),
                ("orig_height", C.c_int32), ("orig_width", C.c_int32), ("K", C.c_float * 9)]


class OvmGdinoConfig(C.Structure):
    _fields_ = [
        ("d_model", C.c_int32), ("enc_layers", C.c_int32), ("dec_layers", C.c_int32), ("heads", C.c_int32), ("ffn_dim", C.c_int32),
        ("n_levels", C.c_int32), ("n_points", C.c_int32), ("num_queries", C.c_int32), ("max_text_len", C.c_int32),
        ("pe_temperature", C.c_float), ("eps", C.c_float), ("bert_heads", C.c_int32),
        ("swin_embed", C.c_int32), ("swin_depths", C.c_int32 * 4), ("swin_heads", C.c_int32 * 4), ("swin_window", C.c_int32),
        ("pixel_mean", C.c_float * 3), ("pixel_std", C.c_float * 3), ("flip_channels", C.c_int32), ("precision", C.c_int32),
        ("use_graphs", C.c_int32), ("max_plans", C.c_int32), ("plan_budget_mb", C.c_int32),
    ]


class OvmJpegInfo(C.Structure):
    """Mirror of include/ovm3d.h OvmJpegInfo."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("ncomp", C.c_int32), ("hmax", C.c_int32), ("vmax", C.c_int32),
        ("h", C.c_int32 * 3), ("v", C.c_int32 * 3), ("bw", C.c_int32 * 3), ("bh", C.c_int32 * 3), ("cw", C.c_int32 * 3),
        ("ch", C.c_int32 * 3), ("qidx", C.c_int32 * 3), ("colorspace", C.c_int32), ("coef_blocks", C.c_int32),
        ("qt", (C.c_uint16 * 64) * 4),
    ]


EXPORTS = [
    "ovm_create", "ovm_destroy", "ovm_last_error", "ovm_version", "ovm_abi_sizeof", "ovm_backbone_forward", "ovm_cube_forward",
    "ovm_rpn_box_forward", "ovm_gather_records", "ovm_gather_counts", "ovm_host_interp_pos_embed", "ovm_host_resize_pos_embed_aa", "ovm_host_sincos_pos_embed", "ovm_host_shard_range",
    "ovm_backbone_num_levels", "ovm_backbone_level",
    "ovm_op_split_f16", "ovm_op_interleave", "ovm_op_gemm", "ovm_op_layernorm", "ovm_op_attention", "ovm_op_roi_align",
    "ovm_op_cube_decode", "ovm_op_nms", "ovm_debug_copy", "ovm_set_corun", "ovm_profile_enable", "ovm_profile_read",
    "ovm_comm_unique_id", "ovm_comm_init", "ovm_comm_destroy", "ovm_tune_set", "ovm_gdino_postprocess", "ovm_box3d_iou", "ovm_host_pil_bilinear_coeffs", "ovm_resize_bilinear_u8", "ovm_resize_bilinear_f32",
    "ovm_g_pack_weight", "ovm_g_linear", "ovm_g_layernorm", "ovm_g_bmm", "ovm_g_bmm2", "ovm_g_softmax", "ovm_g_softmax2", "ovm_g_eltwise", "ovm_g_gather_rows",
    "ovm_g_groupnorm", "ovm_g_msdeform", "ovm_g_sine_embed", "ovm_g_normalize_image", "ovm_g_topk", "ovm_g_rowmax",
    "ovm_gdino_create", "ovm_gdino_destroy", "ovm_gdino_last_error", "ovm_gdino_forward", "ovm_gdino_detect", "ovm_gdino_set_force_topk",
    "ovm_gdino_debug_copy", "ovm_debug_set_ptr", "ovm_gdino_num_queries", "ovm_gdino_last_outputs", "ovm_infer",
    "ovm_host_jpeg_info", "ovm_host_jpeg_entropy_decode", "ovm_jpeg_reconstruct",
]
PROF_NAMES = ("attn", "qkv", "proj", "fc1", "fc2", "ln")

_lib = None


def load() -> C.CDLL:
    """Load libovm3d.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with ovmono3d_amd/csrc/build.sh (or __graft_entry__.build()). "
            "The native HIP path has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.ovm_create.argtypes = [C.POINTER(OvmConfig), C.POINTER(OvmTensor), i32, i32, C.POINTER(vp)]
    lib.ovm_destroy.argtypes = [vp]
    lib.ovm_last_error.argtypes = [vp]
    lib.ovm_last_error.restype = C.c_char_p
    lib.ovm_version.restype = C.c_char_p
    lib.ovm_abi_sizeof.argtypes = [C.c_char_p]
    for name, mirror in (("OvmConfig", OvmConfig), ("OvmTensor", OvmTensor), ("OvmImage", OvmImage), ("OvmGdinoConfig", OvmGdinoConfig),
                         ("OvmJpegInfo", OvmJpegInfo)):
        if lib.ovm_abi_sizeof(name.encode()) != C.sizeof(mirror):
            raise RuntimeError(f"{LIB_PATH}: sizeof({name}) = {lib.ovm_abi_sizeof(name.encode())} but the ctypes mirror has "
                               f"{C.sizeof(mirror)} bytes - rebuild the library (ovmono3d_amd/csrc/build.sh) or update lib.py")
    lib.ovm_host_jpeg_info.argtypes = [vp, C.c_size_t, C.POINTER(OvmJpegInfo)]
    lib.ovm_host_jpeg_entropy_decode.argtypes = [vp, C.c_size_t, vp, i64, C.POINTER(OvmJpegInfo)]
    lib.ovm_jpeg_reconstruct.argtypes = [vp, C.POINTER(OvmJpegInfo), vp, vp, vp]
    lib.ovm_backbone_forward.argtypes = [vp, C.POINTER(OvmImage), i32, vp, i32, i32, vp, vp, vp, vp]
    lib.ovm_cube_forward.argtypes = [vp, C.POINTER(OvmImage), i32, vp, vp, vp, vp, i32, i32, vp, vp, vp]
    lib.ovm_rpn_box_forward.argtypes = [vp, C.POINTER(OvmImage), i32, vp, vp, vp, vp, vp, vp, vp]
    lib.ovm_gather_records.argtypes = [vp, i32, i32, vp, i32, vp, C.POINTER(i32), vp]
    lib.ovm_gather_counts.argtypes = [vp, i32, i32, i32, C.POINTER(i32), vp]
    lib.ovm_host_interp_pos_embed.argtypes = [vp, i32, i32, i32, vp]
    lib.ovm_host_resize_pos_embed_aa.argtypes = [vp, i32, i32, i32, vp]
    lib.ovm_host_sincos_pos_embed.argtypes = [i32, i32, vp]
    lib.ovm_backbone_num_levels.argtypes = [vp]
    lib.ovm_backbone_level.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(f32)]
    lib.ovm_host_shard_range.argtypes = [i64, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    lib.ovm_op_split_f16.argtypes = [vp, i64, vp, vp, vp]
    lib.ovm_op_interleave.argtypes = [vp, vp, i64, i32, vp, vp]
    lib.ovm_op_gemm.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, vp, i32, vp, i32, i32, vp]
    lib.ovm_op_layernorm.argtypes = [vp, i32, i32, vp, vp, f32, vp, vp]
    lib.ovm_op_attention.argtypes = [vp, i32, i32, i32, vp, i32, vp]
    lib.ovm_op_roi_align.argtypes = [vp, vp, vp, C.POINTER(i32), C.POINTER(f32), i32, i32, i32, i32, vp, vp, i32, vp, vp]
    lib.ovm_op_cube_decode.argtypes = [vp, i32, vp, vp, vp, vp, C.POINTER(OvmImage), i32, i32, f32, i32, vp, vp, vp]
    lib.ovm_op_nms.argtypes = [vp, vp, i32, f32, vp, vp, vp]
    lib.ovm_debug_copy.argtypes = [vp, C.c_char_p, vp, i64, vp]
    lib.ovm_debug_copy.restype = i64
    lib.ovm_profile_enable.argtypes = [vp, i32]
    lib.ovm_set_corun.argtypes = [vp, i32]
    lib.ovm_profile_read.argtypes = [vp, C.POINTER(f32), C.POINTER(i32)]
    lib.ovm_comm_unique_id.argtypes = [vp]
    lib.ovm_comm_init.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    lib.ovm_comm_destroy.argtypes = [vp]
    lib.ovm_tune_set.argtypes = [C.c_char_p, i32]
    lib.ovm_debug_set_ptr.argtypes = [C.c_char_p, vp]
    lib.ovm_host_pil_bilinear_coeffs.argtypes = [i32, i32, vp, vp, i32]
    lib.ovm_resize_bilinear_u8.argtypes = [vp, i32, i32, i32, i64, i64, i64, i32, i32, vp, vp, i32, vp, vp, i32, vp, vp, vp]
    lib.ovm_resize_bilinear_f32.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
    lib.ovm_box3d_iou.argtypes = [vp, vp, i32, i32, f32, f32, vp, vp, vp]
    lib.ovm_g_pack_weight.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    lib.ovm_g_linear.argtypes = [vp, i32, i32, i32, vp, vp, i32, i32, vp, i32, vp, i32, vp, i32, i32, vp]
    lib.ovm_g_layernorm.argtypes = [vp, vp, i32, i32, vp, vp, f32, vp, vp]
    lib.ovm_g_bmm.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i64, i64, i64, i32, f32, vp]
    lib.ovm_g_bmm2.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i64, i64, i64, i64, i64, i64, i32, f32, vp]
    lib.ovm_g_softmax2.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, vp, i32, i32, vp]
    lib.ovm_g_softmax.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, vp]
    lib.ovm_g_eltwise.argtypes = [i32, vp, vp, vp, i64, i64, f32, f32, vp]
    lib.ovm_g_gather_rows.argtypes = [vp, i32, vp, i64, i32, i32, vp, vp]
    lib.ovm_g_groupnorm.argtypes = [vp, i32, i32, i32, i32, vp, vp, f32, vp, vp]
    lib.ovm_g_msdeform.argtypes = [vp, C.POINTER(i32), i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.ovm_g_sine_embed.argtypes = [vp, i64, i32, i32, f32, vp, vp]
    lib.ovm_g_normalize_image.argtypes = [C.POINTER(OvmImage), C.POINTER(f32), C.POINTER(f32), i32, vp, vp]
    lib.ovm_g_topk.argtypes = [vp, i32, i32, vp, vp]
    lib.ovm_g_rowmax.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.ovm_gdino_postprocess.argtypes = [vp, i32, i32, vp, C.POINTER(i32), i32, i32, i32, f32, f32, vp, vp, vp, vp, vp]
    lib.ovm_gdino_create.argtypes = [C.POINTER(OvmGdinoConfig), C.POINTER(OvmTensor), i32, i32, C.POINTER(vp)]
    lib.ovm_gdino_destroy.argtypes = [vp]
    lib.ovm_gdino_last_error.argtypes = [vp]
    lib.ovm_gdino_last_error.restype = C.c_char_p
    lib.ovm_gdino_forward.argtypes = [vp, C.POINTER(OvmImage), C.POINTER(i32), i32, C.POINTER(i32), vp, vp, vp]
    lib.ovm_gdino_detect.argtypes = [vp, C.POINTER(OvmImage), C.POINTER(i32), i32, C.POINTER(i32), i32, f32, f32, vp, vp, vp, vp, vp]
    lib.ovm_gdino_set_force_topk.argtypes = [vp, vp]
    lib.ovm_gdino_debug_copy.argtypes = [vp, C.c_char_p, vp, i64, vp]
    lib.ovm_gdino_debug_copy.restype = i64
    lib.ovm_gdino_num_queries.argtypes = [vp]
    lib.ovm_gdino_last_outputs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32)]
    lib.ovm_infer.argtypes = [vp, vp, C.POINTER(OvmImage), C.POINTER(i32), i32, C.POINTER(i32), i32, f32, f32, vp, i32, C.POINTER(i32), vp]
    for name in EXPORTS:
        if name not in ("ovm_last_error", "ovm_version", "ovm_debug_copy", "ovm_gdino_last_error", "ovm_gdino_debug_copy"):
            getattr(lib, name).restype = i32
    # experiment knobs, e.g. OVM_TUNE="gemm_bm=256,attn_tail=0"
    for kv in filter(None, os.environ.get("OVM_TUNE", "").split(",")):
        k, _, v = kv.partition("=")
        lib.ovm_tune_set(k.strip().encode(), int(v))
    _lib = lib
    return lib


class OvmError(RuntimeError):
    pass


def check(rc: int, handle=None, what: str = "") -> None:
    if rc == 0:
        return
    msg = ""
    if handle:
        msg = (load().ovm_last_error(handle) or b"").decode()
    raise OvmError(f"{what} failed with code {rc}: {msg}")


def make_tensor_table(state_dict: Dict[str, "np.ndarray"]):
    """state_dict values: contiguous float32 numpy arrays (host). Returns (ctypes array, keepalive)."""
    items = list(state_dict.items())
    arr = (OvmTensor * len(items))()
    keep = []
    for i, (k, v) in enumerate(items):
        v = np.ascontiguousarray(v, dtype=np.float32)
        if v.ndim > 4:
            raise ValueError(f"{k}: more than 4 dims")
        kb = k.encode()
        keep.append((kb, v))
        arr[i].name = kb
        arr[i].data = v.ctypes.data
        arr[i].ndim = v.ndim
        for d in range(v.ndim):
            arr[i].shape[d] = v.shape[d]
    return arr, keep


def shard_range(n: int, rank: int, world: int):
    b, e = C.c_int64(), C.c_int64()
    check(load().ovm_host_shard_range(n, rank, world, C.byref(b), C.byref(e)), what="ovm_host_shard_range")
    return b.value, e.value


def interp_pos_embed(pos: "np.ndarray", G: int) -> "np.ndarray":
    """pos [1+M*M, D] float32 -> [1+G*G, D] (host, no GPU needed)."""
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    M = int(round((pos.shape[0] - 1) ** 0.5))
    out = np.empty((1 + G * G, pos.shape[1]), dtype=np.float32)
    check(load().ovm_host_interp_pos_embed(pos.ctypes.data, M, pos.shape[1], G, out.ctypes.data),
          what="ovm_host_interp_pos_embed")
    return out
