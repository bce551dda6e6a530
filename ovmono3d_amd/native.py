"""Host-side engine: owns one libovm3d handle and moves torch device tensors through the C ABI.

PyTorch is used here only for device memory, the current HIP stream and H2D/D2H copies;
every arithmetic step of the path runs inside libovm3d's HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as _lib
from .lib import OVM_REC_FLOATS, OVM_TOWER_CLIP, OVM_TOWER_DINOV2, OVM_TOWER_MAE, OVM_TOWER_MIDAS, OVM_TOWER_SAM, OvmConfig, OvmImage, check
from .util.synth_weights import CLIP_ARCH, MAE_ARCH, MIDAS_ARCH, SAM_ARCH, VIT_ARCH


def config_to_native(cfg) -> OvmConfig:
    """Reference config tree -> OvmConfig (keys cited in include/ovm3d.h)."""
    backbone = cfg.MODEL.BACKBONE.NAME
    if backbone == "build_dino_backbone":
        tower, patch = OVM_TOWER_DINOV2, 14
        name = cfg.MODEL.DINO.MODEL_NAME
        if name not in VIT_ARCH:
            raise ValueError(f"unsupported MODEL.DINO.MODEL_NAME {name!r} (known: {sorted(VIT_ARCH)})")
        if cfg.MODEL.DINO.NAME != "dinov2":
            raise ValueError("only MODEL.DINO.NAME == 'dinov2' is on this path")
        if cfg.MODEL.DINO.OUTPUT != "dense" or cfg.MODEL.DINO.RETURN_MULTILAYER or cfg.MODEL.DINO.LAYER != -1:
            raise ValueError("native path supports MODEL.DINO.OUTPUT 'dense', LAYER -1, single layer")
        D, L, heads = VIT_ARCH[name]
        pos_grid, n_levels = 37, 3
    elif backbone == "build_clip_backbone":
        tower, patch = OVM_TOWER_CLIP, 16
        name = cfg.MODEL.CLIP.ARCH
        if name not in CLIP_ARCH:
            raise ValueError(f"unsupported MODEL.CLIP.ARCH {name!r} (known: {sorted(CLIP_ARCH)})")
        if cfg.MODEL.CLIP.OUTPUT != "dense" or cfg.MODEL.CLIP.RETURN_MULTILAYER or cfg.MODEL.CLIP.LAYER != -1:
            raise ValueError("native path supports MODEL.CLIP.OUTPUT 'dense', LAYER -1, single layer")
        D, L, heads, patch, pos_grid = CLIP_ARCH[name]
        n_levels = 4
    elif backbone == "build_mae_backbone":
        tower = OVM_TOWER_MAE
        name = cfg.MODEL.MAE.CHECKPOINT
        if name not in MAE_ARCH:
            raise ValueError(f"unsupported MODEL.MAE.CHECKPOINT {name!r} (known: {sorted(MAE_ARCH)})")
        if cfg.MODEL.MAE.OUTPUT != "dense" or cfg.MODEL.MAE.RETURN_MULTILAYER or cfg.MODEL.MAE.LAYER != -1:
            raise ValueError("native path supports MODEL.MAE.OUTPUT 'dense', LAYER -1, single layer")
        D, L, heads, patch = MAE_ARCH[name]
        # the reference taps hidden_states[num_layers - 1] (mae.py:43-55,110-116): the state before the LAST block, so one block fewer runs
        L = L - 1
        pos_grid, n_levels = 0, 4
    elif backbone == "build_midas_backbone":
        tower = OVM_TOWER_MIDAS
        name = cfg.MODEL.MIDAS.ARCH
        if name not in MIDAS_ARCH:
            raise ValueError(f"unsupported MODEL.MIDAS.ARCH {name!r} (known: {sorted(MIDAS_ARCH)})")
        if cfg.MODEL.MIDAS.OUTPUT != "dense" or cfg.MODEL.MIDAS.RETURN_MULTILAYER or cfg.MODEL.MIDAS.LAYER != -1:
            raise ValueError("native path supports MODEL.MIDAS.OUTPUT 'dense', LAYER -1, single layer")
        D, L, heads, patch, pos_grid = MIDAS_ARCH[name]
        n_levels = 4
    elif backbone == "build_sam_backbone":
        tower = OVM_TOWER_SAM
        name = cfg.MODEL.SAM.ARCH
        if name not in SAM_ARCH:
            raise ValueError(f"unsupported MODEL.SAM.ARCH {name!r} (known: {sorted(SAM_ARCH)})")
        if cfg.MODEL.SAM.OUTPUT != "dense" or cfg.MODEL.SAM.RETURN_MULTILAYER or cfg.MODEL.SAM.LAYER != -1:
            raise ValueError("native path supports MODEL.SAM.OUTPUT 'dense', LAYER -1, single layer")
        D, L, heads, patch, pos_grid, sam_window, sam_global = SAM_ARCH[name]
        n_levels = 4
    else:
        raise ValueError(f"MODEL.BACKBONE.NAME {backbone!r} is not on the native path (build_dino_backbone, build_clip_backbone, "
                         "build_mae_backbone, build_midas_backbone, build_sam_backbone)")
    H = cfg.MODEL.ROI_CUBE_HEAD
    unsupported = []
    if H.Z_TYPE != "direct": unsupported.append("Z_TYPE")
    if H.POSE_TYPE != "6d": unsupported.append("POSE_TYPE")
    if H.DIMS_PRIORS_ENABLED: unsupported.append("DIMS_PRIORS_ENABLED")
    if H.CLUSTER_BINS != 1: unsupported.append("CLUSTER_BINS")
    if not H.SHARED_FC: unsupported.append("SHARED_FC")
    if not H.ALLOCENTRIC_POSE: unsupported.append("ALLOCENTRIC_POSE")
    if not H.VIRTUAL_DEPTH: unsupported.append("VIRTUAL_DEPTH")
    if not H.USE_CONFIDENCE: unsupported.append("USE_CONFIDENCE")
    if H.SCALE_ROI_BOXES: unsupported.append("SCALE_ROI_BOXES")
    if H.NUM_CONV: unsupported.append("NUM_CONV")
    if unsupported:
        raise ValueError("ROI_CUBE_HEAD settings outside the OVMono3D-LIFT path (Base.yaml:71-86): " + ", ".join(unsupported))
    S = int(cfg.MODEL.FPN.SQUARE_PAD)
    if S <= 0 or S % patch != 0:
        raise ValueError(f"MODEL.FPN.SQUARE_PAD must be a positive multiple of {patch}")
    c = OvmConfig()
    c.tower = tower
    if tower == OVM_TOWER_SAM:
        c.sam_window = sam_window
        c.sam_global_mask = sum(1 << i for i in sam_global)
    c.embed_dim, c.depth, c.heads = D, L, heads
    c.pos_grid = pos_grid
    c.canvas = S
    c.fpn_channels = int(cfg.MODEL.FPN.OUT_CHANNELS)
    c.use_depth_fusion = int(bool(cfg.MODEL.DINO.USE_DEPTH_FUSION)) if tower == OVM_TOWER_DINOV2 else 0
    for i in range(3):
        c.pixel_mean[i] = float(cfg.MODEL.PIXEL_MEAN[i])
        c.pixel_std[i] = float(cfg.MODEL.PIXEL_STD[i])
    c.num_classes = int(cfg.MODEL.ROI_HEADS.NUM_CLASSES)
    c.fc_dim = int(H.FC_DIM)
    c.pooler_res = int(H.POOLER_RESOLUTION)
    if tower == OVM_TOWER_DINOV2:
        # strides 7 / 14 / 28 are not powers of two: detectron2's ROIPooler level rule needs the fork's clamp
        c.pooler_min_level = int(cfg.MODEL.ROI_HEADS.POOLER_MIN_LEVEL)
        c.pooler_max_level = int(cfg.MODEL.ROI_HEADS.POOLER_MAX_LEVEL)
    else:
        # strides patch/4 .. 2*patch are powers of two: ROIPooler's own min_level = -log2(scale_0), max_level = -log2(scale_last)
        c.pooler_min_level = int(np.log2(patch / 4))
        c.pooler_max_level = int(np.log2(patch * 2))
    c.virtual_focal = float(H.VIRTUAL_FOCAL)
    sizes = [s[0] for s in cfg.MODEL.ANCHOR_GENERATOR.SIZES]
    ratios = list(cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS[0])
    if len(sizes) != n_levels or len(ratios) != 3:
        raise ValueError(f"native RPN expects one anchor size per pyramid level ({n_levels}) x 3 aspect ratios "
                         "(OVMono3D_dinov2_SFP.yaml:35-36, OVMono3D_clip_SFP.yaml:40-41)")
    for i in range(n_levels):
        c.anchor_sizes[i] = float(sizes[i])
    for i in range(3):
        c.anchor_ratios[i] = float(ratios[i])
    c.rpn_pre_topk = int(cfg.MODEL.RPN.PRE_NMS_TOPK_TEST)
    c.rpn_post_topk = int(cfg.MODEL.RPN.POST_NMS_TOPK_TEST)
    c.rpn_nms_thresh = float(cfg.MODEL.RPN.NMS_THRESH)
    c.score_thresh = float(cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST)
    c.nms_thresh = float(cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST)
    c.detections_per_image = int(cfg.TEST.DETECTIONS_PER_IMAGE)
    prec = cfg.MODEL.AMD.GEMM_PRECISION
    if prec not in ("f16", "f16x3"):
        raise ValueError("MODEL.AMD.GEMM_PRECISION must be 'f16' or 'f16x3'")
    c.precision = 3 if prec == "f16x3" else 1
    c.max_batch = int(cfg.MODEL.AMD.MAX_BATCH)
    c.max_rois = int(cfg.MODEL.AMD.MAX_ROIS)
    return c


class Engine:
    """One native handle on one device. Not thread-safe (mirrors the reference's 1 thread/process)."""

    def __init__(self, cfg, device: Optional[torch.device] = None):
        self.cfg = cfg
        self.ncfg = config_to_native(cfg)
        self.device = torch.device(device if device is not None else cfg.MODEL.DEVICE)
        if self.device.type != "cuda":
            raise RuntimeError("the MI355X-native path runs on a HIP device only (MODEL.DEVICE cuda); "
                               "there is no CPU fallback")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.patch = 14 if self.ncfg.tower == OVM_TOWER_DINOV2 else 16
        self.G = self.ncfg.canvas // self.patch
        self.C = self.ncfg.fpn_channels
        # pyramid levels, finest first: (name, side of the grid on the canvas, stride in pixels)
        scales = (2.0, 1.0, 0.5) if self.ncfg.tower == OVM_TOWER_DINOV2 else (4.0, 2.0, 1.0, 0.5)
        self.levels = [(f"p{2 + i}", int(self.G * sc), self.patch / sc) for i, sc in enumerate(scales)]      # G odd: MaxPool2 floors

    # ---- lifecycle ---------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Dict[str, torch.Tensor]) -> None:
        self.close()
        sd = {k: v.detach().to("cpu", torch.float32).contiguous().numpy() for k, v in state_dict.items()
              if isinstance(v, torch.Tensor) and v.dtype.is_floating_point and v.dim() <= 4}
        table, keep = _lib.make_tensor_table(sd)
        h = C.c_void_p()
        torch.cuda.set_device(self.device)
        rc = self._lib.ovm_create(C.byref(self.ncfg), table, len(keep), self.device.index, C.byref(h))
        if rc != 0:
            msg = (self._lib.ovm_last_error(h) or b"").decode() if h else ""
            if h:
                self._lib.ovm_destroy(h)
            raise _lib.OvmError(f"ovm_create failed ({rc}): {msg}")
        self._h = h

    def close(self) -> None:
        if self._h:
            self._lib.ovm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ready(self) -> bool:
        return bool(self._h)

    def _require(self):
        if not self._h:
            raise RuntimeError("no weights loaded: call model.load_state_dict(...) / DetectionCheckpointer(...).resume_or_load first")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- helpers -----------------------------------------------------------------------------
    def make_images(self, batched_inputs: Sequence[Dict]) -> Tuple[C.Array, List[torch.Tensor]]:
        arr = (OvmImage * len(batched_inputs))()
        keep = []
        for i, b in enumerate(batched_inputs):
            im = b["image"]
            if im.dtype != torch.uint8 or im.dim() != 3:
                raise TypeError("'image' must be a uint8 tensor [3,H,W] (reference demo.py:83) or [H,W,3]")
            if im.shape[0] == 3:
                chw = im
            elif im.shape[2] == 3:
                chw = im.permute(2, 0, 1)          # NHWC storage: a stride change, no copy
            else:
                raise TypeError("'image' must have 3 channels")
            chw = chw.to(self.device, non_blocking=True)
            keep.append(chw)
            arr[i].data = chw.data_ptr()
            arr[i].height, arr[i].width = int(chw.shape[1]), int(chw.shape[2])
            arr[i].stride_c, arr[i].stride_h, arr[i].stride_w = (int(s) for s in chw.stride())
            arr[i].orig_height = int(b.get("height", chw.shape[1]))
            arr[i].orig_width = int(b.get("width", chw.shape[2]))
            K = np.asarray(b["K"], dtype=np.float32).reshape(-1) if "K" in b else np.eye(3, dtype=np.float32).reshape(-1)
            for j in range(9):
                arr[i].K[j] = float(K[j])
        return arr, keep

    # ---- stages ------------------------------------------------------------------------------
    def backbone_forward(self, images, B: int, prompt_depth: Optional[torch.Tensor] = None, export: bool = False):
        self._require()
        dp, dh, dw = None, 0, 0
        if prompt_depth is not None:
            prompt_depth = prompt_depth.to(self.device, torch.float32).contiguous()
            if prompt_depth.dim() == 3:
                prompt_depth = prompt_depth.unsqueeze(1)
            dp, dh, dw = prompt_depth.data_ptr(), int(prompt_depth.shape[-2]), int(prompt_depth.shape[-1])
        outs = [None, None, None]
        ptrs = [None, None, None]
        if export:
            for i, (_, g, _) in enumerate(self.levels[:3]):
                outs[i] = torch.empty((B, g, g, self.C), dtype=torch.float32, device=self.device)
                ptrs[i] = outs[i].data_ptr()
        rc = self._lib.ovm_backbone_forward(self._h, images, B, dp, dh, dw, ptrs[0], ptrs[1], ptrs[2], self._stream())
        check(rc, self._h, "ovm_backbone_forward")
        if export:
            for name, g, _ in self.levels[3:]:                                  # p5 of the 4-level towers: read from the handle
                outs.append(self.debug_tensor(name, B * g * g * self.C).view(B, g, g, self.C))
            # logical NCHW view over NHWC storage (the reference returns NCHW tensors, dino.py:170)
            return {lv[0]: o.permute(0, 3, 1, 2) for lv, o in zip(self.levels, outs)}
        return None

    def debug_tensor(self, name: str, numel: int) -> torch.Tensor:
        self._require()
        out = torch.empty(numel, dtype=torch.float32, device=self.device)
        n = self._lib.ovm_debug_copy(self._h, name.encode(), out.data_ptr(), numel, self._stream())
        if n < 0:
            check(int(n), self._h, "ovm_debug_copy")
        return out[:n]

    def cube_forward(self, images, B: int, boxes: torch.Tensor, scores: torch.Tensor, classes: torch.Tensor,
                     image_idx: torch.Tensor, postprocess: bool = True):
        """Returns (records [n_keep, 48] float32 device tensor, counts list[int])."""
        self._require()
        n = int(boxes.shape[0])
        dev = self.device
        boxes = boxes.to(dev, torch.float32).contiguous()
        scores = scores.to(dev, torch.float32).contiguous()
        classes = classes.to(dev, torch.int32).contiguous()
        image_idx = image_idx.to(dev, torch.int32).contiguous()
        rec = torch.empty((max(n, 1), OVM_REC_FLOATS), dtype=torch.float32, device=dev)
        counts = torch.zeros(B, dtype=torch.int32, device=dev)
        rc = self._lib.ovm_cube_forward(self._h, images, B, boxes.data_ptr(), scores.data_ptr(), classes.data_ptr(),
                                        image_idx.data_ptr(), n, int(bool(postprocess)), rec.data_ptr(),
                                        counts.data_ptr(), self._stream())
        check(rc, self._h, "ovm_cube_forward")
        cl = counts.cpu().tolist()            # one D2H sync per batch (reference: omni3d_evaluation.py:669)
        return rec[: sum(cl)], cl

    def infer_gdino(self, images, gdino_handle, token_ids: Sequence[int], spans: Sequence[Tuple[int, int]], box_threshold: float,
                    nms_threshold: float, capacity: int):
        """``ovm_infer``: preprocess -> backbone -> GroundingDINO engine (internal side stream) -> output glue -> cube head ->
        postprocess for ONE image in one C call. Returns (records [n, 48], n)."""
        self._require()
        ids = (C.c_int32 * len(token_ids))(*[int(i) for i in token_ids])
        flat = [int(v) for sp in spans for v in sp]
        sp = (C.c_int32 * max(len(flat), 1))(*flat)
        rec = torch.empty((max(capacity, 1), OVM_REC_FLOATS), dtype=torch.float32, device=self.device)
        n = C.c_int32(0)
        rc = self._lib.ovm_infer(self._h, gdino_handle, images, ids, len(token_ids), sp, len(spans), float(box_threshold), float(nms_threshold),
                                 rec.data_ptr(), int(capacity), C.byref(n), self._stream())
        check(rc, self._h, "ovm_infer")
        return rec[: n.value], int(n.value)

    def set_corun(self, on: bool = True) -> None:
        """Other work runs on a second stream beside the backbone (ROIHeads3DGDINO's detector): attention keeps to one
        workgroup per CU so that stream's short kernels are not locked out. Scheduling only."""
        self._require()
        if getattr(self, "_corun", None) != bool(on):
            check(self._lib.ovm_set_corun(self._h, int(on)), self._h, "ovm_set_corun")
            self._corun = bool(on)

    def profile_enable(self, on: bool = True, only=None) -> None:
        """HIP-event brackets around the ViT's kernels on the launch stream; ``only`` = category names to bracket (default: all)."""
        self._require()
        v = int(bool(on))
        if on and only is not None:
            v = 0
            for n in only:
                v |= 2 << _lib.PROF_NAMES.index(n)
        check(self._lib.ovm_profile_enable(self._h, v), self._h, "ovm_profile_enable")

    def profile_read(self) -> Dict[str, Tuple[float, int]]:
        """{category: (total ms, launches)} since profile_enable(True); synchronises the device."""
        self._require()
        ms = (C.c_float * len(_lib.PROF_NAMES))()
        cnt = (C.c_int32 * len(_lib.PROF_NAMES))()
        check(self._lib.ovm_profile_read(self._h, ms, cnt), self._h, "ovm_profile_read")
        return {n: (float(ms[i]), int(cnt[i])) for i, n in enumerate(_lib.PROF_NAMES)}

    def rpn_box_forward(self, images, B: int):
        self._require()
        dev = self.device
        k = self.ncfg.detections_per_image
        nc = self.ncfg.num_classes
        boxes = torch.empty((B * k, 4), dtype=torch.float32, device=dev)
        scores = torch.empty(B * k, dtype=torch.float32, device=dev)
        classes = torch.empty(B * k, dtype=torch.int32, device=dev)
        idx = torch.empty(B * k, dtype=torch.int32, device=dev)
        full = torch.empty((B * k, nc), dtype=torch.float32, device=dev)
        counts = torch.zeros(B, dtype=torch.int32, device=dev)
        rc = self._lib.ovm_rpn_box_forward(self._h, images, B, boxes.data_ptr(), scores.data_ptr(), classes.data_ptr(),
                                           idx.data_ptr(), full.data_ptr(), counts.data_ptr(), self._stream())
        check(rc, self._h, "ovm_rpn_box_forward")
        cl = counts.cpu().tolist()
        n = sum(cl)
        return boxes[:n], scores[:n], classes[:n], idx[:n], full[:n], cl


def records_to_fields(rec: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Slice [n,48] records into the Instances fields of reference roi_heads.py:823-843 (views)."""
    n = rec.shape[0]
    return {
        "pred_boxes": rec[:, 0:4],
        "scores": rec[:, 4],
        # class index column: reinterpret the (row-contiguous) record block as int32, one strided -> int64 conversion kernel
        # (rec[:, 5].contiguous() cost 0.44 ms of host time per image on ROCm: a strided fp32 gather takes torch's slow copy path)
        "pred_classes": (rec.view(torch.int32) if rec.is_contiguous() else rec.contiguous().view(torch.int32))[:, 5].to(torch.int64),
        "pred_bbox3D": rec[:, 6:30].reshape(n, 8, 3),
        "pred_center_cam": rec[:, 30:33],
        "pred_center_2D": rec[:, 33:35],
        "pred_dimensions": rec[:, 35:38],
        "pred_pose": rec[:, 38:47].reshape(n, 3, 3),
    }
