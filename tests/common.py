"""Shared builders for tests: config + seeded synthetic checkpoint + seeded inputs + oracle params."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ovmono3d_amd.defaults import make_cfg  # noqa: E402
from ovmono3d_amd.util.synth_weights import CLIP_ARCH, MAE_ARCH, MIDAS_ARCH, SAM_ARCH, VIT_ARCH, synth_state_dict  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def build_cfg(model_name="vittest14", canvas=224, precision="f16x3", max_batch=2, max_rois=1000, roi_heads="ROIHeads3D",
              extra=()):
    opts = ["MODEL.DINO.MODEL_NAME", model_name, "MODEL.FPN.SQUARE_PAD", canvas, "MODEL.AMD.GEMM_PRECISION", precision,
            "MODEL.AMD.MAX_BATCH", max_batch, "MODEL.AMD.MAX_ROIS", max_rois, "MODEL.ROI_HEADS.NAME", roi_heads]
    opts += list(extra)
    return make_cfg("OVMono3D_dinov2_SFP.yaml", opts)


def build_clip_cfg(arch="ViT-test-16", canvas=256, precision="f16x3", max_batch=2, max_rois=1000, roi_heads="ROIHeads3D", extra=()):
    opts = ["MODEL.CLIP.ARCH", arch, "MODEL.FPN.SQUARE_PAD", canvas, "MODEL.AMD.GEMM_PRECISION", precision,
            "MODEL.AMD.MAX_BATCH", max_batch, "MODEL.AMD.MAX_ROIS", max_rois, "MODEL.ROI_HEADS.NAME", roi_heads]
    opts += list(extra)
    return make_cfg("OVMono3D_clip_SFP.yaml", opts)


def build_mae_cfg(checkpoint="test/vit-mae-test", canvas=256, precision="f16x3", max_batch=2, max_rois=1000, roi_heads="ROIHeads3D", extra=()):
    opts = ["MODEL.MAE.CHECKPOINT", checkpoint, "MODEL.FPN.SQUARE_PAD", canvas, "MODEL.AMD.GEMM_PRECISION", precision,
            "MODEL.AMD.MAX_BATCH", max_batch, "MODEL.AMD.MAX_ROIS", max_rois, "MODEL.ROI_HEADS.NAME", roi_heads]
    opts += list(extra)
    return make_cfg("OVMono3D_mae_SFP.yaml", opts)


def build_midas_cfg(arch="DPT_test", canvas=256, precision="f16x3", max_batch=2, max_rois=1000, roi_heads="ROIHeads3D", extra=()):
    opts = ["MODEL.MIDAS.ARCH", arch, "MODEL.FPN.SQUARE_PAD", canvas, "MODEL.AMD.GEMM_PRECISION", precision,
            "MODEL.AMD.MAX_BATCH", max_batch, "MODEL.AMD.MAX_ROIS", max_rois, "MODEL.ROI_HEADS.NAME", roi_heads]
    opts += list(extra)
    return make_cfg("OVMono3D_midas_SFP.yaml", opts)


def build_sam_cfg(arch="vit_test", canvas=256, precision="f16x3", max_batch=2, max_rois=1000, roi_heads="ROIHeads3D", extra=()):
    opts = ["MODEL.SAM.ARCH", arch, "MODEL.FPN.SQUARE_PAD", canvas, "MODEL.AMD.GEMM_PRECISION", precision,
            "MODEL.AMD.MAX_BATCH", max_batch, "MODEL.AMD.MAX_ROIS", max_rois, "MODEL.ROI_HEADS.NAME", roi_heads]
    opts += list(extra)
    return make_cfg("OVMono3D_sam_SFP.yaml", opts)


def oracle_params(cfg):
    from oracle.pipeline import OracleParams
    if cfg.MODEL.BACKBONE.NAME in ("build_clip_backbone", "build_mae_backbone", "build_midas_backbone", "build_sam_backbone"):
        extra = {}
        if cfg.MODEL.BACKBONE.NAME == "build_sam_backbone":
            name, tower = cfg.MODEL.SAM.ARCH, "sam"
            D, L, h, patch, _, ws, glob = SAM_ARCH[name]
            extra = dict(sam_window=ws, sam_global=tuple(glob))
        elif cfg.MODEL.BACKBONE.NAME == "build_clip_backbone":
            name, tower = cfg.MODEL.CLIP.ARCH, "clip"
            D, L, h, patch, _ = CLIP_ARCH[name]
        elif cfg.MODEL.BACKBONE.NAME == "build_midas_backbone":
            name, tower = cfg.MODEL.MIDAS.ARCH, "midas"
            D, L, h, patch, _ = MIDAS_ARCH[name]
        else:
            name, tower = cfg.MODEL.MAE.CHECKPOINT, "mae"
            D, L, h, patch = MAE_ARCH[name]
        return OracleParams(model_name=name, tower=tower, embed_dim=D, depth=L, heads=h,
                            square_pad=cfg.MODEL.FPN.SQUARE_PAD, pixel_mean=tuple(cfg.MODEL.PIXEL_MEAN),
                            pixel_std=tuple(cfg.MODEL.PIXEL_STD), use_depth_fusion=False,
                            strides=(patch // 4, patch // 2, patch, patch * 2),
                            anchor_sizes=tuple(float(s[0]) for s in cfg.MODEL.ANCHOR_GENERATOR.SIZES),
                            anchor_ratios=tuple(cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS[0]),
                            rpn_pre_topk=cfg.MODEL.RPN.PRE_NMS_TOPK_TEST, rpn_post_topk=cfg.MODEL.RPN.POST_NMS_TOPK_TEST,
                            rpn_nms=cfg.MODEL.RPN.NMS_THRESH, score_thresh=cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
                            nms_thresh=cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST, topk=cfg.TEST.DETECTIONS_PER_IMAGE,
                            virtual_focal=cfg.MODEL.ROI_CUBE_HEAD.VIRTUAL_FOCAL, pooler_min_level=2, pooler_max_level=5, **extra)
    D, L, h = VIT_ARCH[cfg.MODEL.DINO.MODEL_NAME]
    return OracleParams(model_name=cfg.MODEL.DINO.MODEL_NAME, embed_dim=D, depth=L, heads=h,
                        square_pad=cfg.MODEL.FPN.SQUARE_PAD, pixel_mean=tuple(cfg.MODEL.PIXEL_MEAN),
                        pixel_std=tuple(cfg.MODEL.PIXEL_STD), use_depth_fusion=cfg.MODEL.DINO.USE_DEPTH_FUSION,
                        anchor_sizes=tuple(float(s[0]) for s in cfg.MODEL.ANCHOR_GENERATOR.SIZES),
                        anchor_ratios=tuple(cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS[0]),
                        rpn_pre_topk=cfg.MODEL.RPN.PRE_NMS_TOPK_TEST, rpn_post_topk=cfg.MODEL.RPN.POST_NMS_TOPK_TEST,
                        rpn_nms=cfg.MODEL.RPN.NMS_THRESH, score_thresh=cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
                        nms_thresh=cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST, topk=cfg.TEST.DETECTIONS_PER_IMAGE,
                        virtual_focal=cfg.MODEL.ROI_CUBE_HEAD.VIRTUAL_FOCAL,
                        pooler_min_level=cfg.MODEL.ROI_HEADS.POOLER_MIN_LEVEL,
                        pooler_max_level=cfg.MODEL.ROI_HEADS.POOLER_MAX_LEVEL)


def synth_inputs(n_images=1, hw=((140, 196),), orig_scale=2.0, n_boxes=12, seed=0, oracle2d=True, depth=False):
    """Seeded uint8 CHW images, K as demo.py:63-76 (focal 4.0*h/2), random oracle-2D boxes in
    original-resolution coordinates (schema reference cubercnn/data/build.py:51-54)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n_images):
        h, w = hw[i % len(hw)]
        img = torch.randint(0, 256, (3, h, w), dtype=torch.uint8, generator=g)
        oh, ow = int(round(h * orig_scale)), int(round(w * orig_scale))
        f = 4.0 * oh / 2
        K = [[f, 0.0, ow / 2], [0.0, f, oh / 2], [0.0, 0.0, 1.0]]
        d = {"image": img, "height": oh, "width": ow, "K": K, "image_id": i}
        if oracle2d:
            x1 = torch.rand(n_boxes, generator=g) * (ow * 0.7)
            y1 = torch.rand(n_boxes, generator=g) * (oh * 0.7)
            bw = (0.06 + torch.rand(n_boxes, generator=g) * 0.5) * ow
            bh = (0.06 + torch.rand(n_boxes, generator=g) * 0.5) * oh
            boxes = torch.stack([x1, y1, torch.minimum(x1 + bw, torch.tensor(float(ow))),
                                 torch.minimum(y1 + bh, torch.tensor(float(oh)))], 1)
            d["oracle2D"] = {"gt_bbox2D": boxes, "gt_classes": torch.randint(0, 50, (n_boxes,), generator=g),
                             "gt_scores": 0.3 + 0.7 * torch.rand(n_boxes, generator=g)}
        if depth:
            d["depth"] = torch.rand(1, 60, 80, generator=g) * 5.0
        out.append(d)
    return out


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b|  (the scale-relative error all float parity checks use)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def assert_close(a, b, tol, name=""):
    assert tuple(a.shape) == tuple(b.shape), f"{name}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e = rel_err(a, b)
    assert e <= tol, f"{name}: scale-relative error {e:.3e} > {tol:.1e}"
    return e


# ------------------------------------------------------------------------------------------ golden
def load_golden(name):
    """Returns (inputs, expected per-image dicts, raw npz) for a fixture written by tests/golden/make_golden.py."""
    z = np.load(os.path.join(GOLDEN, name))
    n = int(z["n_images"])
    inputs, expected = [], []
    for i in range(n):
        h, w = (int(v) for v in z[f"in_meta{i}"])
        d = {"image": torch.from_numpy(z[f"in_img{i}"]), "height": h, "width": w, "K": z[f"in_K{i}"].tolist(), "image_id": i}
        if f"in_box{i}" in z:
            d["oracle2D"] = {"gt_bbox2D": torch.from_numpy(z[f"in_box{i}"]), "gt_classes": torch.from_numpy(z[f"in_cls{i}"]),
                             "gt_scores": torch.from_numpy(z[f"in_sc{i}"])}
        if f"in_depth{i}" in z:
            d["depth"] = torch.from_numpy(z[f"in_depth{i}"])
        inputs.append(d)
        expected.append({k[len(f"out{i}_"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"out{i}_")})
    return inputs, expected, z


def golden_weights(z):
    """Re-creates the fixture's synthetic checkpoint and verifies its fingerprint (guards against a
    change of torch's CPU RNG stream, which would invalidate the stored outputs)."""
    sd = synth_state_dict("vittest14", seed=int(z["weights_seed"]))
    keys = sorted(sd)[:: max(1, len(sd) // 16)]
    fp = np.array([float(sd[k].double().sum()) for k in keys])
    ok = np.allclose(fp, z["weights_fp"], rtol=1e-9, atol=1e-9)
    return sd, ok
