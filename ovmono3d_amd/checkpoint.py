"""Checkpoint loading for the native path.

Mirrors the reference's ``DetectionCheckpointer(model, save_dir).resume_or_load(cfg.MODEL.WEIGHTS, resume=True)``
(demo/demo.py:148, tools/train_net.py:442). A checkpoint is ``{"model": state_dict, ...}`` whose keys follow the
module tree of reference nohup.out:563-684. Files are read with loaders that execute nothing from the file
(``torch.load(weights_only=True)`` or safetensors); the state_dict is handed to ``ovm_create`` which packs it
into device-resident fp16(-split) GEMM layouts.
"""
from __future__ import annotations

import os
from typing import Dict

import torch


def load_state_dict_file(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict):
        obj = obj["model"]
    return {k: v for k, v in obj.items() if isinstance(v, torch.Tensor)}


class DetectionCheckpointer:
    def __init__(self, model, save_dir: str = "", **kw):
        self.model = model
        self.save_dir = save_dir

    def load(self, path: str):
        if not path:
            raise ValueError("MODEL.WEIGHTS is empty: the native path needs a checkpoint "
                             "(or ovmono3d_amd.util.synth_weights for random-init benchmarking)")
        if path.startswith("synthetic://"):
            # synthetic://vitl14?seed=0  - random-init weights of the exact architecture (bench / smoke)
            from .util.synth_weights import synth_state_dict
            spec = path[len("synthetic://"):]
            name, _, q = spec.partition("?")
            seed = int(q.split("=")[1]) if q.startswith("seed=") else 0
            cfg = self.model.cfg
            default = {"build_clip_backbone": cfg.MODEL.CLIP.ARCH, "build_mae_backbone": cfg.MODEL.MAE.CHECKPOINT,
                       "build_midas_backbone": cfg.MODEL.MIDAS.ARCH, "build_sam_backbone": cfg.MODEL.SAM.ARCH}.get(
                cfg.MODEL.BACKBONE.NAME, cfg.MODEL.DINO.MODEL_NAME)
            sd = synth_state_dict(name or default,
                                  num_classes=self.model.cfg.MODEL.ROI_HEADS.NUM_CLASSES, seed=seed)
        else:
            if not os.path.isfile(path):
                raise FileNotFoundError(path)
            sd = load_state_dict_file(path)
        self.model.load_state_dict(sd)
        return {"model": None}

    def resume_or_load(self, path: str, *, resume: bool = True):
        last = os.path.join(self.save_dir, "last_checkpoint") if self.save_dir else ""
        if resume and last and os.path.isfile(last):
            with open(last) as f:
                path = os.path.join(self.save_dir, f.read().strip())
        return self.load(path)
