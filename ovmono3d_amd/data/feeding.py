"""Data feeding for the entry points (image read, ResizeShortestEdge, Omni3D JSON, oracle-2D merge, contiguous per-rank
shards). With a HIP device the pixel work runs there: baseline JPEGs are reconstructed on the device from
host-decoded coefficients (gpu_jpeg.py), ``ResizeShortestEdge`` and the depth-prompt resizes are device kernels (gpu_resize.py);
the plain host code below is what the entry points use without one. The OUTPUT schema is the hot path's input
contract: per image ``{"image": uint8 CHW, "height", "width", "K", "image_id", ["oracle2D"], ["depth"]}``
(reference cubercnn/data/dataset_mapper.py:33-80, cubercnn/data/build.py:45-54,281-311, demo/demo.py:46-85).
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from .. import lib as _lib


def read_image(path: str, fmt: str = "RGB") -> np.ndarray:
    """uint8 HxWx3. fmt 'BGR' mirrors cv2.imread (reference util.imread used at demo/demo.py:52)."""
    from PIL import Image, ImageOps
    with Image.open(path) as im:
        # both readers of the reference honour the Exif Orientation tag: cv2.imread (demo/demo.py:52) and detectron2's read_image
        # (_apply_exif_orientation, dataset_mapper.py:38)
        arr = np.asarray(ImageOps.exif_transpose(im).convert("RGB"))
    return arr[:, :, ::-1].copy() if fmt == "BGR" else arr


class ResizeShortestEdge:
    """detectron2 ``T.ResizeShortestEdge(min_size, max_size, "choice")`` for inference (one size)."""

    def __init__(self, short_edge_length: int, max_size: int):
        self.size, self.max_size = int(short_edge_length), int(max_size)

    def output_shape(self, h: int, w: int):
        scale = self.size * 1.0 / min(h, w)
        newh, neww = (self.size, scale * w) if h < w else (scale * h, self.size)
        if max(newh, neww) > self.max_size:
            s = self.max_size * 1.0 / max(newh, neww)
            newh, neww = newh * s, neww * s
        return int(newh + 0.5), int(neww + 0.5)

    def __call__(self, img: np.ndarray) -> np.ndarray:
        from PIL import Image
        h, w = img.shape[:2]
        nh, nw = self.output_shape(h, w)
        if (nh, nw) == (h, w):
            return img
        if img.dtype == np.uint8:
            return np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
        t = torch.from_numpy(np.ascontiguousarray(img)).float()
        t = t[None, None] if t.dim() == 2 else t.permute(2, 0, 1)[None]
        t = torch.nn.functional.interpolate(t, (nh, nw), mode="bilinear", align_corners=False)
        return t[0, 0].numpy() if img.ndim == 2 else t[0].permute(1, 2, 0).numpy()


def load_omni3d_json(path_to_json: str, image_root: str = "datasets") -> List[Dict]:
    """Image-level records of an Omni3D JSON (reference cubercnn/data/datasets.py:370-395)."""
    with open(path_to_json) as f:
        data = json.load(f)
    out = []
    for img in data["images"]:
        out.append({"file_name": os.path.join(image_root, img["file_path"]), "dataset_id": img.get("dataset_id", 0),
                    "height": img["height"], "width": img["width"], "K": img["K"], "image_id": img["id"]})
    return out


def xywh_to_xyxy(b):
    x, y, w, h = b
    return [x, y, x + w, y + h]


def merge_oracle2d_to_detection_dicts(dataset_dicts: List[Dict], oracle_json: str) -> None:
    """reference cubercnn/data/build.py:45-54 (same on-disk schema: list of {image_id, instances[{bbox xywh,
    category_id, score}]} aligned with the dataset order)."""
    with open(oracle_json) as f:
        oracle = json.load(f)
    for d, o in zip(dataset_dicts, oracle):
        assert d["image_id"] == o["image_id"]
        inst = o["instances"]
        d["oracle2D"] = {
            "gt_bbox2D": torch.tensor([xywh_to_xyxy(i["bbox"]) for i in inst], dtype=torch.float32).reshape(-1, 4),
            "gt_classes": torch.tensor([i["category_id"] for i in inst], dtype=torch.int64),
            "gt_scores": torch.tensor([i["score"] for i in inst], dtype=torch.float32),
        }


class DatasetMapper3D:
    """Inference branch of reference cubercnn/data/dataset_mapper.py:18-80."""

    def __init__(self, cfg, is_train: bool = False, depth_dir: Optional[str] = None):
        assert not is_train, "training is out of scope"
        self.image_format = cfg.INPUT.FORMAT
        self.resize = ResizeShortestEdge(cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
        # only the DINOv2 tower has the depth-fusion conv; the other backbones take no depth (backbone/clip.py), so no depth
        # prompt is loaded for them even when a folder is given
        self.use_depth = (bool(cfg.MODEL.DINO.USE_DEPTH_FUSION) and cfg.MODEL.BACKBONE.NAME == "build_dino_backbone"
                          and depth_dir is not None)
        self.depth_dir = depth_dir
        # device-side ResizeShortestEdge (bit-identical to the Pillow path; gpu_resize.py) when a HIP device is present
        self.gpu_resize = None
        if bool(cfg.MODEL.AMD.get("GPU_RESIZE", False)) and str(cfg.MODEL.DEVICE).startswith("cuda") and torch.cuda.is_available():
            from .gpu_resize import DepthPromptResizeGPU, ResizeShortestEdgeGPU
            self.gpu_resize = ResizeShortestEdgeGPU(cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
            self.gpu_depth_resize = DepthPromptResizeGPU(cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
        # baseline JPEGs reconstructed on the device from host-decoded coefficients (gpu_jpeg.py; bit-identical to the Pillow read)
        self.gpu_jpeg = self.gpu_resize is not None and bool(cfg.MODEL.AMD.get("GPU_JPEG", False))

    def __call__(self, d: Dict) -> Dict:
        d = dict(d)
        if self.gpu_jpeg:
            from .gpu_jpeg import read_image_device
            image = read_image_device(d["file_name"], self.image_format, torch.device("cuda"))      # uint8 [H, W, 3] in HBM
        else:
            image = read_image(d["file_name"], self.image_format)
        depth = None
        if self.use_depth:
            base = os.path.splitext(os.path.basename(d["file_name"]))[0]
            try:
                depth = np.load(os.path.join(self.depth_dir, "test", base + ".npz"))["depth"].astype("float32")
                if self.gpu_resize is not None:
                    pass                                                       # both resizes run on the device below
                elif depth.shape[:2] != image.shape[:2]:
                    depth = torch.nn.functional.interpolate(torch.from_numpy(depth)[None, None], size=tuple(image.shape[:2]),
                                                            mode="bilinear", align_corners=False)[0, 0].numpy()
            except Exception:
                depth = np.zeros(tuple(image.shape[:2]), dtype=np.float32)
        if self.gpu_resize is not None:
            dev_img = image if torch.is_tensor(image) else torch.from_numpy(np.ascontiguousarray(image)).cuda()
            d["image"] = self.gpu_resize(dev_img).permute(2, 0, 1)   # CHW view of the HWC result
        else:
            image = self.resize(image)
            d["image"] = torch.as_tensor(np.ascontiguousarray(image.transpose(2, 0, 1)))
        if depth is not None and self.gpu_resize is not None:
            d["depth"] = self.gpu_depth_resize(torch.from_numpy(np.ascontiguousarray(depth)).cuda(), tuple(image.shape[:2])).unsqueeze(0)
        elif depth is not None:
            d["depth"] = torch.as_tensor(np.ascontiguousarray(self.resize(depth))).unsqueeze(0)
        return d


class _ShardLoader:
    """batch-size-1 loader over this rank's contiguous shard (InferenceSampler + BatchSampler(1),
    reference cubercnn/data/build.py:313-330); also forwards ``oracle2D`` (the fork's collate drops it,
    build.py:281-311 - see SURVEY.md Appendix C D3; upstream behaviour is kept here)."""

    KEYS = ("image", "height", "width", "image_id", "depth", "file_name", "dataset_id", "K", "oracle2D")

    def __init__(self, dataset: Sequence[Dict], mapper, rank: int, world: int, batch_size: int = 1):
        self.dataset, self.mapper = dataset, mapper
        self.begin, self.end = _lib.shard_range(len(dataset), rank, world)
        self.batch_size = batch_size

    def __len__(self) -> int:
        n = self.end - self.begin
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[List[Dict]]:
        for s in range(self.begin, self.end, self.batch_size):
            batch = []
            for i in range(s, min(s + self.batch_size, self.end)):
                rec = self.mapper(self.dataset[i])
                batch.append({k: rec[k] for k in self.KEYS if k in rec})
            yield batch


def build_detection_test_loader(cfg, dataset: Sequence[Dict], mapper=None, rank: int = 0, world: int = 1, batch_size: int = 1):
    return _ShardLoader(dataset, mapper or DatasetMapper3D(cfg, False), rank, world, batch_size)
