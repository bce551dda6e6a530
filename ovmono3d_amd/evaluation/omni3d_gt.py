"""Ground truth of an Omni3D annotation file the way the reference's evaluator sees it.

What the reference does (cubercnn/data/datasets.py) and where this module restates it:

* ``get_filter_settings_from_cfg`` :52-78 and the overrides ``do_test`` applies before evaluating
  (tools/train_net.py:59-63, omni3d_evaluation.py:254)                      -> ``filter_settings_from_cfg`` / ``eval_filter_settings``
* ``is_ignore`` :81-123 (one flag feeds ignore, ignore2D and ignore3D :256-258) -> ``annotation_ignored``
* ``Omni3D(COCO).__init__`` :146-274 (category subset, 2D-box choice, area / bbox3D / depth fields, annotations of
  categories outside ``category_names | ignore_names`` dropped, annotations without any 2D box dropped)
                                                                                -> ``Omni3DGroundTruth``
* ``register_and_store_model_metadata`` :294-320 and the id un-mapping of the evaluator
  (omni3d_evaluation.py:1015-1075)                                              -> ``CategoryMap``

Everything here is host bookkeeping over JSON (no device work); the IoU kernels are called from ``omni3d_eval.py``.
"""
from __future__ import annotations

import json
from typing import Dict, Iterable, List, Optional, Sequence, Union

import numpy as np

_FILTER_DEFAULTS = {
    "category_names": [], "ignore_names": [], "truncation_thres": 0.99, "visibility_thres": 0.01, "min_height_thres": 0.0,
    "max_height_thres": 1.5, "modal_2D_boxes": False, "trunc_2D_boxes": False, "max_depth": 1e8,
}


def filter_settings_from_cfg(cfg=None) -> Dict:
    """datasets.py:52-78. Without a cfg: the literal defaults (trunc_2D_boxes False there, True in the cfg defaults)."""
    fs = {k: (list(v) if isinstance(v, list) else v) for k, v in _FILTER_DEFAULTS.items()}
    if cfg is None:
        return fs
    D = cfg.DATASETS
    fs.update(category_names=list(D.CATEGORY_NAMES), ignore_names=list(D.IGNORE_NAMES), truncation_thres=float(D.TRUNCATION_THRES),
              visibility_thres=float(D.VISIBILITY_THRES), min_height_thres=float(D.MIN_HEIGHT_THRES), modal_2D_boxes=bool(D.MODAL_2D_BOXES),
              trunc_2D_boxes=bool(D.TRUNC_2D_BOXES), max_depth=float(D.MAX_DEPTH))
    return fs                                                                  # max_height_thres stays 1.50 (:77)


def eval_filter_settings(cfg, mode: str = "base") -> Dict:
    """The settings ``do_test`` evaluates with (tools/train_net.py:59-70): TEST thresholds, min height 1/16 of the image,
    no depth limit, and the evaluated category list of the mode (omni3d_evaluation.py:254)."""
    fs = filter_settings_from_cfg(cfg)
    fs["visibility_thres"] = float(cfg.TEST.VISIBILITY_THRES)
    fs["truncation_thres"] = float(cfg.TEST.TRUNCATION_THRES)
    fs["min_height_thres"] = 0.0625
    fs["max_depth"] = 1e8
    if mode == "novel":
        fs["category_names"] = list(cfg.DATASETS.CATEGORY_NAMES_NOVEL)
    elif mode == "base":
        fs["category_names"] = list(cfg.DATASETS.CATEGORY_NAMES_BASE)
    else:
        raise ValueError("wrong mode")
    return fs


def _xyxy_to_xywh(b: Sequence[float]) -> List[float]:
    return [b[0], b[1], b[2] - b[0], b[3] - b[1]]


def _all_minus_one(b: Sequence[float]) -> bool:
    return all(v == -1 for v in b)


def annotation_ignored(anno: Dict, fs: Dict, image_height: float) -> bool:
    """datasets.py:81-123. The 3D-validity tests short-circuit; every later test may only turn the flag on."""
    if anno["behind_camera"] or not bool(anno["valid3D"]):
        return True
    dims = anno["dimensions"]
    bad = (dims[0] <= 0 or dims[1] <= 0 or dims[2] <= 0 or anno["center_cam"][2] > fs["max_depth"] or anno["lidar_pts"] == 0
           or anno["segmentation_pts"] == 0 or anno["depth_error"] > 0.5)
    # the 2D box whose height is screened: tight (modal) > truncated > projected > plain 'bbox' (:98-111); note that the rule
    # for the box STORED as ground truth (``_gt_box_xywh`` below) is a different one
    if fs["modal_2D_boxes"] and "bbox2D_tight" in anno and anno["bbox2D_tight"][0] != -1:
        box = _xyxy_to_xywh(anno["bbox2D_tight"])
    elif fs["trunc_2D_boxes"] and "bbox2D_trunc" in anno and not _all_minus_one(anno["bbox2D_trunc"]):
        box = _xyxy_to_xywh(anno["bbox2D_trunc"])
    elif "bbox2D_proj" in anno:
        box = _xyxy_to_xywh(anno["bbox2D_proj"])
    else:
        box = anno["bbox"]
    bad = bad or box[3] <= fs["min_height_thres"] * image_height or box[3] >= fs["max_height_thres"] * image_height
    bad = bad or (anno["truncation"] >= 0 and anno["truncation"] >= fs["truncation_thres"])
    bad = bad or (anno["visibility"] >= 0 and anno["visibility"] <= fs["visibility_thres"])
    if "ignore_names" in fs:
        bad = bad or anno["category_name"] in fs["ignore_names"]
    return bool(bad)


def _gt_box_xywh(anno: Dict, fs: Dict) -> Optional[List[float]]:
    """datasets.py:241-266: truncated box when enabled and present, else projected, else tight, else the annotation is
    dropped; with modal boxes enabled a present tight box replaces the stored box (area keeps the first choice)."""
    if fs["trunc_2D_boxes"] and "bbox2D_trunc" in anno and not _all_minus_one(anno["bbox2D_trunc"]):
        return _xyxy_to_xywh(anno["bbox2D_trunc"])
    if anno["bbox2D_proj"][0] != -1:
        return _xyxy_to_xywh(anno["bbox2D_proj"])
    if anno["bbox2D_tight"][0] != -1:
        return _xyxy_to_xywh(anno["bbox2D_tight"])
    return None


class Omni3DGroundTruth:
    """Filtered annotations of one or several Omni3D JSON files (paths or already-loaded dicts).

    ``categories``   evaluated categories, sorted by dataset id (subset ``fs['category_names']`` when that is non-empty)
    ``annotations``  kept annotations with ``bbox`` (xywh), ``area``, ``bbox3D``, ``depth``, ``iscrowd``, ``ignore`` /
                     ``ignore2D`` / ``ignore3D``; category ids stay the DATASET ids
    ``images``       the image entries; ``image_ids`` their ids

    With ``filter_settings=None`` the files are taken as they are (datasets.py:194-201): no field is derived, so such an
    object serves the category table only.
    """

    def __init__(self, annotation_files: Union[str, Dict, Iterable[Union[str, Dict]]], filter_settings: Optional[Dict] = None):
        if isinstance(annotation_files, (str, dict)):
            annotation_files = [annotation_files]
        images, annos, master = [], [], {}
        self.info = []
        for src in annotation_files:
            if isinstance(src, str):
                with open(src) as f:
                    src = json.load(f)
            info = src.get("info", {})
            info = dict(info[0] if isinstance(info, list) else info)
            info["known_category_ids"] = [c["id"] for c in src["categories"]]
            self.info.append(info)
            images += src["images"]
            annos += src["annotations"]
            for c in src["categories"]:
                master.setdefault(c["id"], c)                                  # first file that names an id wins (:188-192)
        ordered = [master[i] for i in sorted(master)]
        self.images = images
        self.all_categories = ordered                                          # before the category_names subset
        self.filter_settings = filter_settings
        if filter_settings is None:
            self.categories, self.annotations = ordered, [dict(a) for a in annos]
        else:
            fs = filter_settings
            keep_names = set(fs["ignore_names"]) | set(fs["category_names"])
            if len(fs["category_names"]) > 0:
                self.categories = [c for c in ordered if c["name"] in fs["category_names"]]
            else:                                                              # no list given: every category of the files (:218-226)
                self.categories = ordered
                fs["category_names"] = [c["name"] for c in ordered]
                keep_names |= set(fs["category_names"])
            heights = {im["id"]: im["height"] for im in images}
            self.annotations = []
            for a in annos:
                ign = annotation_ignored(a, fs, heights[a["image_id"]])
                box = _gt_box_xywh(a, fs)
                if box is None:
                    continue
                g = dict(a)
                g["area"] = box[2] * box[3]
                g["iscrowd"] = False
                g["ignore"] = g["ignore2D"] = g["ignore3D"] = ign
                g["bbox"] = _xyxy_to_xywh(a["bbox2D_tight"]) if fs["modal_2D_boxes"] and a["bbox2D_tight"][0] != -1 else box
                g["bbox3D"] = a["bbox3D_cam"]
                g["depth"] = a["center_cam"][2]
                if a["category_name"] in keep_names:
                    self.annotations.append(g)
        self.image_ids = [im["id"] for im in images]
        self.category_ids = [c["id"] for c in self.categories]
        self.category_names = [c["name"] for c in self.categories]

    def __len__(self):
        return len(self.annotations)


class CategoryMap:
    """Dataset category id <-> contiguous class index of the model.

    The model's classes are ``thing_classes`` sorted by their dataset id, index = rank (datasets.py:294-320). Sources:
    ``from_names`` (names + the ``categories`` table of a dataset / stats file) or ``from_meta`` (a ``category_meta.json``
    style file: ``thing_classes`` + ``thing_dataset_id_to_contiguous_id`` with string keys, tools/train_net.py:404-416).
    """

    def __init__(self, thing_classes: Sequence[str], dataset_id_to_contiguous: Dict[int, int]):
        self.thing_classes = list(thing_classes)
        self.dataset_id_to_contiguous = {int(k): int(v) for k, v in dataset_id_to_contiguous.items()}
        self.contiguous_to_dataset_id = {v: k for k, v in self.dataset_id_to_contiguous.items()}
        if len(self.contiguous_to_dataset_id) != len(self.dataset_id_to_contiguous):
            raise ValueError("thing_dataset_id_to_contiguous_id is not one-to-one")

    @classmethod
    def from_names(cls, names: Sequence[str], categories: Sequence[Dict]) -> "CategoryMap":
        by_name = {c["name"]: c["id"] for c in categories}
        missing = [n for n in names if n not in by_name]
        if missing:
            raise KeyError(f"categories {missing} are not in the category table")
        pairs = sorted((by_name[n], n) for n in names)
        return cls([n for _, n in pairs], {cid: i for i, (cid, _) in enumerate(pairs)})

    @classmethod
    def from_meta(cls, path_or_dict: Union[str, Dict]) -> "CategoryMap":
        meta = path_or_dict
        if isinstance(meta, str):
            with open(meta) as f:
                meta = json.load(f)
        return cls(meta["thing_classes"], meta["thing_dataset_id_to_contiguous_id"])

    def to_meta(self) -> Dict:
        return {"thing_classes": self.thing_classes,
                "thing_dataset_id_to_contiguous_id": {str(k): v for k, v in sorted(self.dataset_id_to_contiguous.items())}}

    def detections_to_dataset_ids(self, detections: Sequence[Dict], passthrough_dataset_ids: bool = False) -> List[Dict]:
        """Detections carry the model's contiguous class index; ground truth carries dataset ids. Returns copies with
        ``category_id`` un-mapped; detections of a class outside the map are dropped (omni3d_evaluation.py:1052-1093).

        ``passthrough_dataset_ids=True`` reproduces the fork's extra rule (:1046-1050): an id that happens to be a KEY of
        the map is taken for a dataset id and kept. With the 9-class Objectron map (keys 11,14..21, indices 0..8) the two
        readings agree; with a map whose keys overlap its indices (the 50-class one) the fork's rule mislabels classes,
        so it is off by default.
        """
        out = []
        for d in detections:
            cid = int(d["category_id"])
            if passthrough_dataset_ids and cid in self.dataset_id_to_contiguous:
                out.append(dict(d))
            elif cid in self.contiguous_to_dataset_id:
                out.append(dict(d, category_id=self.contiguous_to_dataset_id[cid]))
        return out


def ground_truth_records(gt: Omni3DGroundTruth) -> List[Dict]:
    """The fields the AP evaluator reads, one dict per kept annotation (plus what the NHD needs: centre, dimensions, R)."""
    keys = ("id", "image_id", "category_id", "bbox", "area", "bbox3D", "depth", "iscrowd", "ignore2D", "ignore3D", "center_cam", "dimensions",
            "R_cam")
    return [{k: a[k] for k in keys if k in a} for a in gt.annotations]


def bbox3d_is_finite(b) -> bool:
    a = np.asarray(b, dtype=np.float64)
    return a.shape == (8, 3) and bool(np.isfinite(a).all())
