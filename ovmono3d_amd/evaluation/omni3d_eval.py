"""Omni3D AP evaluation with true 3D IoU ("next" row 1 of SURVEY.md 8f).

Upstream Omni3D semantics of the reference's ``Omni3Deval`` (cubercnn/evaluation/omni3d_evaluation.py):
``Omni3DParams`` :1394-1461 (2D: IoU 0.5:0.05:0.95, area ranges; 3D: IoU 0.05:0.05:0.50, depth ranges [0,10,35,1e5]),
``_prepare`` :1515-1545 (ignore2D / ignore3D flags), COCO matching per (image, category, range), ``accumulate``
:1547-1688 (101 recall points, stable sort on scores), ``summarize`` :2072-2224 (AP, AP15 / AP25 / AP50, near / medium /
far), ``Omni3DevalWithNHD`` :2293-2484 (disentangled NHD of IoU-matched pairs, ``nhd.py``). The 3D IoU is
``box3d_overlap`` (:109-169) on the HIP kernel ``ovm_box3d_iou``. Ground-truth construction (filter settings, the ignore
rule, dataset <-> contiguous category ids) is ``omni3d_gt.py``.

The reference FORK lost its ``computeIoU`` override, so its "3D" AP is really computed with pycocotools' 2D IoU
(SURVEY.md 0.5); ``fork_compat_2d_iou=True`` reproduces that behaviour, the default is upstream's.
pycocotools is not a dependency (and not importable here, so AP values are pinned by cases with a known answer and by
a by-definition restatement in tests/test_eval.py, not by running the reference's evaluator: "parity unpinned" for
real-data AP).
"""
from __future__ import annotations

import ctypes as C
from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import lib as _lib
from . import nhd as _nhd


def box3d_overlap(boxes_dt: torch.Tensor, boxes_gt: torch.Tensor, eps_coplanar: float = 1e-4, eps_nonzero: float = 1e-8) -> torch.Tensor:
    """(N, 8, 3) x (M, 8, 3) corners (pytorch3d order) -> (N, M) IoU on the HIP device. Detections that fail the
    reference's coplanarity / non-zero-area screening get IoU 0 (:160-167). No CPU fallback."""
    L = _lib.load()
    if boxes_dt.device.type != "cuda":
        raise RuntimeError("box3d_overlap runs on the HIP device only (no CPU fallback)")
    dt = boxes_dt.to(torch.float32).reshape(-1, 24).contiguous()
    gt = boxes_gt.to(dt.device, torch.float32).reshape(-1, 24).contiguous()
    N, M = int(dt.shape[0]), int(gt.shape[0])
    out = torch.zeros((N, M), dtype=torch.float32, device=dt.device)
    if N and M:
        stream = C.c_void_p(torch.cuda.current_stream(dt.device).cuda_stream)
        _lib.check(L.ovm_box3d_iou(dt.data_ptr(), gt.data_ptr(), N, M, float(eps_coplanar), float(eps_nonzero), out.data_ptr(), None, stream),
                   what="ovm_box3d_iou")
    return out


def iou2d_xywh(dt: np.ndarray, gt: np.ndarray) -> np.ndarray:
    """pycocotools maskUtils.iou for boxes without crowd regions: (N,4) x (M,4) xywh -> (N,M)."""
    if len(dt) == 0 or len(gt) == 0:
        return np.zeros((len(dt), len(gt)))
    d, g = np.asarray(dt, np.float64)[:, None, :], np.asarray(gt, np.float64)[None, :, :]
    iw = np.clip(np.minimum(d[..., 0] + d[..., 2], g[..., 0] + g[..., 2]) - np.maximum(d[..., 0], g[..., 0]), 0, None)
    ih = np.clip(np.minimum(d[..., 1] + d[..., 3], g[..., 1] + g[..., 3]) - np.maximum(d[..., 1], g[..., 1]), 0, None)
    inter = iw * ih
    union = d[..., 2] * d[..., 3] + g[..., 2] * g[..., 3] - inter
    return np.where(union > 0, inter / np.where(union > 0, union, 1), 0.0)


class Omni3DParams:
    """reference :1394-1461"""

    def __init__(self, mode: str = "2D"):
        if mode == "2D":
            self.iouThrs = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)
            self.areaRng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
            self.areaRngLbl = ["all", "small", "medium", "large"]
        elif mode == "3D":
            self.iouThrs = np.linspace(0.05, 0.5, int(np.round((0.5 - 0.05) / 0.05)) + 1, endpoint=True)
            self.areaRng = [[0, 1e5], [0, 10], [10, 35], [35, 1e5]]
            self.areaRngLbl = ["all", "near", "medium", "far"]
        else:
            raise Exception("mode %s not supported" % mode)
        self.recThrs = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)
        self.maxDets = [1, 10, 100]
        self.imgIds: List = []
        self.catIds: List = []
        self.useCats = 1
        self.iouType = "bbox"
        self.mode = mode
        self.proximity_thresh = 0.3


class _Cell:
    """One (image, category) cell: detections in descending score order (stable), cut to the largest maxDets; ground truth
    in file order; ``iou`` [D,G] between them."""
    __slots__ = ("dt", "gt", "iou", "score", "dt_rng", "gt_rng", "gt_flag", "gt_crowd")


class Omni3Deval:
    """COCO-style AP over Omni3D annotations. ``gts`` / ``dts`` are lists of dicts:
    gt: image_id, category_id, bbox [x,y,w,h], bbox3D [8][3], depth, optional area, ignore2D, ignore3D, iscrowd
    dt: image_id, category_id, bbox, score, and for 3D bbox3D + depth (the records of ``instances_to_coco_json``).
    ``img_ids`` / ``cat_ids``: the evaluated images and categories (the reference takes them from the ground-truth dataset,
    :1489-1491); by default every id that occurs. In 3D mode, pairs of a detection and its best-IoU ground truth with
    IoU >= ``nhd_iou_threshold`` also get the disentangled NHD (``Omni3DevalWithNHD`` :2293-2484) when both carry
    centre / dimensions / rotation.

    The evaluation is the published COCOeval procedure (greedy matching per IoU threshold in score order, ignore rules,
    101-point interpolated precision); it is organised here as array operations over the thresholds rather than the
    per-threshold loops of pycocotools, and ``tests/test_eval.py`` holds it against a by-definition restatement."""

    def __init__(self, gts: Sequence[Dict], dts: Sequence[Dict], mode: str = "3D", device: Optional[torch.device] = None,
                 fork_compat_2d_iou: bool = False, img_ids: Optional[Sequence] = None, cat_ids: Optional[Sequence] = None,
                 nhd_iou_threshold: float = 0.5):
        self.mode = mode
        self.params = Omni3DParams(mode)
        self.device = device
        self.fork_compat_2d_iou = fork_compat_2d_iou
        self.nhd_iou_threshold = nhd_iou_threshold
        self._gts_all, self._dts_all = [dict(g) for g in gts], [dict(d) for d in dts]
        for i, g in enumerate(self._gts_all):
            g.setdefault("id", i + 1)
            g.setdefault("iscrowd", 0)
            if "area" not in g:
                g["area"] = float(g["bbox"][2] * g["bbox"][3])
        for i, d in enumerate(self._dts_all):
            d.setdefault("id", i + 1)
            if "area" not in d:
                d["area"] = float(d["bbox"][2] * d["bbox"][3])
        self.params.imgIds = sorted(img_ids) if img_ids is not None else \
            sorted({g["image_id"] for g in self._gts_all} | {d["image_id"] for d in self._dts_all})
        self.params.catIds = sorted(cat_ids) if cat_ids is not None else \
            sorted({g["category_id"] for g in self._gts_all} | {d["category_id"] for d in self._dts_all})
        self.cells: Dict = {}
        self.per_cell: Dict = {}
        self.eval: Dict = {}
        self.nhd_pairs: List[Dict] = []

    # ---- cells and their IoU ------------------------------------------------------------------------------------------
    def _iou_2d(self, dt, gt):
        return iou2d_xywh(np.array([d["bbox"] for d in dt]), np.array([g["bbox"] for g in gt]))

    def _build_cells(self):
        p = self.params
        flag = "ignore2D" if self.mode == "2D" else "ignore3D"                  # _prepare :1515-1545
        rng_key = "area" if self.mode == "2D" else "depth"
        imgs, cats = set(p.imgIds), set(p.catIds)
        by_gt, by_dt = defaultdict(list), defaultdict(list)
        for g in self._gts_all:
            if g["image_id"] in imgs and g["category_id"] in cats:
                by_gt[g["image_id"], g["category_id"]].append(g)
        for d in self._dts_all:
            if d["image_id"] in imgs and d["category_id"] in cats:
                by_dt[d["image_id"], d["category_id"]].append(d)
        top = p.maxDets[-1]
        self.cells = {}
        for key in set(by_gt) | set(by_dt):
            c = _Cell()
            dt = by_dt.get(key, [])
            order = np.argsort(-np.array([d["score"] for d in dt], dtype=np.float64), kind="mergesort")[:top]
            c.dt = [dt[i] for i in order]
            c.gt = by_gt.get(key, [])
            c.score = np.array([d["score"] for d in c.dt], dtype=np.float64)
            c.dt_rng = np.array([d[rng_key] for d in c.dt], dtype=np.float64)
            c.gt_rng = np.array([g[rng_key] for g in c.gt], dtype=np.float64)
            c.gt_flag = np.array([bool(g.get(flag, 0)) for g in c.gt], dtype=bool)
            c.gt_crowd = np.array([bool(g["iscrowd"]) for g in c.gt], dtype=bool)
            c.iou = None
            self.cells[key] = c
        true_3d = self.mode == "3D" and not self.fork_compat_2d_iou
        if not true_3d:
            for c in self.cells.values():
                c.iou = self._iou_2d(c.dt, c.gt) if c.dt and c.gt else np.zeros((len(c.dt), len(c.gt)))
            return
        # true 3D IoU: one kernel call per IMAGE (all its detections x all its ground truth), sliced per category
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        per_image = defaultdict(list)
        for (img, cat), c in self.cells.items():
            if c.dt and c.gt:
                per_image[img].append(c)
            else:
                c.iou = np.zeros((len(c.dt), len(c.gt)))
        for img, cs in per_image.items():
            bd = torch.tensor(np.asarray([d["bbox3D"] for c in cs for d in c.dt], np.float32), device=dev)
            bg = torch.tensor(np.nan_to_num(np.asarray([g["bbox3D"] for c in cs for g in c.gt], np.float32), nan=0.0, posinf=0.0, neginf=0.0),
                              device=dev)
            full = np.nan_to_num(box3d_overlap(bd, bg).cpu().numpy().astype(np.float64))
            r = q = 0
            for c in cs:
                c.iou = full[r:r + len(c.dt), q:q + len(c.gt)]
                r, q = r + len(c.dt), q + len(c.gt)

    # ---- greedy matching of one cell for one range, all IoU thresholds at once ----------------------------------------------
    @staticmethod
    def _match(iou: np.ndarray, gt_ign: np.ndarray, gt_crowd: np.ndarray, thrs: np.ndarray):
        """``iou`` [D,G] with the ground truth already ordered not-ignored first. Per threshold, detections pick in score
        order the free ground truth of highest IoU >= threshold - a later one on equal IoU - among the not-ignored ones, and
        only when none qualifies among the ignored ones; crowd ground truth never fills up. Returns the picked index [T,D]
        (-1: none) and whether the pick is an ignored ground truth [T,D]."""
        T, (D, G) = len(thrs), iou.shape
        pick = -np.ones((T, D), dtype=np.int64)
        if D == 0 or G == 0:
            return pick, np.zeros((T, D), dtype=bool)
        floor = np.minimum(thrs, 1 - 1e-10)
        taken = np.zeros((T, G), dtype=bool)
        rows = np.arange(T)
        keep_open = gt_crowd[None, :]
        groups = (~gt_ign[None, :], gt_ign[None, :])
        for d in range(D):
            free = ~taken | keep_open
            chosen = -np.ones(T, dtype=np.int64)
            for grp in groups:
                v = np.where(free & grp, iou[d][None, :], -1.0)
                last_best = G - 1 - np.argmax(v[:, ::-1], axis=1)
                ok = (v[rows, last_best] >= floor) & (chosen < 0)
                chosen = np.where(ok, last_best, chosen)
            hit = chosen >= 0
            taken[rows[hit], chosen[hit]] = True
            pick[:, d] = chosen
        return pick, np.where(pick >= 0, gt_ign[np.clip(pick, 0, None)], False)

    def _evaluate_cell(self, c: _Cell, rng):
        lo, hi = rng
        gt_ign = c.gt_flag | (c.gt_rng < lo) | (c.gt_rng > hi)
        order = np.argsort(gt_ign, kind="mergesort")                           # not-ignored first, file order within
        pick, on_ignored = self._match(c.iou[:, order] if c.iou.size else c.iou.reshape(len(c.dt), len(c.gt)), gt_ign[order],
                                       c.gt_crowd[order], self.params.iouThrs)
        matched = pick >= 0
        outside = (c.dt_rng < lo) | (c.dt_rng > hi)
        return {"score": c.score, "matched": matched, "ignored": on_ignored | (~matched & outside[None, :]),
                "n_gt": int(np.count_nonzero(~gt_ign)), "gt_order": order, "pick": pick}

    def evaluate(self):
        p = self.params
        self._build_cells()
        self.per_cell = {(key, a): self._evaluate_cell(c, rng) for key, c in self.cells.items() for a, rng in enumerate(p.areaRng)}
        if self.mode == "3D":
            self._collect_nhd()

    # ---- precision / recall tables ----------------------------------------------------------------------------------------
    def accumulate(self):
        """precision[T,R,K,A,M], recall[T,K,A,M], scores[T,R,K,A,M] with -1 where a (category, range) has no countable ground
        truth - the layout of COCOeval.accumulate (reference :1547-1688), which the per-category tables read."""
        p = self.params
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(p.catIds), len(p.areaRng), len(p.maxDets)
        precision, recall, scores = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M)), -np.ones((T, R, K, A, M))
        cells_of_cat = defaultdict(list)
        for (img, cat) in self.cells:
            cells_of_cat[cat].append(img)
        img_rank = {img: i for i, img in enumerate(p.imgIds)}
        tiny = np.spacing(1)
        for k, cat in enumerate(p.catIds):
            imgs = sorted(cells_of_cat.get(cat, ()), key=img_rank.__getitem__)     # concatenation order decides ties between images
            if not imgs:
                continue
            for a in range(A):
                res = [self.per_cell[(img, cat), a] for img in imgs]
                n_gt = sum(r["n_gt"] for r in res)
                if n_gt == 0:
                    continue
                for m, cap in enumerate(p.maxDets):
                    sc = np.concatenate([r["score"][:cap] for r in res])
                    order = np.argsort(-sc, kind="mergesort")
                    sc = sc[order]
                    hit = np.concatenate([r["matched"][:, :cap] for r in res], axis=1)[:, order]
                    ign = np.concatenate([r["ignored"][:, :cap] for r in res], axis=1)[:, order]
                    tp = np.cumsum(hit & ~ign, axis=1, dtype=np.float64)
                    fp = np.cumsum(~hit & ~ign, axis=1, dtype=np.float64)
                    nd = sc.shape[0]
                    recall[:, k, a, m] = tp[:, -1] / n_gt if nd else 0.0
                    prec_tab, score_tab = np.zeros((T, R)), np.zeros((T, R))
                    if nd:
                        rc = tp / n_gt
                        pr = tp / (fp + tp + tiny)
                        envelope = np.maximum.accumulate(pr[:, ::-1], axis=1)[:, ::-1]      # best precision at this recall or beyond
                        for t in range(T):
                            at = np.searchsorted(rc[t], p.recThrs, side="left")
                            reach = at < nd
                            prec_tab[t, reach] = envelope[t, at[reach]]
                            score_tab[t, reach] = sc[at[reach]]
                    precision[:, :, k, a, m] = prec_tab
                    scores[:, :, k, a, m] = score_tab
        self.eval = {"params": p, "counts": [T, R, K, A, M], "precision": precision, "recall": recall, "scores": scores}
        if self.mode == "3D":
            acc = {key: [pair[key] for pair in self.nhd_pairs] for key in ("overall",) + _nhd.COMPONENTS}
            self.eval["nhd_accumulators"] = acc
            self.eval["average_nhd"] = {key: (float(np.mean(v)) if v else float("nan")) for key, v in acc.items()}

    # ---- disentangled NHD over IoU-matched pairs (reference :2342-2484) -----------------------------------------------------
    def _collect_nhd(self):
        """A detection is paired with its highest-IoU ground truth of the cell (the first one on ties, IoU > 0) when that IoU
        reaches ``nhd_iou_threshold``. The reference evaluates the same pairs once per depth range and averages over all of
        them, which leaves the mean where it is; here each pair is scored once. (Its IoU lookup reads the matrix in
        un-reordered ground-truth order while walking the reordered list, :2363-2392; the pairing here follows the intent.)"""
        need_dt, need_gt = ("center_cam", "dimensions", "pose", "depth"), ("center_cam", "dimensions", "R_cam", "depth")
        self.nhd_pairs = []
        for key in sorted(self.cells, key=lambda k: (str(k[1]), str(k[0]))):
            c = self.cells[key]
            if not c.dt or not c.gt or c.iou.size == 0:
                continue
            best = np.argmax(c.iou, axis=1)
            for di, gi in enumerate(best):
                v = c.iou[di, gi]
                d, g = c.dt[di], c.gt[gi]
                if not (v > 0 and v >= self.nhd_iou_threshold) or any(f not in d for f in need_dt) or any(f not in g for f in need_gt):
                    continue
                pred = {"xy": d["center_cam"][:2], "z": d["depth"], "dimensions": d["dimensions"], "pose": d["pose"]}
                true = {"xy": g["center_cam"][:2], "z": g["depth"], "dimensions": g["dimensions"], "pose": g["R_cam"]}
                try:
                    self.nhd_pairs.append(_nhd.disentangled_nhd(pred, true))
                except Exception:                                               # a degenerate box: the reference logs and moves on (:2424-2426)
                    continue

    # ---- summarize (reference :2072-2224) ----------------------------------------------------------------------------------
    def _summarize(self, ap=1, iouThr=None, areaRng="all", maxDets=100):
        p = self.params
        aind = [i for i, lbl in enumerate(p.areaRngLbl) if lbl == areaRng]
        mind = [i for i, m in enumerate(p.maxDets) if m == maxDets]
        s = self.eval["precision"] if ap == 1 else self.eval["recall"]
        if iouThr is not None:
            t = np.where(np.isclose(iouThr, p.iouThrs.astype(float)))[0]
            s = s[t]
        s = s[:, :, :, aind, mind] if ap == 1 else s[:, :, aind, mind]
        return -1 if len(s[s > -1]) == 0 else float(np.mean(s[s > -1]))

    def summarize(self) -> Dict[str, float]:
        if self.mode == "2D":
            keys = [("AP", None, "all"), ("AP50", 0.5, "all"), ("AP75", 0.75, "all"), ("AP95", 0.95, "all"), ("APs", None, "small"),
                    ("APm", None, "medium"), ("APl", None, "large")]
        else:
            keys = [("AP", None, "all"), ("AP15", 0.15, "all"), ("AP25", 0.25, "all"), ("AP50", 0.50, "all"), ("APn", None, "near"),
                    ("APm", None, "medium"), ("APf", None, "far")]
        out = {k: self._summarize(1, thr, rng) * 100 for k, thr, rng in keys}
        for md in (1, 10, 100):                                                 # the reference's AR1 / AR10 / AR100 (:2190-2224)
            out[f"AR{md}"] = self._summarize(0, None, "all", md) * 100
        if self.mode == "3D" and "average_nhd" in self.eval:
            for comp, v in self.eval["average_nhd"].items():
                out["NHD" if comp == "overall" else f"NHD-{comp}"] = v
        self.stats = out
        return out

    def per_category_ap(self, class_names=None) -> Dict:
        """mean precision per category at area 'all', maxDets 100 (reference _derive_omni_results :1729-1819). ``class_names``:
        a dict id -> name, or a sequence indexed by the id."""
        prec = self.eval["precision"]
        out = {}
        for k, cid in enumerate(self.params.catIds):
            s = prec[:, :, k, 0, -1]
            s = s[s > -1]
            if isinstance(class_names, dict):
                name = class_names.get(cid, cid)
            else:
                name = class_names[cid] if class_names is not None and cid < len(class_names) else cid
            out[name] = float(np.mean(s) * 100) if s.size else float("nan")
        return out

    def per_category_ar(self, class_names=None) -> Dict:
        """mean recall per category over the IoU thresholds at area 'all', maxDets 100 (the AR-<category> entries of the
        reference's _derive_omni_results)."""
        rec = self.eval["recall"]
        out = {}
        for k, cid in enumerate(self.params.catIds):
            s = rec[:, k, 0, -1]
            s = s[s > -1]
            if isinstance(class_names, dict):
                name = class_names.get(cid, cid)
            else:
                name = class_names[cid] if class_names is not None and cid < len(class_names) else cid
            out[name] = float(np.mean(s) * 100) if s.size else float("nan")
        return out


def evaluate_omni3d(gts, dts: Sequence[Dict], device=None, only_2d: bool = False, fork_compat_2d_iou: bool = False,
                    category_map=None, passthrough_dataset_ids: bool = False) -> Dict:
    """AP2D and AP3D dictionaries for one dataset (reference _evaluate_predictions_on_omni :1255-1391 without the file plumbing).

    ``gts``: an ``Omni3DGroundTruth`` (images and categories evaluated = the dataset's, as the reference; detections on unknown
    images or categories are dropped :1319-1334) or a plain list of ground-truth dicts. ``category_map`` (``CategoryMap``): the
    detections carry the model's contiguous class index and are un-mapped to dataset ids first (:1029-1093)."""
    from .omni3d_gt import Omni3DGroundTruth, ground_truth_records
    img_ids = cat_ids = names = None
    if isinstance(gts, Omni3DGroundTruth):
        img_ids, cat_ids = list(gts.image_ids), list(gts.category_ids)
        names = dict(zip(gts.category_ids, gts.category_names))
        gts = ground_truth_records(gts)
    if category_map is not None:
        dts = category_map.detections_to_dataset_ids(dts, passthrough_dataset_ids)
    if img_ids is not None:
        known_i, known_c = set(img_ids), set(cat_ids)
        dts = [d for d in dts if d["image_id"] in known_i and d["category_id"] in known_c]
    res = {}
    e2 = Omni3Deval(gts, dts, "2D", img_ids=img_ids, cat_ids=cat_ids)
    e2.evaluate(); e2.accumulate()
    res["bbox_2D"] = e2.summarize()
    if names is not None:
        res["bbox_2D_per_category"] = e2.per_category_ap(names)
        res["bbox_2D_per_category_AR"] = e2.per_category_ar(names)
    if not only_2d:
        d3 = [d for d in dts if "bbox3D" in d]
        e3 = Omni3Deval(gts, d3, "3D", device=device, fork_compat_2d_iou=fork_compat_2d_iou, img_ids=img_ids, cat_ids=cat_ids)
        e3.evaluate(); e3.accumulate()
        res["bbox_3D"] = e3.summarize()
        if names is not None:
            res["bbox_3D_per_category"] = e3.per_category_ap(names)
            res["bbox_3D_per_category_AR"] = e3.per_category_ar(names)
    return res


def omni3d_json_to_gt(dataset_json: Dict, filter_settings: Optional[Dict] = None) -> List[Dict]:
    """Ground-truth records of an Omni3D annotation dict under the reference's filter settings (defaults: datasets.py:55-66);
    see ``omni3d_gt.Omni3DGroundTruth`` for the rules."""
    from .omni3d_gt import Omni3DGroundTruth, filter_settings_from_cfg, ground_truth_records
    return ground_truth_records(Omni3DGroundTruth(dataset_json, filter_settings if filter_settings is not None else filter_settings_from_cfg(None)))


# Omni3D's indoor / outdoor category groups (cubercnn/data/builtin.py:15-20: facts of the benchmark)
OMNI3D_OUT = frozenset("pedestrian car cyclist van truck bus trailer motorcycle bicycle barrier".split() + ["traffic cone"])
OMNI3D_IN = frozenset(["chair", "table", "cabinet", "lamp", "books", "sofa", "picture", "window", "pillow", "door", "blinds", "sink", "shelves",
                       "television", "shoes", "cup", "bottle", "bookcase", "laptop", "desk", "floor mat", "mirror", "counter", "bicycle", "toilet", "bed",
                       "refrigerator", "box", "oven", "clothes", "towel", "night stand", "stove", "machine", "stationery", "bathtub", "curtain", "bin"])
OMNI3D_ALL = OMNI3D_OUT | OMNI3D_IN | frozenset(["camera", "cereal box"])      # the benchmark's 50 categories (builtin.py:12-14)


def collective_summary(results: Dict) -> Dict:
    """The cross-dataset numbers of the reference's ``Omni3DEvaluationHelper.summarize_all`` (:427-620) from ONE ``evaluate_omni3d`` result
    over the concatenation of the datasets (re-accumulating the cached per-image results of every dataset, as the reference does, is the
    same computation: a cell's matching depends on its own image and category only). ``<Concat>`` averages the per-category AP / AR
    over the categories that have ground truth; ``Omni3D_Out`` / ``Omni3D_In`` / ``Omni3D`` over the benchmark's outdoor / indoor / all 50
    categories, NaN unless every one of them was evaluated."""
    def mean_over(table, cats):
        vals = [table[c] for c in cats]
        return float(np.mean(vals)) if vals and not any(np.isnan(v) for v in vals) else float("nan")
    ap2, ar2 = results["bbox_2D_per_category"], results["bbox_2D_per_category_AR"]
    ap3, ar3 = results.get("bbox_3D_per_category"), results.get("bbox_3D_per_category_AR")
    have = {c for c, v in ap2.items() if not np.isnan(v)}
    out = {"<Concat>": {"AP2D": mean_over(ap2, sorted(have)), "AR2D": mean_over(ar2, sorted(have)),
                        "AP3D": mean_over(ap3, sorted(have)) if ap3 else float("nan"), "AR3D": mean_over(ar3, sorted(have)) if ar3 else float("nan")}}
    if "bbox_3D" in results:
        out["<Concat>"].update({"AP3D@15": results["bbox_3D"]["AP15"], "AP3D@25": results["bbox_3D"]["AP25"], "AP3D@50": results["bbox_3D"]["AP50"],
                                "AP3D-N": results["bbox_3D"]["APn"], "AP3D-M": results["bbox_3D"]["APm"], "AP3D-F": results["bbox_3D"]["APf"]})
        for k in ("NHD", "NHD-xy", "NHD-z", "NHD-dimensions", "NHD-pose"):
            if k in results["bbox_3D"]:
                out["<Concat>"][k] = results["bbox_3D"][k]
    for label, group in (("Omni3D_Out", OMNI3D_OUT), ("Omni3D_In", OMNI3D_IN), ("Omni3D", OMNI3D_ALL)):
        full = group <= have
        out[label] = {"AP2D": mean_over(ap2, sorted(group)) if full else float("nan"), "AR2D": mean_over(ar2, sorted(group)) if full else float("nan"),
                      "AP3D": mean_over(ap3, sorted(group)) if full and ap3 else float("nan"),
                      "AR3D": mean_over(ar3, sorted(group)) if full and ar3 else float("nan")}
    return out
