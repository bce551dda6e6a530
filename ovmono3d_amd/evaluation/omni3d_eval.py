"""Omni3D AP evaluation with true 3D IoU ("next" row 1 of SURVEY.md 8f).

Upstream Omni3D semantics of the reference's ``Omni3Deval`` (cubercnn/evaluation/omni3d_evaluation.py):
``Omni3DParams`` :1394-1461 (2D: IoU 0.5:0.05:0.95, area ranges; 3D: IoU 0.05:0.05:0.50, depth ranges [0,10,35,1e5]),
``_prepare`` :1515-1545 (ignore2D / ignore3D flags), COCO matching per (image, category, range) and ``accumulate``
:1547-1688 (101 recall points, mergesort on scores), ``summarize`` :2072-2224 (AP, AP15 / AP25 / AP50, near / medium /
far). The 3D IoU is ``box3d_overlap`` (:109-169) on the HIP kernel ``ovm_box3d_iou``.

The reference FORK lost its ``computeIoU`` override, so its "3D" AP is really computed with pycocotools' 2D IoU
(SURVEY.md 0.5); ``fork_compat_2d_iou=True`` reproduces that behaviour, the default is upstream's.
pycocotools is not a dependency: the matching / accumulation below is the published COCOeval algorithm.
"""
from __future__ import annotations

import ctypes as C
from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import lib as _lib


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


class Omni3Deval:
    """COCO-style AP over Omni3D annotations. ``gts`` / ``dts`` are lists of dicts:
    gt: image_id, category_id, bbox [x,y,w,h], bbox3D [8][3], depth, optional area, ignore2D, ignore3D, iscrowd
    dt: image_id, category_id, bbox, score, and for 3D bbox3D + depth (the records of ``instances_to_coco_json``)."""

    def __init__(self, gts: Sequence[Dict], dts: Sequence[Dict], mode: str = "3D", device: Optional[torch.device] = None,
                 fork_compat_2d_iou: bool = False):
        self.mode = mode
        self.params = Omni3DParams(mode)
        self.device = device
        self.fork_compat_2d_iou = fork_compat_2d_iou
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
        self.params.imgIds = sorted({g["image_id"] for g in self._gts_all} | {d["image_id"] for d in self._dts_all})
        self.params.catIds = sorted({g["category_id"] for g in self._gts_all} | {d["category_id"] for d in self._dts_all})
        self.evalImgs, self.eval, self.ious = [], {}, {}

    # ---- reference _prepare :1515-1545 -------------------------------------------------------------------------------
    def _prepare(self):
        flag = "ignore2D" if self.mode == "2D" else "ignore3D"
        self._gts, self._dts = defaultdict(list), defaultdict(list)
        for g in self._gts_all:
            g[flag] = g[flag] if flag in g else 0
            self._gts[g["image_id"], g["category_id"]].append(g)
        for d in self._dts_all:
            self._dts[d["image_id"], d["category_id"]].append(d)

    def _range_value(self, ann):
        return ann["area"] if self.mode == "2D" else ann["depth"]

    def computeIoU(self, imgId, catId):
        gt, dt = self._gts[imgId, catId], self._dts[imgId, catId]
        if len(gt) == 0 or len(dt) == 0:
            return []
        inds = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in inds]
        if len(dt) > self.params.maxDets[-1]:
            dt = dt[0:self.params.maxDets[-1]]
        if self.mode == "2D" or self.fork_compat_2d_iou:
            return iou2d_xywh(np.array([d["bbox"] for d in dt]), np.array([g["bbox"] for g in gt]))
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        bd = torch.tensor(np.asarray([d["bbox3D"] for d in dt], np.float32), device=dev)
        bg = torch.tensor(np.asarray([g["bbox3D"] for g in gt], np.float32), device=dev)
        return box3d_overlap(bd, bg).cpu().numpy().astype(np.float64)

    # ---- COCOeval.evaluateImg with the Omni3D ignore / range rules --------------------------------------------------------
    def evaluateImg(self, imgId, catId, aRng, maxDet):
        p = self.params
        gt, dt = self._gts[imgId, catId], self._dts[imgId, catId]
        if len(gt) == 0 and len(dt) == 0:
            return None
        flag = "ignore2D" if self.mode == "2D" else "ignore3D"
        for g in gt:
            v = self._range_value(g)
            g["_ignore"] = 1 if (g[flag] or v < aRng[0] or v > aRng[1]) else 0
        gtind = np.argsort([g["_ignore"] for g in gt], kind="mergesort")
        gt = [gt[i] for i in gtind]
        dtind = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in dtind[0:maxDet]]
        iscrowd = [int(o["iscrowd"]) for o in gt]
        ious = self.ious[imgId, catId][:, gtind] if len(self.ious[imgId, catId]) > 0 else self.ious[imgId, catId]
        T, G, D = len(p.iouThrs), len(gt), len(dt)
        gtm, dtm = np.zeros((T, G)), np.zeros((T, D))
        gtIg = np.array([g["_ignore"] for g in gt])
        dtIg = np.zeros((T, D))
        if len(ious) != 0:
            for tind, t in enumerate(p.iouThrs):
                for dind, d in enumerate(dt):
                    iou = min([t, 1 - 1e-10])
                    m = -1
                    for gind, g in enumerate(gt):
                        if gtm[tind, gind] > 0 and not iscrowd[gind]:
                            continue
                        if m > -1 and gtIg[m] == 0 and gtIg[gind] == 1:
                            break
                        if ious[dind, gind] < iou:
                            continue
                        iou = ious[dind, gind]
                        m = gind
                    if m == -1:
                        continue
                    dtIg[tind, dind] = gtIg[m]
                    dtm[tind, dind] = gt[m]["id"]
                    gtm[tind, m] = d["id"]
        a = np.array([self._range_value(d) < aRng[0] or self._range_value(d) > aRng[1] for d in dt]).reshape((1, len(dt)))
        dtIg = np.logical_or(dtIg, np.logical_and(dtm == 0, np.repeat(a, T, 0)))
        return {"image_id": imgId, "category_id": catId, "aRng": aRng, "maxDet": maxDet, "dtIds": [d["id"] for d in dt],
                "gtIds": [g["id"] for g in gt], "dtMatches": dtm, "gtMatches": gtm, "dtScores": [d["score"] for d in dt],
                "gtIgnore": gtIg, "dtIgnore": dtIg}

    def evaluate(self):
        p = self.params
        self._prepare()
        self.ious = {(i, c): self.computeIoU(i, c) for i in p.imgIds for c in p.catIds}
        maxDet = p.maxDets[-1]
        self.evalImgs = [self.evaluateImg(i, c, a, maxDet) for c in p.catIds for a in p.areaRng for i in p.imgIds]

    # ---- COCOeval.accumulate (reference :1547-1688) -----------------------------------------------------------------------
    def accumulate(self):
        p = self.params
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(p.catIds), len(p.areaRng), len(p.maxDets)
        precision, recall, scores = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M)), -np.ones((T, R, K, A, M))
        I0, A0 = len(p.imgIds), len(p.areaRng)
        for k in range(K):
            Nk = k * A0 * I0
            for a in range(A):
                Na = a * I0
                for m, maxDet in enumerate(p.maxDets):
                    E = [self.evalImgs[Nk + Na + i] for i in range(I0)]
                    E = [e for e in E if e is not None]
                    if len(E) == 0:
                        continue
                    dtScores = np.concatenate([e["dtScores"][0:maxDet] for e in E])
                    inds = np.argsort(-dtScores, kind="mergesort")
                    dtScoresSorted = dtScores[inds]
                    dtm = np.concatenate([e["dtMatches"][:, 0:maxDet] for e in E], axis=1)[:, inds]
                    dtIg = np.concatenate([e["dtIgnore"][:, 0:maxDet] for e in E], axis=1)[:, inds]
                    gtIg = np.concatenate([e["gtIgnore"] for e in E])
                    npig = np.count_nonzero(gtIg == 0)
                    if npig == 0:
                        continue
                    tps = np.logical_and(dtm, np.logical_not(dtIg))
                    fps = np.logical_and(np.logical_not(dtm), np.logical_not(dtIg))
                    tp_sum = np.cumsum(tps, axis=1).astype(dtype=float)
                    fp_sum = np.cumsum(fps, axis=1).astype(dtype=float)
                    for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
                        tp, fp = np.array(tp), np.array(fp)
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        q, ss = np.zeros((R,)), np.zeros((R,))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr, q = pr.tolist(), q.tolist()
                        for i in range(nd - 1, 0, -1):
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        inds_r = np.searchsorted(rc, p.recThrs, side="left")
                        try:
                            for ri, pi in enumerate(inds_r):
                                q[ri] = pr[pi]
                                ss[ri] = dtScoresSorted[pi]
                        except IndexError:
                            pass
                        precision[t, :, k, a, m] = np.array(q)
                        scores[t, :, k, a, m] = np.array(ss)
        self.eval = {"params": p, "counts": [T, R, K, A, M], "precision": precision, "recall": recall, "scores": scores}

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
            keys = [("AP", None, "all"), ("AP50", 0.5, "all"), ("AP75", 0.75, "all"), ("APs", None, "small"), ("APm", None, "medium"),
                    ("APl", None, "large")]
        else:
            keys = [("AP", None, "all"), ("AP15", 0.15, "all"), ("AP25", 0.25, "all"), ("AP50", 0.50, "all"), ("APn", None, "near"),
                    ("APm", None, "medium"), ("APf", None, "far")]
        out = {k: self._summarize(1, thr, rng) * 100 for k, thr, rng in keys}
        out["AR100"] = self._summarize(0, None, "all", 100) * 100
        self.stats = out
        return out

    def per_category_ap(self, class_names: Optional[Sequence[str]] = None) -> Dict:
        """mean precision per category at area 'all', maxDets 100 (reference _derive_omni_results :1729-1819)."""
        prec = self.eval["precision"]
        out = {}
        for k, cid in enumerate(self.params.catIds):
            s = prec[:, :, k, 0, -1]
            s = s[s > -1]
            name = class_names[cid] if class_names is not None and cid < len(class_names) else cid
            out[name] = float(np.mean(s) * 100) if s.size else float("nan")
        return out


def evaluate_omni3d(gts: Sequence[Dict], dts: Sequence[Dict], device=None, only_2d: bool = False, fork_compat_2d_iou: bool = False) -> Dict:
    """AP2D and AP3D dictionaries for one dataset (reference _evaluate_predictions_on_omni :1255-1391 without the file plumbing)."""
    res = {}
    e2 = Omni3Deval(gts, dts, "2D")
    e2.evaluate(); e2.accumulate()
    res["bbox_2D"] = e2.summarize()
    if not only_2d:
        d3 = [d for d in dts if "bbox3D" in d]
        e3 = Omni3Deval(gts, d3, "3D", device=device, fork_compat_2d_iou=fork_compat_2d_iou)
        e3.evaluate(); e3.accumulate()
        res["bbox_3D"] = e3.summarize()
    return res


def omni3d_json_to_gt(dataset_json: Dict) -> List[Dict]:
    """Ground-truth records from an Omni3D annotation file (the fields the reference's loader keeps, cubercnn/data/datasets.py:
    `bbox2D_proj` / `bbox2D_tight` / `bbox2D_trunc` xyxy, `bbox3D_cam` 8x3, `center_cam`, `behind_camera`). A simplification of the
    reference's `is_ignore` filter settings: an annotation is ignored in 3D when it is behind the camera or has no valid 3D box,
    in 2D when it has no valid 2D box."""
    out = []
    for a in dataset_json.get("annotations", []):
        box = None
        for k in ("bbox2D_proj", "bbox2D_tight", "bbox2D_trunc", "bbox"):
            b = a.get(k)
            if b is not None and len(b) == 4 and b[0] != -1:
                box = [float(b[0]), float(b[1]), float(b[2] - b[0]), float(b[3] - b[1])] if k != "bbox" else [float(v) for v in b]
                break
        c3 = a.get("bbox3D_cam")
        valid3 = c3 is not None and np.asarray(c3).shape == (8, 3) and not a.get("behind_camera", False)
        depth = float(a["center_cam"][2]) if a.get("center_cam") is not None else (float(np.mean(np.asarray(c3)[:, 2])) if valid3 else 0.0)
        out.append({"image_id": a["image_id"], "category_id": a["category_id"], "bbox": box if box is not None else [0.0, 0.0, 0.0, 0.0],
                    "bbox3D": c3 if valid3 else np.zeros((8, 3)).tolist(), "depth": depth, "ignore2D": int(box is None),
                    "ignore3D": int(not valid3), "iscrowd": 0})
    return out
