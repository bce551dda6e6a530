"""Detection-level parity report between the HIP path's Instances and the CPU oracle's result dict for one image.

With the full-size GroundingDINO network the two routes can order near-tied proposals differently (two-stage top-900 and NMS
work on scores that agree to ~1e-4), so detections are paired by their 2D boxes before the fields are compared; with given
boxes (oracle-2D branch) the pairing is the identity. Used by bench.py's `parity` object and by the headline-size GPU tests."""
from __future__ import annotations

from typing import Dict

import torch

FIELDS = ("pred_boxes", "scores", "pred_bbox3D", "pred_center_cam", "pred_center_2D", "pred_dimensions", "pred_pose")


def _field(inst, f):
    v = inst.get(f)
    return (v.tensor if hasattr(v, "tensor") else v).detach().cpu().double()


def parity_report(inst, ref: Dict[str, torch.Tensor], box_tol: float = 2e-3) -> Dict:
    """inst: Instances from the HIP path; ref: oracle dict (same field names). Pairs every oracle detection with the HIP
    detection of nearest 2D box (max abs corner difference relative to the box scale, must be <= box_tol); reports the number of
    unpaired detections on either side, class-id mismatches among pairs, per-field scale-relative error max|a-b| / max|b|
    (the metric of tests/common.py:rel_err) and whether the paired order is the identity."""
    n_ref = int(ref["scores"].shape[0])
    n_got = len(inst) if inst.get_fields() else 0
    rep = {"n_det": n_got, "n_det_oracle": n_ref, "matched": 0, "unmatched_oracle": n_ref, "unmatched_hip": n_got,
           "class_id_mismatches": 0, "same_order": n_ref == n_got, "max_rel_err": {}}
    if n_ref == 0 or n_got == 0:
        return rep
    gb, rb = _field(inst, "pred_boxes"), ref["pred_boxes"].double()
    scale = rb.abs().max().clamp_min(1.0)
    d = (gb[None, :, :] - rb[:, None, :]).abs().amax(-1) / scale            # [n_ref, n_got]
    # the two routes may keep different members of a near-duplicate cluster apart; greedy nearest, each HIP detection once
    order = torch.argsort(d.min(1)[0])
    used = torch.zeros(n_got, dtype=torch.bool)
    pair = torch.full((n_ref,), -1, dtype=torch.int64)
    for i in order.tolist():
        row = d[i].clone()
        row[used] = float("inf")
        j = int(torch.argmin(row))
        if row[j] <= box_tol:
            pair[i] = j
            used[j] = True
    ok = pair >= 0
    m = int(ok.sum())
    rep.update(matched=m, unmatched_oracle=n_ref - m, unmatched_hip=n_got - m,
               same_order=bool(n_ref == n_got and torch.equal(pair, torch.arange(n_ref))))
    if m == 0:
        return rep
    gi = pair[ok]
    gc = inst.pred_classes.detach().cpu().to(torch.int64)[gi]
    rep["class_id_mismatches"] = int((gc != ref["pred_classes"].to(torch.int64)[ok]).sum())
    for f in FIELDS:
        a, b = _field(inst, f)[gi], ref[f].double()[ok]
        rep["max_rel_err"][f] = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    return rep


def parity_ok(rep: Dict, tol: float = 1e-3) -> bool:
    return (rep["unmatched_oracle"] == 0 and rep["unmatched_hip"] == 0 and rep["class_id_mismatches"] == 0
            and all(v <= tol for v in rep["max_rel_err"].values()))
