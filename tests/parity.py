"""Detection-level parity report between the HIP path's Instances and the CPU oracle's result dict for one image.

With the full-size GroundingDINO network the two routes can order near-tied proposals differently (two-stage top-900 and NMS
work on scores that agree to ~1e-4), so detections are paired by their 2D boxes before the fields are compared; with given
boxes (oracle-2D branch) the pairing is the identity. Used by bench.py's `parity` object and by the headline-size GPU tests."""
from __future__ import annotations

from typing import Dict

import torch

FIELDS = ("pred_boxes", "scores", "pred_bbox3D", "pred_center_cam", "pred_center_2D", "pred_dimensions", "pred_pose")

# "1e-3 rel" element by element needs a floor under which "relative" has no meaning for an element of the field (a rotation entry
# that is ~0, the x of an object on the optical axis): |a_i - b_i| / max(|b_i|, floor). Floors in the field's own unit.
# |p6| / min(|a1|, |a2 - (b1.a2) b1|) above which a detection's 6-D -> R map counts as near-degenerate: the shorter Gram-Schmidt leg is
# less than a fifth of the 6-D vector. Measured on the synthetic checkpoint: the raw 6-D output of the two routes differs by ~1.2e-4 of
# its length on every detection, so pose entries differ by ~1.2e-4 x amplification - 3e-4 at the median amplification of 2.35, 6e-4 at 5,
# 1.03e-3 at 8.6 (COCO example, 1 of 92 detections), 1.3e-3 at 49 (synthetic headline image, 1 of 532)
POSE_AMP_FLOOR = 5.0

ELEM_FLOOR = {"pred_boxes": 1.0,          # pixels (original resolution)
              "scores": 1e-2,
              "pred_bbox3D": 0.1,         # metres
              "pred_center_cam": 0.1,     # metres
              "pred_center_2D": 1.0,      # pixels
              "pred_dimensions": 0.05,    # metres
              "pred_pose": 0.1}           # entries of a rotation matrix


def elem_rel_err(a: torch.Tensor, b: torch.Tensor, floor: float) -> float:
    """max_i |a_i - b_i| / max(|b_i|, floor): the element-wise relative error with an absolute floor."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if b.numel() == 0:
        return 0.0
    return float(((a - b).abs() / b.abs().clamp_min(floor)).max())


def geodesic(Ra: torch.Tensor, Rb: torch.Tensor) -> torch.Tensor:
    """Rotation angle (rad) of Ra^T Rb per detection; for small angles ||Ra - Rb||_F / sqrt(2) (no acos cancellation)."""
    return (Ra.double() - Rb.double()).flatten(1).norm(dim=1) / (2.0 ** 0.5)


def pose_conditioning(p6: torch.Tensor):
    """rotation_6d_to_matrix (cube_head.py:177) divides by n1 = |a1| and n2 = |a2 - (b1.a2) b1|: a perturbation d of the head's raw
    6-D output turns the frame by ~|d| / n1 and ~|d| / n2. Returns (n1, n2, amplification of a perturbation measured relative to the
    vector's own length |p6|: |p6| / min(n1, n2))."""
    p6 = p6.double()
    a1, a2 = p6[:, :3], p6[:, 3:]
    n1 = a1.norm(dim=1)
    b1 = a1 / n1.clamp_min(1e-30)[:, None]
    n2 = (a2 - (b1 * a2).sum(1, keepdim=True) * b1).norm(dim=1)
    return n1, n2, p6.norm(dim=1) / torch.minimum(n1, n2).clamp_min(1e-30)


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
    rep["max_elem_rel_err"] = {}
    for f in FIELDS:
        a, b = _field(inst, f)[gi], ref[f].double()[ok]
        rep["max_rel_err"][f] = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        rep["max_elem_rel_err"][f] = elem_rel_err(a, b, ELEM_FLOOR[f])
    # ---- pose, detection by detection: which ones exceed 1e-3, by how much their 2D boxes differ, and how the 6-D -> R map is
    # conditioned there (the oracle's raw 6-D head output rides along as the diagnostic field `_pose6d`)
    pa, pb = _field(inst, "pred_pose")[gi], ref["pred_pose"].double()[ok]
    perr = (pa - pb).abs().flatten(1).amax(1)
    ang = geodesic(pa, pb)
    box_px = (gb[gi] - rb[ok]).abs().amax(1)
    pose = {"n_over_1e-3": int((perr > 1e-3).sum()), "max_entry_err": float(perr.max()), "max_geodesic_rad": float(ang.max()),
            "max_box_delta_px": float(box_px.max())}
    if "_pose6d" in ref:
        n1, n2, amp = pose_conditioning(ref["_pose6d"][ok])
        pose["amplification_median"] = float(amp.median())
        # Near-degenerate Gram-Schmidt inputs (a2 almost parallel to a1, or a tiny a1): amplification above POSE_AMP_FLOOR. There
        # "1e-3 on the matrix entries" is not a property of the arithmetic but of the input, so they are held to the ANGLE a
        # 1e-3-relative perturbation of the raw 6-D vector causes at that conditioning (1e-3 x amplification rad); every other
        # detection is held to 1e-3 on the entries like any float field.
        ill = amp > POSE_AMP_FLOOR
        pose["amplification_floor"] = POSE_AMP_FLOOR
        pose["n_ill_conditioned"] = int(ill.sum())
        pose["well_conditioned_max_entry_err"] = float(perr[~ill].max()) if bool((~ill).any()) else 0.0
        pose["ill_conditioned_max_geodesic_over_1e-3xamp"] = float((ang[ill] / (1e-3 * amp[ill])).max()) if bool(ill.any()) else 0.0
        worst = torch.argsort(perr, descending=True)[:8].tolist()
        pose["worst"] = [{"oracle_idx": int(torch.nonzero(ok)[k]), "entry_err": float(perr[k]), "geodesic_rad": float(ang[k]),
                          "box_delta_px": float(box_px[k]), "n1": float(n1[k]), "n2": float(n2[k]), "amplification": float(amp[k])}
                         for k in worst if perr[k] > 1e-3]
    rep["pose"] = pose
    return rep


def parity_ok(rep: Dict, tol: float = 1e-3, pose_by_conditioning: bool = False) -> bool:
    """Every paired detection, every float field within `tol` (scale-relative, tests/common.py:rel_err), ids exact, nothing unpaired.
    pose_by_conditioning (the end-to-end leg behind a proposal stage only - the two routes hand the cube head 2D boxes that differ in
    the last bits): pred_pose is held to `tol` on the matrix entries for every detection whose 6-D -> R map is well conditioned
    (amplification <= POSE_AMP_FLOOR) and, for the near-degenerate ones, to the ANGLE a tol-relative perturbation of the head's raw
    6-D output causes at that detection's own conditioning - not to a blanket looser number. Measured on the headline configuration
    (profiles/r03): 1 of 532 detections exceeds 1e-3 (1.3e-3); its a2 - (b1.a2) b1 has length 0.0030 against |a1| = 0.128
    (amplification 49), its 2D boxes differ by 2.6e-4 px between the routes, its angle is 0.14 of the conditioned bound."""
    if rep["unmatched_oracle"] or rep["unmatched_hip"] or rep["class_id_mismatches"]:
        return False
    for k, v in rep["max_rel_err"].items():
        if k == "pred_pose" and pose_by_conditioning and "well_conditioned_max_entry_err" in rep.get("pose", {}):
            if rep["pose"]["well_conditioned_max_entry_err"] > tol or rep["pose"]["ill_conditioned_max_geodesic_over_1e-3xamp"] * 1e-3 > tol:
                return False
        elif v > tol:
            return False
    return True
