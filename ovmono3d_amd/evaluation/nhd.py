"""Normalised Hungarian Distance between cuboids, whole and per component.

Behaviour of the reference's ``calculate_nhd`` / ``disentangled_nhd`` (cubercnn/evaluation/omni3d_evaluation.py:2227-2290):
the 8x8 corner-distance matrix is solved as an assignment problem (scipy ``linear_sum_assignment``), the matched distances
are summed and divided by the diagonal of the ground truth's axis-aligned extent. The disentangled variant scores one
component of the prediction (xy, z, dimensions, pose) at a time with the other three replaced by the ground truth's.
Corners follow ``get_cuboid_verts_faces`` (cubercnn/util/math_util.py:116-190): dimensions are (W, H, L) with L along the
box's x axis, H along y, W along z, in float32. The corner ORDER does not matter to the assignment.
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
from scipy.optimize import linear_sum_assignment

COMPONENTS = ("xy", "z", "dimensions", "pose")
_SIGNS = np.array([[sx, sy, sz] for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)], dtype=np.float32)     # [8,3]


def cuboid_corners(xy: Sequence[float], z: float, dimensions: Sequence[float], pose) -> np.ndarray:
    """[8,3] float32 corners of the cuboid centred at (xy, z) with (W, H, L) ``dimensions`` and rotation ``pose``."""
    w, h, l = (np.float32(v) for v in dimensions)
    half = np.array([l, h, w], dtype=np.float32) / np.float32(2)
    R = np.asarray(pose, dtype=np.float32).reshape(3, 3)
    c = np.array([xy[0], xy[1], z], dtype=np.float32)
    return (_SIGNS * half) @ R.T + c


def hungarian_distance(pred_corners: np.ndarray, gt_corners: np.ndarray) -> float:
    """:2227-2246 - sum of optimally assigned corner distances over the ground truth's bounding diagonal."""
    cost = np.linalg.norm(pred_corners[:, None, :] - gt_corners[None, :, :], axis=2)
    rows, cols = linear_sum_assignment(cost)
    return float(cost[rows, cols].sum() / np.linalg.norm(gt_corners.max(axis=0) - gt_corners.min(axis=0)))


def disentangled_nhd(pred: Dict, gt: Dict, components: Sequence[str] = COMPONENTS) -> Dict[str, float]:
    """``pred`` / ``gt``: {"xy": (x, y), "z": depth, "dimensions": (W, H, L), "pose": 3x3}. Returns ``overall`` plus one entry
    per component (:2249-2290)."""
    gt_corners = cuboid_corners(gt["xy"], gt["z"], gt["dimensions"], gt["pose"])
    out = {"overall": hungarian_distance(cuboid_corners(pred["xy"], pred["z"], pred["dimensions"], pred["pose"]), gt_corners)}
    for comp in components:
        mixed = {c: (pred[c] if c == comp else gt[c]) for c in components}
        out[comp] = hungarian_distance(cuboid_corners(mixed["xy"], mixed["z"], mixed["dimensions"], mixed["pose"]), gt_corners)
    return out
