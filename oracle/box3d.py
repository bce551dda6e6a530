"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). Exact volume of the intersection of two oriented boxes in float64 via
scipy (half-space intersection + convex hull): an independent route to what pytorch3d's `_C.iou_box3d` computes behind the
reference's `box3d_overlap` (cubercnn/evaluation/omni3d_evaluation.py:109-169). pytorch3d is not installed here, so parity with
it is unpinned; the two are exact algorithms for the same geometric quantity."""
from __future__ import annotations

import numpy as np
from scipy.spatial import ConvexHull, HalfspaceIntersection, QhullError

_FACES = [[0, 1, 2, 3], [3, 2, 6, 7], [0, 1, 5, 4], [0, 3, 7, 4], [1, 2, 6, 5], [4, 5, 6, 7]]      # pytorch3d `_box_planes`


def _halfspaces(c: np.ndarray) -> np.ndarray:
    ctr = c.mean(0)
    hs = []
    for f in _FACES:
        v = c[f]
        n = np.cross(v[1] - v[0], v[3] - v[0])
        n = n / np.linalg.norm(n)
        fc = v.mean(0)
        if np.dot(n, fc - ctr) < 0:
            n = -n
        hs.append(np.append(n, -np.dot(n, fc)))               # n.x + b <= 0
    return np.array(hs)


def box_volume(c: np.ndarray) -> float:
    return float(ConvexHull(np.asarray(c, np.float64)).volume)


def intersection_volume(a: np.ndarray, b: np.ndarray) -> float:
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    hs = np.concatenate([_halfspaces(a), _halfspaces(b)], 0)
    # an interior point: Chebyshev centre by linear programming
    from scipy.optimize import linprog
    A = np.hstack([hs[:, :3], np.linalg.norm(hs[:, :3], axis=1, keepdims=True)])
    res = linprog(c=[0, 0, 0, -1], A_ub=A, b_ub=-hs[:, 3], bounds=[(None, None)] * 3 + [(0, None)])
    if not res.success or res.x[3] <= 1e-9:
        return 0.0
    try:
        pts = HalfspaceIntersection(hs, res.x[:3]).intersections
        return float(ConvexHull(pts).volume)
    except QhullError:
        return 0.0


def iou_matrix(dt: np.ndarray, gt: np.ndarray) -> np.ndarray:
    out = np.zeros((len(dt), len(gt)))
    vd, vg = [box_volume(d) for d in dt], [box_volume(g) for g in gt]
    for i, d in enumerate(dt):
        for j, g in enumerate(gt):
            v = intersection_volume(d, g)
            out[i, j] = v / (vd[i] + vg[j] - v)
    return out


def make_box(center, dims, R) -> np.ndarray:
    """8 corners in pytorch3d order for a box with half-extents dims/2 along the columns of R."""
    unit = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], np.float64) - 0.5
    return (unit * np.asarray(dims, np.float64)) @ np.asarray(R, np.float64).T + np.asarray(center, np.float64)
