"""AP evaluator ("next" row 1): the 3D IoU oracle against closed forms, and the COCO-style accumulation on cases with known AP.
The HIP IoU kernel is checked in the gpu-marked tests at the bottom."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from oracle import box3d as ob


def _rand_boxes(n, seed, spread=1.5):
    g = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        R = Rotation.from_rotvec(g.normal(size=3) * g.uniform(0, 1.5)).as_matrix()
        out.append(ob.make_box(g.normal(size=3) * spread + [0, 0, 10], g.uniform(0.5, 3.0, size=3), R))
    return np.array(out)


def test_oracle_iou_closed_forms():
    a = ob.make_box([0, 0, 5], [2, 2, 2], np.eye(3))
    assert abs(ob.box_volume(a) - 8.0) < 1e-9
    b = ob.make_box([1, 0, 5], [2, 2, 2], np.eye(3))                       # overlap 1 x 2 x 2
    assert abs(ob.intersection_volume(a, b) - 4.0) < 1e-9
    assert abs(ob.iou_matrix(a[None], b[None])[0, 0] - 4.0 / 12.0) < 1e-9
    Rz = Rotation.from_euler("z", 45, degrees=True).as_matrix()
    c = ob.make_box([0, 0, 5], [2, 2, 2], Rz)                              # square rotated 45 deg: octagon of area 8(sqrt2 - 1), height 2
    assert abs(ob.intersection_volume(a, c) - 2 * 8 * (2 ** 0.5 - 1)) < 1e-9
    assert ob.intersection_volume(a, ob.make_box([5, 0, 5], [2, 2, 2], np.eye(3))) == 0.0


def _ann(img, cat, box, depth=5.0, score=None, **kw):
    x, y, w, h = box
    c3 = ob.make_box([x + w / 2, y + h / 2, depth], [w, h, 1.0], np.eye(3)).tolist()
    d = {"image_id": img, "category_id": cat, "bbox": list(box), "bbox3D": c3, "depth": depth, **kw}
    if score is not None:
        d["score"] = score
    return d


def test_ap_known_values_2d_and_fork_compat():
    from ovmono3d_amd.evaluation.omni3d_eval import Omni3Deval, evaluate_omni3d
    gts = [_ann(1, 0, [10, 10, 50, 50]), _ann(1, 0, [100, 100, 40, 40]), _ann(2, 0, [20, 20, 60, 60]), _ann(2, 1, [5, 5, 30, 30])]
    perfect = [dict(g, score=0.9 - 0.1 * i) for i, g in enumerate(gts)]
    r = evaluate_omni3d(gts, perfect, only_2d=True)["bbox_2D"]
    assert abs(r["AP"] - 100.0) < 1e-9 and abs(r["AP50"] - 100.0) < 1e-9 and abs(r["AR100"] - 100.0) < 1e-9
    # one true positive ranked above one false positive, a second GT missed: precision 1 up to recall 0.5, then nothing
    dts = [_ann(1, 0, [10, 10, 50, 50], score=0.9), _ann(1, 0, [300, 300, 20, 20], score=0.8)]
    e = Omni3Deval([g for g in gts if g["image_id"] == 1], dts, "2D")
    e.evaluate(); e.accumulate()
    s = e.summarize()
    assert abs(s["AP"] - 100.0 * 51 / 101) < 1e-9                          # 51 of the 101 recall points (0 .. 0.50) have precision 1
    # false positive ranked first: precision 0.5 at recall 0.5
    dts2 = [_ann(1, 0, [10, 10, 50, 50], score=0.7), _ann(1, 0, [300, 300, 20, 20], score=0.8)]
    e = Omni3Deval([g for g in gts if g["image_id"] == 1], dts2, "2D")
    e.evaluate(); e.accumulate()
    assert abs(e.summarize()["AP"] - 100.0 * 0.5 * 51 / 101) < 1e-9
    # 3D mode with the fork's 2D-IoU behaviour needs no device; thresholds 0.05..0.50, depth ranges
    e3 = Omni3Deval(gts, perfect, "3D", fork_compat_2d_iou=True)
    e3.evaluate(); e3.accumulate()
    s3 = e3.summarize()
    assert abs(s3["AP"] - 100.0) < 1e-9 and abs(s3["APn"] - 100.0) < 1e-9 and s3["APf"] == -100          # no far GT -> undefined (-1)
    # ignore3D ground truth neither counts as a miss nor makes its match a false positive
    gi = [dict(gts[0]), dict(gts[1], ignore3D=1)]
    e3 = Omni3Deval(gi, [dict(gts[0], score=0.9), dict(gts[1], score=0.8)], "3D", fork_compat_2d_iou=True)
    e3.evaluate(); e3.accumulate()
    assert abs(e3.summarize()["AP"] - 100.0) < 1e-9


def test_depth_range_assignment():
    from ovmono3d_amd.evaluation.omni3d_eval import Omni3Deval
    gts = [_ann(1, 0, [10, 10, 50, 50], depth=5.0), _ann(1, 0, [100, 100, 40, 40], depth=20.0), _ann(1, 0, [200, 10, 40, 40], depth=50.0)]
    dts = [dict(gts[0], score=0.9), dict(gts[2], score=0.8)]                # near and far found, medium missed
    e = Omni3Deval(gts, dts, "3D", fork_compat_2d_iou=True)
    e.evaluate(); e.accumulate()
    s = e.summarize()
    assert abs(s["APn"] - 100.0) < 1e-9 and abs(s["APf"] - 100.0) < 1e-9 and abs(s["APm"]) < 1e-9
    assert abs(s["AP"] - 100.0 * 67 / 101) < 1e-9                           # recall reaches 2/3: points 0 .. 0.66


def test_omni3d_json_to_gt_fields_and_ignore_flags():
    from ovmono3d_amd.evaluation.omni3d_eval import omni3d_json_to_gt
    c3 = ob.make_box([0, 0, 5], [1, 1, 1], np.eye(3)).tolist()
    js = {"annotations": [
        {"image_id": 1, "category_id": 4, "bbox2D_proj": [10, 20, 60, 80], "bbox3D_cam": c3, "center_cam": [0, 0, 5.0], "behind_camera": False},
        {"image_id": 1, "category_id": 4, "bbox2D_proj": [-1, -1, -1, -1], "bbox2D_tight": [5, 5, 25, 45], "bbox3D_cam": c3, "center_cam": [0, 0, 7.0],
         "behind_camera": True},
        {"image_id": 2, "category_id": 9, "bbox2D_proj": [-1, -1, -1, -1], "bbox2D_tight": [-1, -1, -1, -1], "bbox2D_trunc": [-1, -1, -1, -1]}]}
    g = omni3d_json_to_gt(js)
    assert g[0]["bbox"] == [10.0, 20.0, 50.0, 60.0] and g[0]["depth"] == 5.0 and g[0]["ignore2D"] == 0 and g[0]["ignore3D"] == 0
    assert g[1]["bbox"] == [5.0, 5.0, 20.0, 40.0] and g[1]["ignore3D"] == 1 and g[1]["ignore2D"] == 0      # falls back to the tight box; behind the camera
    assert g[2]["ignore2D"] == 1 and g[2]["ignore3D"] == 1 and np.asarray(g[2]["bbox3D"]).shape == (8, 3)


# ---------------------------------------------------------------- GPU: the HIP kernel -----------------------------------------
@pytest.mark.gpu
def test_box3d_iou_kernel_matches_exact_oracle(device):
    from ovmono3d_amd.evaluation.omni3d_eval import box3d_overlap
    dt, gt = _rand_boxes(40, 1), _rand_boxes(30, 2)
    ref = ob.iou_matrix(dt, gt)
    got = box3d_overlap(torch.tensor(dt, dtype=torch.float32, device=device), torch.tensor(gt, dtype=torch.float32, device=device)).cpu().numpy()
    assert (ref > 0.05).sum() > 20                                          # the sample does exercise real overlaps
    assert np.abs(got - ref).max() < 2e-5, np.abs(got - ref).max()
    same = box3d_overlap(torch.tensor(dt, dtype=torch.float32, device=device), torch.tensor(dt, dtype=torch.float32, device=device)).cpu().numpy()
    assert np.abs(np.diag(same) - 1.0).max() < 1e-5                         # coincident faces are counted once
    # screening of the detections (reference :68-107, :160-167): a twisted (non-coplanar) and a flat (zero-area) box get IoU 0
    bad = dt[:2].copy()
    bad[0, 6] += [0.0, 0.0, 0.7]
    bad[1, 4:] = bad[1, :4]
    z = box3d_overlap(torch.tensor(bad, dtype=torch.float32, device=device), torch.tensor(dt[:2], dtype=torch.float32, device=device)).cpu().numpy()
    assert (z == 0).all()


@pytest.mark.gpu
def test_ap3d_true_iou_vs_fork_compat(device):
    """Two boxes that coincide in the image plane but sit 1.2 m apart in depth: the fork's 2D IoU calls it a match at every
    threshold, the true 3D IoU (0 here: the boxes are 1 m deep) does not."""
    from ovmono3d_amd.evaluation.omni3d_eval import Omni3Deval
    gts = [_ann(1, 0, [10, 10, 50, 50], depth=5.0)]
    dts = [_ann(1, 0, [10, 10, 50, 50], depth=6.2, score=0.9)]
    e = Omni3Deval(gts, dts, "3D", device=device)
    e.evaluate(); e.accumulate()
    assert abs(e.summarize()["AP"]) < 1e-9
    e = Omni3Deval(gts, dts, "3D", fork_compat_2d_iou=True)
    e.evaluate(); e.accumulate()
    assert abs(e.summarize()["AP"] - 100.0) < 1e-9
    # half a metre apart: IoU = 0.5 / 1.5 = 1/3 -> matched at thresholds 0.05 .. 0.30 (6 of 10)
    dts = [_ann(1, 0, [10, 10, 50, 50], depth=5.5, score=0.9)]
    e = Omni3Deval(gts, dts, "3D", device=device)
    e.evaluate(); e.accumulate()
    assert abs(e.summarize()["AP"] - 60.0) < 1e-6
