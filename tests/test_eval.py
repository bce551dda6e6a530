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


def _omni_anno(i, img, cat_id, cat_name, proj, **kw):
    """A complete Omni3D annotation (the fields datasets.py reads); keyword arguments override."""
    x1, y1, x2, y2 = proj
    c = [0.1 * i, 0.0, 5.0 + i]
    a = {"id": i, "image_id": img, "category_id": cat_id, "category_name": cat_name, "behind_camera": False, "valid3D": True,
         "dimensions": [1.0, 2.0, 3.0], "center_cam": c, "lidar_pts": -1, "segmentation_pts": -1, "depth_error": -1,
         "truncation": 0.0, "visibility": 1.0, "bbox2D_proj": [x1, y1, x2, y2], "bbox2D_tight": [-1, -1, -1, -1],
         "bbox2D_trunc": [x1, y1, x2, y2], "R_cam": np.eye(3).tolist(),
         "bbox3D_cam": ob.make_box(c, [3.0, 2.0, 1.0], np.eye(3)).tolist()}
    a.update(kw)
    return a


def _omni_json(annos, cats=((11, "bicycle"), (14, "books"), (18, "chair"), (2, "dontcare"), (40, "sofa"))):
    return {"info": {"name": "toy"}, "images": [{"id": 1, "height": 480, "width": 640}, {"id": 2, "height": 480, "width": 640}],
            "categories": [{"id": i, "name": n} for i, n in cats], "annotations": annos}


def test_annotation_ignore_rule_follows_the_reference_filter_settings():
    """Each clause of datasets.py:81-123 on its own, with the settings of the reference's do_test."""
    from ovmono3d_amd.evaluation import annotation_ignored, filter_settings_from_cfg
    fs = filter_settings_from_cfg(None)
    assert fs == {"category_names": [], "ignore_names": [], "truncation_thres": 0.99, "visibility_thres": 0.01, "min_height_thres": 0.0,
                  "max_height_thres": 1.5, "modal_2D_boxes": False, "trunc_2D_boxes": False, "max_depth": 1e8}
    fs.update(truncation_thres=1 / 3, visibility_thres=1 / 3, min_height_thres=0.0625, ignore_names=["dontcare"])
    base = dict(i=1, img=1, cat_id=18, cat_name="chair", proj=[100, 100, 200, 300])
    ok = lambda **kw: annotation_ignored(_omni_anno(**base, **kw), fs, 480)
    assert ok() is False
    assert ok(behind_camera=True) and ok(valid3D=False)
    assert ok(valid3D=False, dimensions=None)                                  # the validity tests return before anything else is read
    assert ok(dimensions=[1.0, 0.0, 1.0]) and ok(lidar_pts=0) and ok(segmentation_pts=0) and ok(depth_error=0.6)
    assert not ok(depth_error=0.5) and not ok(lidar_pts=7)
    assert ok(bbox2D_proj=[100, 100, 200, 130]) and not ok(bbox2D_proj=[100, 100, 200, 131])        # height <= 480/16 = 30
    assert ok(bbox2D_proj=[0, -200, 50, 520]) and not ok(bbox2D_proj=[0, -200, 50, 519])            # height >= 1.5 * 480
    assert ok(truncation=1 / 3) and not ok(truncation=0.33) and not ok(truncation=-1)
    assert ok(visibility=1 / 3) and not ok(visibility=0.34) and not ok(visibility=-1)
    assert annotation_ignored(_omni_anno(1, 1, 2, "dontcare", [100, 100, 200, 300]), fs, 480)
    # which 2D box is screened: truncated only when enabled, tight only with modal boxes
    tall_trunc = dict(bbox2D_trunc=[100, 100, 200, 120])
    assert not ok(**tall_trunc)
    fs["trunc_2D_boxes"] = True
    assert ok(**tall_trunc) and not ok(bbox2D_trunc=[-1, -1, -1, -1])
    fs["modal_2D_boxes"] = True
    assert not ok(bbox2D_tight=[100, 100, 200, 300], **tall_trunc)
    fs2 = dict(fs, max_depth=5.5)
    assert annotation_ignored(_omni_anno(**base), fs2, 480) and not annotation_ignored(_omni_anno(**dict(base, i=0)), fs2, 480)


def test_ground_truth_table_keeps_dataset_ids_and_the_reference_fields():
    from ovmono3d_amd.evaluation import Omni3DGroundTruth, filter_settings_from_cfg, ground_truth_records
    annos = [_omni_anno(1, 1, 18, "chair", [10, 20, 60, 120]),
             _omni_anno(2, 1, 11, "bicycle", [-1, -1, -1, -1], bbox2D_tight=[5, 5, 25, 45], bbox2D_trunc=[-1, -1, -1, -1], behind_camera=True),
             _omni_anno(3, 2, 14, "books", [-1, -1, -1, -1], bbox2D_trunc=[-1, -1, -1, -1]),       # no 2D box at all: dropped
             _omni_anno(4, 2, 40, "sofa", [10, 10, 90, 90]),                                        # not an evaluated category: dropped
             _omni_anno(5, 2, 2, "dontcare", [10, 10, 90, 90]),                                     # ignore name: kept, flagged
             _omni_anno(6, 2, 18, "chair", [10, 10, 90, 90], bbox2D_trunc=[12, 12, 80, 70], bbox2D_tight=[20, 20, 60, 50])]
    fs = filter_settings_from_cfg(None)
    fs.update(category_names=["chair", "books", "bicycle"], ignore_names=["dontcare"], trunc_2D_boxes=True)
    gt = Omni3DGroundTruth(_omni_json(annos), fs)
    assert gt.category_ids == [11, 14, 18] and gt.category_names == ["bicycle", "books", "chair"] and gt.image_ids == [1, 2]
    rec = {r["id"]: r for r in ground_truth_records(gt)}
    assert sorted(rec) == [1, 2, 5, 6]
    assert rec[1]["category_id"] == 18 and rec[1]["bbox"] == [10, 20, 50, 100] and rec[1]["area"] == 5000 and rec[1]["depth"] == 6.0
    assert rec[1]["ignore2D"] is False and rec[1]["ignore3D"] is False and rec[1]["iscrowd"] is False
    assert rec[2]["bbox"] == [5, 5, 20, 40] and rec[2]["ignore2D"] and rec[2]["ignore3D"]          # tight box as last resort; behind the camera
    assert rec[5]["ignore3D"] and rec[5]["category_id"] == 2
    assert rec[6]["bbox"] == [12, 12, 68, 58] and rec[6]["area"] == 68 * 58                         # truncated box wins over the projected one
    fs_modal = dict(fs, modal_2D_boxes=True)
    r6 = {r["id"]: r for r in ground_truth_records(Omni3DGroundTruth(_omni_json(annos), fs_modal))}[6]
    assert r6["bbox"] == [20, 20, 40, 30] and r6["area"] == 68 * 58                                # modal box stored, area of the first choice
    # no category list: every category of the file is evaluated and written back into the settings (:218-226)
    fs_all = filter_settings_from_cfg(None)
    g_all = Omni3DGroundTruth(_omni_json(annos), fs_all)
    assert g_all.category_ids == [2, 11, 14, 18, 40] and fs_all["category_names"] == ["dontcare", "bicycle", "books", "chair", "sofa"]
    # two files: images and annotations concatenate, the category table is the union sorted by id
    g2 = Omni3DGroundTruth([_omni_json(annos[:1]), _omni_json(annos[5:], cats=((18, "chair"), (50, "stove")))], filter_settings_from_cfg(None))
    assert g2.category_ids == [2, 11, 14, 18, 40, 50] and len(g2.images) == 4 and len(g2) == 2


def test_category_map_and_ap_when_dataset_ids_differ_from_class_indices():
    """The model emits class indices 0..2; the annotation file numbers the same categories 11 / 14 / 18."""
    from ovmono3d_amd.evaluation import CategoryMap, Omni3DGroundTruth, evaluate_omni3d, filter_settings_from_cfg
    cm = CategoryMap.from_names(["chair", "bicycle", "books"], _omni_json([])["categories"])
    assert cm.thing_classes == ["bicycle", "books", "chair"] and cm.dataset_id_to_contiguous == {11: 0, 14: 1, 18: 2}
    assert CategoryMap.from_meta(cm.to_meta()).contiguous_to_dataset_id == {0: 11, 1: 14, 2: 18}
    assert cm.to_meta()["thing_dataset_id_to_contiguous_id"] == {"11": 0, "14": 1, "18": 2}         # the category_meta.json schema
    with pytest.raises(KeyError):
        CategoryMap.from_names(["chair", "unicorn"], _omni_json([])["categories"])
    annos = [_omni_anno(1, 1, 18, "chair", [10, 20, 60, 120]), _omni_anno(2, 1, 11, "bicycle", [200, 50, 300, 250]),
             _omni_anno(3, 2, 14, "books", [30, 30, 130, 230]), _omni_anno(4, 2, 18, "chair", [300, 100, 400, 300], visibility=0.0)]
    fs = filter_settings_from_cfg(None)
    fs.update(category_names=["chair", "books", "bicycle"], trunc_2D_boxes=True)
    gt = Omni3DGroundTruth(_omni_json(annos), fs)

    def det(a, cls, score):
        b = a["bbox2D_proj"]
        return {"image_id": a["image_id"], "category_id": cls, "bbox": [b[0], b[1], b[2] - b[0], b[3] - b[1]], "score": score,
                "bbox3D": a["bbox3D_cam"], "depth": a["center_cam"][2], "center_cam": a["center_cam"], "dimensions": a["dimensions"],
                "pose": a["R_cam"]}
    dts = [det(annos[0], 2, 0.9), det(annos[1], 0, 0.8), det(annos[2], 1, 0.7), det(annos[3], 2, 0.6),
           det(annos[0], 7, 0.99),                                              # a class outside the map: dropped
           dict(det(annos[0], 2, 0.5), image_id=99)]                           # an image the dataset does not have: dropped
    r = evaluate_omni3d(gt, dts, only_2d=True, category_map=cm)
    assert abs(r["bbox_2D"]["AP"] - 100.0) < 1e-9
    assert sorted(r["bbox_2D_per_category"]) == ["bicycle", "books", "chair"]
    assert all(abs(v - 100.0) < 1e-9 for v in r["bbox_2D_per_category"].values())
    # without the map nothing lines up - what round 1's omni_ap.json silently did
    assert evaluate_omni3d(gt, dts, only_2d=True)["bbox_2D"]["AP"] == 0.0
    # the fork's passthrough rule mislabels as soon as an index is also a key
    cm50 = CategoryMap(["a", "b", "c", "d"], {0: 0, 1: 1, 3: 2, 4: 3})
    got = cm50.detections_to_dataset_ids([{"category_id": c} for c in range(4)])
    assert [d["category_id"] for d in got] == [0, 1, 3, 4]
    fork = cm50.detections_to_dataset_ids([{"category_id": c} for c in range(4)], passthrough_dataset_ids=True)
    assert [d["category_id"] for d in fork] == [0, 1, 3, 3]                     # index 3 ('d', id 4) taken for dataset id 3 ('c')


def test_collective_summary_over_two_datasets():
    """The cross-dataset numbers (reference summarize_all :427-620): one evaluation over the concatenated annotation files equals
    accumulating the datasets together - a category that is perfect in dataset A and missed in dataset B ends at the AP of the
    union, and the group means need every category of the group."""
    from ovmono3d_amd.evaluation import (OMNI3D_ALL, OMNI3D_IN, OMNI3D_OUT, CategoryMap, Omni3DGroundTruth, collective_summary, evaluate_omni3d,
                                         filter_settings_from_cfg)
    assert len(OMNI3D_ALL) == 50 and len(OMNI3D_IN) == 38 and len(OMNI3D_OUT) == 11 and (OMNI3D_IN & OMNI3D_OUT) == {"bicycle"}
    cats = ((11, "bicycle"), (18, "chair"))
    ja = _omni_json([_omni_anno(1, 1, 18, "chair", [10, 20, 60, 120]), _omni_anno(2, 1, 11, "bicycle", [200, 50, 300, 250])], cats)
    jb = _omni_json([_omni_anno(3, 1, 18, "chair", [30, 30, 130, 230])], cats)
    for im in jb["images"]:
        im["id"] += 10
    jb["annotations"][0]["image_id"] += 10
    fs = filter_settings_from_cfg(None)
    fs.update(category_names=["chair", "bicycle"], trunc_2D_boxes=True)
    gt = Omni3DGroundTruth([ja, jb], fs)
    assert gt.image_ids == [1, 2, 11, 12] and len(gt) == 3
    cm = CategoryMap.from_names(["chair", "bicycle"], ja["categories"])

    def det(a, cls, score):
        b = a["bbox2D_proj"]
        return {"image_id": a["image_id"], "category_id": cls, "bbox": [b[0], b[1], b[2] - b[0], b[3] - b[1]], "score": score}
    dts = [det(ja["annotations"][0], 1, 0.9), det(ja["annotations"][1], 0, 0.8)]          # dataset B's chair is missed
    r = evaluate_omni3d(gt, dts, only_2d=True, category_map=cm)
    assert abs(r["bbox_2D_per_category"]["bicycle"] - 100.0) < 1e-9 and abs(r["bbox_2D_per_category_AR"]["bicycle"] - 100.0) < 1e-9
    assert abs(r["bbox_2D_per_category"]["chair"] - 100.0 * 51 / 101) < 1e-9            # 1 of 2 chairs found: recall points 0 .. 0.50
    assert abs(r["bbox_2D_per_category_AR"]["chair"] - 50.0) < 1e-9
    assert {"AP", "AP50", "AP75", "AP95", "APs", "APm", "APl", "AR1", "AR10", "AR100"} <= set(r["bbox_2D"])
    c = collective_summary(r)
    assert abs(c["<Concat>"]["AP2D"] - 0.5 * (100.0 + 100.0 * 51 / 101)) < 1e-9 and abs(c["<Concat>"]["AR2D"] - 75.0) < 1e-9
    assert all(np.isnan(c[g]["AP2D"]) for g in ("Omni3D_Out", "Omni3D_In", "Omni3D"))     # groups need all of their categories
    full = {"bbox_2D_per_category": {n: 50.0 for n in OMNI3D_ALL}, "bbox_2D_per_category_AR": {n: 60.0 for n in OMNI3D_ALL}}
    full["bbox_2D_per_category"]["car"] = 94.0
    cf = collective_summary(full)
    assert abs(cf["Omni3D_Out"]["AP2D"] - (50.0 + 44.0 / 11)) < 1e-9 and cf["Omni3D_In"]["AP2D"] == 50.0 and abs(cf["Omni3D"]["AP2D"] - (50.0 + 44.0 / 50)) < 1e-9
    assert cf["Omni3D"]["AR2D"] == 60.0 and np.isnan(cf["Omni3D"]["AP3D"])


def test_nhd_known_values():
    from ovmono3d_amd.evaluation import cuboid_corners, disentangled_nhd, hungarian_distance
    gt = {"xy": [0.0, 0.0], "z": 5.0, "dimensions": [1.0, 2.0, 3.0], "pose": np.eye(3)}
    c = cuboid_corners(gt["xy"], gt["z"], gt["dimensions"], gt["pose"])
    assert c.shape == (8, 3) and np.allclose(c.max(0) - c.min(0), [3.0, 2.0, 1.0])                # L on x, H on y, W on z
    assert disentangled_nhd(gt, gt) == {"overall": 0.0, "xy": 0.0, "z": 0.0, "dimensions": 0.0, "pose": 0.0}
    diag = float(np.sqrt(14.0))
    shifted = dict(gt, xy=[0.3, 0.4])                                           # every corner moves 0.5; the identity assignment is optimal
    r = disentangled_nhd(shifted, gt)
    assert abs(r["overall"] - 8 * 0.5 / diag) < 1e-6 and abs(r["xy"] - r["overall"]) < 1e-6 and r["z"] == 0 and r["dimensions"] == 0 and r["pose"] == 0
    deeper = dict(shifted, z=5.25)
    r = disentangled_nhd(deeper, gt)
    assert abs(r["z"] - 8 * 0.25 / diag) < 1e-6 and abs(r["xy"] - 8 * 0.5 / diag) < 1e-6
    assert abs(r["overall"] - 8 * float(np.sqrt(0.25 + 0.0625)) / diag) < 1e-6
    # a half turn about y maps the box onto itself: the assignment absorbs it
    Ry = Rotation.from_euler("y", 180, degrees=True).as_matrix()
    assert disentangled_nhd(dict(gt, pose=Ry), gt)["pose"] < 1e-5
    # dimensions only: each corner moves by half the growth along each axis
    r = disentangled_nhd(dict(gt, dimensions=[1.0, 2.0, 3.4]), gt)
    assert abs(r["dimensions"] - 8 * 0.2 / diag) < 1e-6 and r["pose"] == 0
    # the assignment is a true minimum: never above the identity pairing, and invariant to corner order
    g = np.random.default_rng(0)
    a, b = g.normal(size=(8, 3)), g.normal(size=(8, 3))
    d = hungarian_distance(a, b)
    assert d <= np.linalg.norm(a - b, axis=1).sum() / np.linalg.norm(b.max(0) - b.min(0)) + 1e-12
    assert abs(hungarian_distance(a[g.permutation(8)], b) - d) < 1e-12


def test_nhd_is_collected_over_iou_matched_pairs():
    from ovmono3d_amd.evaluation import Omni3Deval
    R = np.eye(3).tolist()
    def rec(img, box, depth, **kw):
        x, y, w, h = box
        d = _ann(img, 0, box, depth=depth, **kw)
        d.update(center_cam=[x + w / 2, y + h / 2, depth], dimensions=[1.0, h, w])
        d["pose" if "score" in kw else "R_cam"] = R
        return d
    gts = [rec(1, [10, 10, 50, 50], 5.0), rec(1, [100, 100, 40, 40], 8.0), rec(2, [20, 20, 60, 60], 5.0)]
    dts = [rec(1, [10, 10, 50, 50], 5.0, score=0.9),                           # exact
           rec(1, [100, 100, 40, 40], 8.0 + 0.1, score=0.8),                    # 2D IoU 1 (fork-compatible IoU): paired, z off by 0.1
           rec(2, [200, 200, 60, 60], 5.0, score=0.7)]                          # no overlap: not paired
    e = Omni3Deval(gts, dts, "3D", fork_compat_2d_iou=True)
    e.evaluate(); e.accumulate()
    acc = e.eval["nhd_accumulators"]
    assert len(acc["overall"]) == 2 and sorted(acc) == ["dimensions", "overall", "pose", "xy", "z"]
    diag = float(np.sqrt(40 ** 2 + 40 ** 2 + 1))
    assert abs(sorted(acc["z"])[1] - 8 * 0.1 / diag) < 1e-5 and sorted(acc["z"])[0] == 0
    s = e.summarize()
    assert abs(s["NHD"] - 0.5 * 8 * 0.1 / diag) < 1e-5 and s["NHD-xy"] == 0 and s["NHD-pose"] == 0
    e2 = Omni3Deval(gts, dts, "2D")
    e2.evaluate(); e2.accumulate()
    assert "nhd_accumulators" not in e2.eval


def _ap_by_definition(gts, dts, mode, img_ids, cat_ids):
    """COCO AP written out from its definition with scalar loops, for one evaluator configuration: per category and range,
    detections in score order are matched greedily per threshold; AP samples, at 101 recall levels, the best precision
    reached at that recall or beyond."""
    from ovmono3d_amd.evaluation.omni3d_eval import Omni3DParams, iou2d_xywh
    p = Omni3DParams(mode)
    flag, key = ("ignore2D", "area") if mode == "2D" else ("ignore3D", "depth")
    T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(cat_ids), len(p.areaRng), len(p.maxDets)
    prec, rec = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M))
    for k, cat in enumerate(cat_ids):
        for a, (lo, hi) in enumerate(p.areaRng):
            for m, cap in enumerate(p.maxDets):
                for t, thr in enumerate(p.iouThrs):
                    rows, n_pos = [], 0                                         # (score, is_tp, counts)
                    for img in img_ids:
                        G = [g for g in gts if g["image_id"] == img and g["category_id"] == cat]
                        D = sorted([d for d in dts if d["image_id"] == img and d["category_id"] == cat], key=lambda d: -d["score"])[:cap]
                        ign = [bool(g.get(flag, 0)) or g[key] < lo or g[key] > hi for g in G]
                        n_pos += sum(1 for i in ign if not i)
                        iou = iou2d_xywh(np.array([d["bbox"] for d in D]), np.array([g["bbox"] for g in G])) if D and G else None
                        used = set()
                        for di, d in enumerate(D):
                            best, best_v = None, min(thr, 1 - 1e-10)
                            for want_ignored in (False, True):
                                if best is not None:
                                    break
                                for gi in range(len(G)):
                                    if ign[gi] == want_ignored and gi not in used and iou[di, gi] >= best_v:
                                        best, best_v = gi, iou[di, gi]
                            if best is not None:
                                used.add(best)
                            out = d[key] < lo or d[key] > hi
                            counts = not (ign[best] if best is not None else out)
                            rows.append((d["score"], best is not None, counts))
                    if n_pos == 0:
                        continue
                    order = sorted(range(len(rows)), key=lambda i: -rows[i][0])                   # stable
                    tp = fp = 0
                    curve = []
                    for i in order:
                        _, hit, counts = rows[i]
                        tp += int(hit and counts)
                        fp += int((not hit) and counts)
                        curve.append((tp / n_pos, tp / (tp + fp + np.spacing(1))))
                    rec[t, k, a, m] = curve[-1][0] if curve else 0.0
                    for r, level in enumerate(p.recThrs):
                        beyond = [pr for rc, pr in curve if rc >= level]
                        prec[t, r, k, a, m] = max(beyond) if beyond else 0.0
    return prec, rec


@pytest.mark.parametrize("mode", ["2D", "3D"])
def test_vectorised_evaluator_equals_the_definition_on_random_scenes(mode):
    """Crowded random scenes with ties in score and IoU, ignored and out-of-range ground truth, more than maxDets detections
    per cell, empty cells: the array formulation must give the same precision / recall tables, exactly."""
    from ovmono3d_amd.evaluation.omni3d_eval import Omni3Deval
    g = np.random.default_rng(5 if mode == "2D" else 6)
    gts, dts = [], []
    grid = [10, 40, 70, 100]
    for img in range(1, 7):
        for cat in range(3):
            if g.random() < 0.15:
                continue
            for _ in range(int(g.integers(0, 6))):
                x, y = g.choice(grid), g.choice(grid)
                w, h = g.choice([20, 30, 110]), g.choice([20, 30, 110])
                gts.append(_ann(img, cat, [float(x), float(y), float(w), float(h)], depth=float(g.choice([5.0, 10.0, 20.0, 35.0, 60.0])),
                                ignore2D=int(g.random() < 0.2), ignore3D=int(g.random() < 0.2)))
            nd = int(g.integers(0, 9)) if (img, cat) != (2, 1) else 130
            for _ in range(nd):
                x, y = g.choice(grid), g.choice(grid)
                w, h = g.choice([20, 30, 110]), g.choice([20, 30, 110])
                dts.append(_ann(img, cat, [float(x), float(y), float(w), float(h)], depth=float(g.choice([5.0, 10.0, 20.0, 35.0, 60.0])),
                                score=float(g.choice([0.9, 0.8, 0.8, 0.5, 0.3, 0.3, 0.1]) if nd < 100 else g.random())))
    img_ids, cat_ids = list(range(1, 8)), [0, 1, 2, 3]                          # image 7 and category 3 exist but are empty
    e = Omni3Deval(gts, dts, mode, fork_compat_2d_iou=True, img_ids=img_ids, cat_ids=cat_ids)
    e.evaluate(); e.accumulate()
    prec, rec = _ap_by_definition(e._gts_all, e._dts_all, mode, img_ids, cat_ids)
    assert (prec > 0).sum() > 500 and ((prec > 0) & (prec < 1)).sum() > 200    # the sample has real curves, not all-or-nothing
    assert np.array_equal(e.eval["precision"], prec)
    assert np.array_equal(e.eval["recall"], rec)


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
