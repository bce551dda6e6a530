"""GPU: the drop-in entry points run end to end on synthetic data (demo.py with an oracle-2D boxes file,
inference_on_dataset over a tiny Omni3D-format dataset)."""
import json
import os
import subprocess
import sys

import numpy as np
import torch
import pytest

from common import ROOT

pytestmark = pytest.mark.gpu


def _write_images(folder, n=2, size=None):
    from PIL import Image
    rng = np.random.default_rng(0)
    names = []
    for i in range(n):
        arr = rng.integers(0, 255, ((120 + 20 * i, 160, 3) if size is None else (size[0], size[1], 3)), dtype=np.uint8)
        name = f"img{i:03d}"
        Image.fromarray(arr).save(os.path.join(folder, name + ".png"))
        names.append(name)
    return names


def test_demo_with_boxes_file(device, tmp_path):
    inp = tmp_path / "in"; inp.mkdir()
    out = tmp_path / "out"
    names = _write_images(str(inp))
    labels = {n: ["chair", "table"] for n in names}
    boxes = {n: [{"bbox": [20, 20, 60, 50], "category_id": 0, "score": 0.9}, {"bbox": [70, 40, 50, 60], "category_id": 1, "score": 0.8}]
             for n in names}
    (tmp_path / "labels.json").write_text(json.dumps(labels))
    (tmp_path / "boxes.json").write_text(json.dumps(boxes))
    cmd = [sys.executable, os.path.join(ROOT, "demo", "demo.py"), "--config-file", os.path.join(ROOT, "configs", "OVMono3D_dinov2_SFP.yaml"),
           "--input-folder", str(inp), "--labels-file", str(tmp_path / "labels.json"), "--boxes-file", str(tmp_path / "boxes.json"),
           "--threshold", "0.0", "MODEL.DINO.MODEL_NAME", "vittest14", "MODEL.FPN.SQUARE_PAD", "224", "INPUT.MIN_SIZE_TEST", "140",
           "INPUT.MAX_SIZE_TEST", "224", "MODEL.WEIGHTS", "synthetic://vittest14?seed=3", "OUTPUT_DIR", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for n in names:
        d = json.loads((out / f"{n}_dets.json").read_text())
        assert len(d["detections"]) == 2 and d["detections"][0]["category"] in ("chair", "table")
        assert np.asarray(d["detections"][0]["corners3D"]).shape == (8, 3)


def test_gdino_head_demo_fails_loudly_without_detector(device, tmp_path):
    inp = tmp_path / "in"; inp.mkdir()
    names = _write_images(str(inp), 1)
    (tmp_path / "labels.json").write_text(json.dumps({names[0]: ["chair"]}))
    cmd = [sys.executable, os.path.join(ROOT, "demo", "demo.py"), "--config-file", os.path.join(ROOT, "configs", "OVMono3D_dinov2_SFP.yaml"),
           "--input-folder", str(inp), "--labels-file", str(tmp_path / "labels.json"), "MODEL.ROI_HEADS.NAME", "ROIHeads3DGDINO",
           "MODEL.DINO.MODEL_NAME", "vittest14", "MODEL.FPN.SQUARE_PAD", "224", "INPUT.MIN_SIZE_TEST", "140", "INPUT.MAX_SIZE_TEST", "224",
           "MODEL.WEIGHTS", "synthetic://vittest14?seed=3", "OUTPUT_DIR", str(tmp_path / "o")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "NotImplementedError" in r.stderr


def test_gdino_head_demo_with_native_detector(device, tmp_path):
    """ROIHeads3DGDINO end to end through demo.py: native GroundingDINO (random Swin-B/BERT weights) -> phrase logits ->
    NMS -> cube head."""
    inp = tmp_path / "in"; inp.mkdir()
    names = _write_images(str(inp), 1)
    (tmp_path / "labels.json").write_text(json.dumps({names[0]: ["chair", "dining table"]}))
    cmd = [sys.executable, os.path.join(ROOT, "demo", "demo.py"), "--config-file", os.path.join(ROOT, "configs", "OVMono3D_dinov2_SFP.yaml"),
           "--input-folder", str(inp), "--labels-file", str(tmp_path / "labels.json"), "--threshold", "0.0",
           "MODEL.ROI_HEADS.NAME", "ROIHeads3DGDINO", "MODEL.AMD.GDINO_WEIGHTS", "synthetic://gdino?seed=1",
           "MODEL.DINO.MODEL_NAME", "vittest14", "MODEL.FPN.SQUARE_PAD", "280", "INPUT.MIN_SIZE_TEST", "210", "INPUT.MAX_SIZE_TEST", "280",
           "MODEL.WEIGHTS", "synthetic://vittest14?seed=3", "OUTPUT_DIR", str(tmp_path / "o")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads((tmp_path / "o" / f"{names[0]}_dets.json").read_text())
    # box_threshold 0.001 on sigmoid phrase scores: random weights keep (almost) every query, NMS leaves many
    assert len(d["detections"]) > 0 and all(x["category"] in ("chair", "dining table") for x in d["detections"])
    assert np.isfinite(np.asarray(d["detections"][0]["corners3D"])).all()


def test_eval_only_entry_point(device, tmp_path):
    root = tmp_path / "datasets"; (root / "Omni3D").mkdir(parents=True); (root / "imgs").mkdir()
    names = _write_images(str(root / "imgs"), 3)
    images = [{"id": 100 + i, "file_path": f"imgs/{n}.png", "width": 160, "height": 120 + 20 * i, "dataset_id": 0,
               "K": [[300.0, 0, 80], [0, 300.0, 60 + 10 * i], [0, 0, 1]]} for i, n in enumerate(names)]
    c3 = [[x, y, z] for z in (4.5, 5.5) for (x, y) in ((-0.5, -0.5), (0.5, -0.5), (0.5, 0.5), (-0.5, 0.5))]
    full = {"valid3D": True, "dimensions": [1.0, 1.0, 1.0], "lidar_pts": -1, "segmentation_pts": -1, "depth_error": -1, "truncation": 0.0,
            "visibility": 1.0, "bbox2D_trunc": [-1, -1, -1, -1], "bbox2D_tight": [-1, -1, -1, -1], "R_cam": np.eye(3).tolist(),
            "bbox3D_cam": c3, "center_cam": [0, 0, 5.0], "behind_camera": False}
    anns = [dict(full, id=1, image_id=100, category_id=18, category_name="chair", bbox2D_proj=[20, 20, 80, 70]),
            dict(full, id=2, image_id=101, category_id=19, category_name="cup", bbox2D_proj=[-1, -1, -1, -1], bbox2D_tight=[30, 30, 90, 90],
                 behind_camera=True),
            dict(full, id=3, image_id=102, category_id=40, category_name="sofa", bbox2D_proj=[20, 20, 80, 70])]     # not a base category
    # the Objectron numbering (ids 11, 14..21) plus one category that is not evaluated: dataset ids differ from class indices
    cats = [{"id": i, "name": n} for i, n in zip((11, 14, 15, 16, 17, 18, 19, 20, 21, 40),
                                                 ("bicycle", "books", "bottle", "camera", "cereal box", "chair", "cup", "laptop", "shoes", "sofa"))]
    (root / "Omni3D" / "Objectron_test.json").write_text(json.dumps({"info": {"name": "Objectron"}, "images": images, "annotations": anns,
                                                                     "categories": cats}))
    # a second dataset of the mode (its own image / annotation ids): the entry point then also writes the collective numbers
    images2 = [dict(im, id=im["id"] + 100) for im in images[:2]]
    anns2 = [dict(full, id=11, image_id=200, category_id=11, category_name="bicycle", bbox2D_proj=[10, 10, 90, 100])]
    (root / "Omni3D" / "Toy_test.json").write_text(json.dumps({"info": {"name": "Toy"}, "images": images2, "annotations": anns2, "categories": cats}))
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "train_net.py"), "--eval-only", "--config-file",
           os.path.join(ROOT, "configs", "OVMono3D_dinov2_SFP.yaml"), "--datasets-root", str(root / "Omni3D"), "--image-root", str(root),
           "MODEL.DINO.MODEL_NAME", "vittest14", "MODEL.FPN.SQUARE_PAD", "224", "INPUT.MIN_SIZE_TEST", "140", "INPUT.MAX_SIZE_TEST", "224",
           "MODEL.WEIGHTS", "synthetic://vittest14?seed=3", "OUTPUT_DIR", str(out), "DATASETS.TEST_BASE", "('Objectron_test', 'Toy_test')"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    both = json.loads((out / "inference" / "iter_final" / "omni_ap_all.json").read_text())
    assert set(both["collective"]) == {"<Concat>", "Omni3D_Out", "Omni3D_In", "Omni3D"} and "AP3D" in both["collective"]["<Concat>"]
    assert set(both["bbox_3D_per_category_AR"]) == set(both["bbox_2D_per_category"])
    res = json.loads((out / "inference" / "iter_final" / "Objectron_test" / "omni_instances_results.json").read_text())
    assert len(res) > 0 and {"image_id", "category_id", "bbox", "score", "bbox3D", "pose", "depth"} <= set(res[0])
    ap = json.loads((out / "inference" / "iter_final" / "Objectron_test" / "omni_ap.json").read_text())      # AP evaluator ran on the ground truth
    assert {"AP", "AP15", "AP25", "AP50", "APn", "APm", "APf"} <= set(ap["bbox_3D"]) and {"AP", "AP50", "AP75"} <= set(ap["bbox_2D"])
    assert {r_["image_id"] for r_ in res} <= {100, 101, 102}
    assert "NHD" in ap["bbox_3D"] and set(ap["bbox_3D_per_category"]) == {"bicycle", "books", "bottle", "camera", "cereal box", "chair", "cup",
                                                                            "laptop", "shoes"}
    meta = json.loads((out / "inference" / "iter_final" / "Objectron_test" / "category_meta.json").read_text())
    assert meta["thing_dataset_id_to_contiguous_id"] == {"11": 0, "14": 1, "15": 2, "16": 3, "17": 4, "18": 5, "19": 6, "20": 7, "21": 8}


def test_oracle2d_producer_roundtrip(device, tmp_path):
    """tools/make_oracle2d.py writes the file format merge_oracle2d_to_detection_dicts reads (reference build.py:45-54)."""
    from ovmono3d_amd.data.feeding import load_omni3d_json, merge_oracle2d_to_detection_dicts
    root = tmp_path / "datasets"; (root / "Omni3D").mkdir(parents=True); (root / "imgs").mkdir()
    names = _write_images(str(root / "imgs"), 2, size=(300, 400))
    images = [{"id": 7 + i, "file_path": f"imgs/{n}.png", "width": 400, "height": 300, "dataset_id": 0, "K": [[500.0, 0, 200], [0, 500.0, 150], [0, 0, 1]]}
              for i, n in enumerate(names)]
    cats = [{"id": 3, "name": "chair"}, {"id": 9, "name": "dining table"}]
    ds = root / "Omni3D" / "Toy_test.json"
    ds.write_text(json.dumps({"images": images, "annotations": [], "categories": cats}))
    out = root / "Omni3D" / "gdino_toy_oracle_2d.json"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "make_oracle2d.py"), "--config-file", os.path.join(ROOT, "configs", "OVMono3D_dinov2_SFP.yaml"),
           "--dataset", str(ds), "--image-root", str(root), "--output", str(out), "INPUT.MIN_SIZE_TEST", "300", "INPUT.MAX_SIZE_TEST", "400",
           "MODEL.AMD.GDINO_WEIGHTS", "synthetic://gdino?seed=1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    data = json.loads(out.read_text())
    assert [d["image_id"] for d in data] == [7, 8] and all(i["category_id"] in (0, 1) for d in data for i in d["instances"])
    assert sum(len(d["instances"]) for d in data) > 0
    dicts = load_omni3d_json(str(ds), str(root))
    merge_oracle2d_to_detection_dicts(dicts, str(out))
    o = dicts[0]["oracle2D"]
    assert o["gt_bbox2D"].shape[1] == 4 and len(o["gt_classes"]) == len(o["gt_scores"]) == o["gt_bbox2D"].shape[0]
    assert torch.isfinite(o["gt_bbox2D"]).all()              # not clipped to the image: the reference's GroundingDINO glue does not clip either


def test_rccl_gather_c_abi_one_rank(device):
    """ovm_comm_unique_id / ovm_comm_init / ovm_gather_counts / ovm_gather_records on a one-rank RCCL communicator (the N > 1
    exchange needs N GPUs; this exercises the C ABI, the counts exchange and the payload path so that the first 8-GPU run is
    not also its first execution), and the routing of evaluation.distributed.gather_records through it.
    Replaces comm.gather(dst=0) of reference omni3d_evaluation.py:717-720."""
    from ovmono3d_amd.evaluation import distributed as D
    comm = D.NativeComm(D.NativeComm.new_unique_id(), 0, 1, device)
    try:
        rec = torch.arange(5 * 48, dtype=torch.float32, device=device).view(5, 48)
        out, counts = comm.gather(rec)
        assert counts == [5] and torch.equal(out, rec) and out.data_ptr() != rec.data_ptr()
        out0, counts0 = comm.gather(rec[:0])                                  # a rank without detections
        assert counts0 == [0] and out0.shape == (0, 48)
        D.set_native_comm(comm)
        out2, counts2 = D.gather_records(rec)
        assert counts2 == [5] and torch.equal(out2, rec) and out2.data_ptr() != rec.data_ptr()
    finally:
        D.set_native_comm(None)
        comm.close()


def test_bench_gpus2_real_steps_gloo(device):
    """`python bench.py --gpus 2` as the driver calls it (no launcher): bench.py starts its own two ranks as a child process before
    touching the GPU, both ranks run real steps of a tiny model on this one card (OVM_BENCH_BACKEND=gloo: ranks may share a device),
    and rank 0 prints one line with n_gpus 2 and the whole-job rate."""
    env = dict(os.environ, OVM_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "OVM_BENCH_DRYRUN"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--model", "vittest14", "--canvas", "224",
           "--net-res", "140", "--proposals", "oracle2d", "--boxes", "8", "--no-alt", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-2
    assert d["scaling"] == "weak" and "dry_run" not in d
