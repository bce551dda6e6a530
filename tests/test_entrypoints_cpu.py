"""Host-side entry-point plumbing that needs no GPU: data feeding, record <-> JSON, the per-rank loop with a stub model,
and the 2-rank gloo run of inference_on_dataset (dataset order preserved through shards + gather)."""
import json
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import build_cfg
from ovmono3d_amd.structures import Boxes, Instances


def test_resize_shortest_edge_matches_reference_examples():
    """500x500 -> 532x532, 480x640 -> 532x709 under ResizeShortestEdge(532, 896)."""
    from ovmono3d_amd.data import ResizeShortestEdge
    r = ResizeShortestEdge(532, 896)
    assert r.output_shape(500, 500) == (532, 532)
    assert r.output_shape(480, 640) == (532, 709)
    assert r.output_shape(300, 1200) == (224, 896)
    img = np.random.default_rng(0).integers(0, 255, (120, 160, 3), dtype=np.uint8)
    assert ResizeShortestEdge(60, 896)(img).shape == (60, 80, 3)


def _fake_instances(n, seed):
    g = torch.Generator().manual_seed(seed)
    i = Instances((10, 10))
    i.pred_boxes = Boxes(torch.rand(n, 4, generator=g) * 10)
    i.scores = torch.rand(n, generator=g)
    i.pred_classes = torch.randint(0, 50, (n,), generator=g)
    i.pred_bbox3D = torch.randn(n, 8, 3, generator=g)
    i.pred_center_cam = torch.randn(n, 3, generator=g)
    i.pred_center_2D = torch.randn(n, 2, generator=g)
    i.pred_dimensions = torch.rand(n, 3, generator=g)
    i.pred_pose = torch.randn(n, 3, 3, generator=g)
    return i


def test_records_json_equals_instances_to_coco_json():
    from ovmono3d_amd.evaluation.omni3d_evaluation import _json_of_records, _records_of, instances_to_coco_json
    inst = _fake_instances(5, 1)
    a = instances_to_coco_json(inst, 42)
    b = _json_of_records(_records_of(inst, 0, torch.device("cpu")), 42)
    assert len(a) == len(b) == 5
    for x, y in zip(a, b):
        assert x["image_id"] == y["image_id"] == 42 and x["category_id"] == y["category_id"]
        for k in ("bbox", "bbox3D", "center_cam", "center_2D", "dimensions", "pose"):
            assert np.allclose(np.asarray(x[k]), np.asarray(y[k]), atol=1e-6), k
        assert abs(x["score"] - y["score"]) < 1e-7 and abs(x["depth"] - y["depth"]) < 1e-7
    assert instances_to_coco_json(Instances((4, 4)), 1) == []


class _StubModel:
    """Deterministic stand-in: image i yields (image_id % 3) detections."""

    def eval(self):
        return self

    def __call__(self, inputs, prompt_depth=None):
        return [{"instances": _fake_instances(int(x["image_id"]) % 3, int(x["image_id"]))} for x in inputs]


class _EmptyFieldsModel(_StubModel):
    """image_id % 3 == 0 -> an Instances with NO fields at all (the n == 0 early return of _forward_cube for a
    GroundingDINO image without boxes); the others as _StubModel."""

    def __call__(self, inputs, prompt_depth=None):
        out = []
        for x in inputs:
            n = int(x["image_id"]) % 3
            out.append({"instances": _fake_instances(n, int(x["image_id"])) if n else Instances((10, 10))})
        return out


class _Loader(list):
    pass


def _make_loader(n, rank, world):
    from ovmono3d_amd import lib
    b, e = lib.shard_range(n, rank, world)
    return _Loader([[{"image_id": i, "K": np.eye(3).tolist(), "width": 4, "height": 4}] for i in range(b, e)])


def test_inference_on_dataset_single_process():
    from ovmono3d_amd.evaluation import inference_on_dataset
    res = inference_on_dataset(_StubModel(), _make_loader(7, 0, 1))
    assert [r["image_id"] for r in res] == list(range(7))
    assert [len(r["instances"]) for r in res] == [i % 3 for i in range(7)]
    assert set(res[2]["instances"][0]) >= {"image_id", "category_id", "bbox", "score", "bbox3D", "center_cam", "depth"}


def test_inference_on_dataset_empty_images_between_nonempty():
    """reference omni3d_evaluation.py:717-720: images without detections still occupy their slot in dataset order."""
    from ovmono3d_amd.evaluation import inference_on_dataset
    res = inference_on_dataset(_EmptyFieldsModel(), _make_loader(7, 0, 1))
    assert [len(r["instances"]) for r in res] == [i % 3 for i in range(7)]
    assert res[3]["instances"] == [] and res[4]["image_id"] == 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ovmono3d_amd.evaluation import inference_on_dataset
    res = inference_on_dataset(_StubModel(), _make_loader(n, rank, world))
    if rank == 0:
        q.put(json.dumps([[r["image_id"], len(r["instances"]), r["instances"][0]["score"] if r["instances"] else None] for r in res]))
    else:
        assert res == []
    dist.barrier()
    dist.destroy_process_group()


def test_inference_on_dataset_two_ranks_gloo():
    world, n = 2, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = json.loads(q.get(timeout=120))
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    from ovmono3d_amd.evaluation import inference_on_dataset
    ref = inference_on_dataset(_StubModel(), _make_loader(n, 0, 1))
    exp = [[r["image_id"], len(r["instances"]), r["instances"][0]["score"] if r["instances"] else None] for r in ref]
    assert got == exp


class _OnlyFirstShardModel(_StubModel):
    """detections only for image ids < 2: with 4 images on 2 ranks the second rank gathers zero records."""

    def __call__(self, inputs, prompt_depth=None):
        return [{"instances": _fake_instances(2, int(x["image_id"])) if int(x["image_id"]) < 2 else Instances((10, 10))} for x in inputs]


def _worker_empty_rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ovmono3d_amd.evaluation import inference_on_dataset
    res = inference_on_dataset(_OnlyFirstShardModel(), _make_loader(4, rank, world))
    if rank == 0:
        q.put(json.dumps([[r["image_id"], len(r["instances"])] for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def test_inference_on_dataset_two_ranks_one_rank_all_empty():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_empty_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = json.loads(q.get(timeout=120))
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == [[0, 2], [1, 2], [2, 0], [3, 0]]


def test_synthetic_checkpoint_uri_and_key_tree():
    """DetectionCheckpointer accepts synthetic://<arch>?seed=N; keys follow the reference module tree (nohup.out:563-684)."""
    from ovmono3d_amd.util.synth_weights import synth_state_dict
    sd = synth_state_dict("vittest14", seed=0)
    for k in ("backbone.net.vit.blocks.0.attn.qkv.weight", "backbone.simfp_2.0.weight", "backbone.simfp_4.2.norm.bias",
              "backbone.net.depth_fusion.weight", "proposal_generator.rpn_head.anchor_deltas.weight",
              "roi_heads.box_predictor.bbox_pred.weight", "roi_heads.cube_head.feature_generator.fc1.weight",
              "roi_heads.cube_head.bbox_3D_uncertainty.bias"):
        assert k in sd, k
    assert sd["backbone.simfp_2.0.weight"].shape == (128, 64, 2, 2)
    assert sd["roi_heads.cube_head.feature_generator.fc1.weight"].shape == (1024, 12544)


def test_oracle2d_file_defaults_resolve(tmp_path):
    """DATASETS.ORACLE2D_FILES carries root-relative defaults with the reference's file names (config.py:42-76) and
    tools/train_net.py finds them under --datasets-root for TEST.CAT_MODE novel + target_aware (BASELINE config 3)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ovm_train_net", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "train_net.py"))
    tn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tn)
    cfg = build_cfg()
    assert cfg.DATASETS.ORACLE2D_FILES.EVAL_MODE == "target_aware"
    assert tn.oracle2d_file(cfg, "novel", "KITTI_test_novel", str(tmp_path)) is None
    f = tmp_path / "gdino_kitti_novel_oracle_2d.json"
    f.write_text("[]")
    assert tn.oracle2d_file(cfg, "novel", "KITTI_test_novel", str(tmp_path)) == str(f)
    assert tn.oracle2d_file(cfg, "base", "Objectron_test", str(tmp_path)) is None
    (tmp_path / "gdino_objectron_base_oracle_2d.json").write_text("[]")
    assert tn.oracle2d_file(cfg, "base", "Objectron_test", str(tmp_path)).endswith("gdino_objectron_base_oracle_2d.json")
    assert tn.oracle2d_file(cfg, "base", "NoSuch_test", str(tmp_path)) is None


def test_category_map_of_the_eval_entry_point(tmp_path):
    """tools/train_net.py picks the class-index <-> dataset-id map from the annotation file's category table (for Objectron
    the numbering 11, 14..21 -> 0..8, i.e. what the reference hard-codes as configs/category_objectron.json), or from a
    category_meta.json-style file."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from tools.train_net import category_map_for, make_cfg
    from ovmono3d_amd.evaluation import Omni3DGroundTruth, eval_filter_settings
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = make_cfg(os.path.join(root, "configs", "OVMono3D_dinov2_SFP.yaml"), [])
    fs = eval_filter_settings(cfg, "base")
    assert fs["min_height_thres"] == 0.0625 and fs["max_depth"] == 1e8 and fs["trunc_2D_boxes"] is True
    assert fs["truncation_thres"] == cfg.TEST.TRUNCATION_THRES and fs["visibility_thres"] == cfg.TEST.VISIBILITY_THRES
    assert fs["category_names"] == list(cfg.DATASETS.CATEGORY_NAMES_BASE) and fs["ignore_names"] == ["dontcare", "ignore", "void"]
    names = ("bicycle", "books", "bottle", "camera", "cereal box", "chair", "cup", "laptop", "shoes", "sofa")
    cats = [{"id": i, "name": n} for i, n in zip((11, 14, 15, 16, 17, 18, 19, 20, 21, 40), names)]
    gt = Omni3DGroundTruth({"info": {}, "images": [{"id": 1, "height": 100}], "annotations": [], "categories": cats}, fs)
    cm = category_map_for(cfg, "base", gt)
    assert cm.contiguous_to_dataset_id == dict(enumerate((11, 14, 15, 16, 17, 18, 19, 20, 21))) and cm.thing_classes == list(names[:9])
    meta = tmp_path / "category_meta.json"
    meta.write_text(json.dumps({"thing_classes": ["chair", "sofa"], "thing_dataset_id_to_contiguous_id": {"18": 0, "40": 1}}))
    assert category_map_for(cfg, "base", gt, str(meta)).contiguous_to_dataset_id == {0: 18, 1: 40}
