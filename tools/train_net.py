#!/usr/bin/env python3
"""Drop-in for ``tools/train_net.py --eval-only`` of the reference (tools/train_net.py:359-459,561-570) on the
native path: one process per GPU, contiguous image shards per rank, the per-rank loop of
``inference_on_dataset`` and one gather of detection records to rank 0, predictions saved as JSON per
dataset (the reference's ``eval_helper.save_predictions``, tools/train_net.py:108-109).

Launch: ``python tools/train_net.py --eval-only --config-file configs/OVMono3D_dinov2_SFP.yaml --num-gpus N
MODEL.WEIGHTS <ckpt> OUTPUT_DIR <dir>`` (N > 1 re-launches itself under torch.distributed.run, one rank
per GPU over RCCL). Rank 0 then runs the Omni3D AP evaluator under the reference's evaluation rules (``evaluation/omni3d_gt.py``:
filter settings of its do_test, the ignore rule, ground truth in dataset category ids, detections un-mapped from the model's
class index; ``evaluation/omni3d_eval.py``: AP2D / AP3D with true 3D IoU and the disentangled NHD), written to ``omni_ap.json``
next to the ``category_meta.json`` that was applied. Training (``do_train``) is out of scope.

Deviations from the fork, both restoring upstream intent: oracle-2D boxes are
forwarded to the model when TEST.ORACLE2D is set and the oracle file exists, and TEST.CAT_MODE selects the mode.
"""
import argparse
import json
import logging
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ovmono3d_amd.checkpoint import DetectionCheckpointer  # noqa: E402
from ovmono3d_amd.data import (DatasetMapper3D, build_detection_test_loader, load_omni3d_json,  # noqa: E402
                               merge_oracle2d_to_detection_dicts)
from ovmono3d_amd.defaults import make_cfg  # noqa: E402
from ovmono3d_amd.evaluation import (CategoryMap, Omni3DEvaluator, Omni3DGroundTruth, collective_summary,  # noqa: E402
                                     eval_filter_settings, evaluate_omni3d, inference_on_dataset)
from ovmono3d_amd.evaluation.distributed import get_rank, get_world_size  # noqa: E402
from ovmono3d_amd.modeling import build_model  # noqa: E402

logger = logging.getLogger("cubercnn")


def oracle2d_file(cfg, mode, name, datasets_root):
    """cfg.DATASETS.ORACLE2D_FILES[EVAL_MODE][mode][dataset] (reference config.py:42-76, consumed at
    cubercnn/data/build.py:45-54): the configured path as given, else its base name under --datasets-root."""
    files = cfg.DATASETS.ORACLE2D_FILES.get(cfg.DATASETS.ORACLE2D_FILES.EVAL_MODE, None)
    split = files.get(mode, None) if files is not None and hasattr(files, "get") else None
    path = split.get(name, None) if split is not None and hasattr(split, "get") else None
    if not path:
        return None
    for cand in (path, os.path.join(datasets_root, os.path.basename(path))):
        if os.path.exists(cand):
            return cand
    return None


def do_test(cfg, model, mode="base", datasets_root="datasets/Omni3D", image_root="datasets", depth_dir=None, category_meta=None):
    if mode == "novel":
        names = cfg.DATASETS.TEST_NOVEL
    elif mode == "base":
        names = cfg.DATASETS.TEST_BASE
    else:
        raise ValueError("wrong mode")
    out_dir = os.path.join(cfg.OUTPUT_DIR, "inference", "iter_final")
    all_files, all_dets = [], []
    for name in names:
        dicts = load_omni3d_json(os.path.join(datasets_root, name + ".json"), image_root)
        if cfg.TEST.ORACLE2D:
            path = oracle2d_file(cfg, mode, name, datasets_root)
            if path:
                merge_oracle2d_to_detection_dicts(dicts, path)
            else:
                logger.warning("%s: TEST.ORACLE2D is set but no oracle-2D file was found (DATASETS.ORACLE2D_FILES.%s.%s); "
                               "running the RPN + box-head path", name, cfg.DATASETS.ORACLE2D_FILES.EVAL_MODE, mode)
        loader = build_detection_test_loader(cfg, dicts, DatasetMapper3D(cfg, False, depth_dir), get_rank(), get_world_size())
        results = inference_on_dataset(model, loader, Omni3DEvaluator(name, out_dir))
        if get_rank() == 0:
            os.makedirs(os.path.join(out_dir, name), exist_ok=True)
            with open(os.path.join(out_dir, name, "omni_instances_results.json"), "w") as f:
                json.dump([inst for r in results for inst in r["instances"]], f)
            logger.info("%s: %d images, %d detections", name, len(results), sum(len(r["instances"]) for r in results))
            # AP2D / AP3D (true 3D IoU on the device) under the reference's evaluation rules: the filter settings of its
            # do_test (tools/train_net.py:59-70), ground truth in DATASET category ids, detections un-mapped from the model's
            # contiguous class index (omni3d_evaluation.py:1029-1093)
            gt = Omni3DGroundTruth(os.path.join(datasets_root, name + ".json"), eval_filter_settings(cfg, mode))
            all_files.append(os.path.join(datasets_root, name + ".json"))
            all_dets += [inst for r in results for inst in r["instances"]]
            if len(gt):
                cmap = category_map_for(cfg, mode, gt, category_meta)
                ap = evaluate_omni3d(gt, [inst for r in results for inst in r["instances"]], category_map=cmap)
                with open(os.path.join(out_dir, name, "omni_ap.json"), "w") as f:
                    json.dump(ap, f)
                with open(os.path.join(out_dir, name, "category_meta.json"), "w") as f:
                    json.dump(cmap.to_meta(), f)
                logger.info("%s: AP2D %.2f  AP3D %.2f  (AP3D@15 %.2f, @25 %.2f, @50 %.2f)  NHD %.4f", name, ap["bbox_2D"]["AP"],
                            ap["bbox_3D"]["AP"], ap["bbox_3D"]["AP15"], ap["bbox_3D"]["AP25"], ap["bbox_3D"]["AP50"], ap["bbox_3D"]["NHD"])


    if get_rank() == 0 and len(all_files) > 1:
        # the collective numbers of the reference's eval_helper.summarize_all (omni3d_evaluation.py:427-620): AP / AR over the union of the
        # datasets' images, per category, and averaged over the Omni3D outdoor / indoor / all-50 category groups
        gt = Omni3DGroundTruth(all_files, eval_filter_settings(cfg, mode))
        if len(gt):
            ap = evaluate_omni3d(gt, all_dets, category_map=category_map_for(cfg, mode, gt, category_meta))
            ap["collective"] = collective_summary(ap)
            with open(os.path.join(out_dir, "omni_ap_all.json"), "w") as f:
                json.dump(ap, f)
            logger.info("all %d datasets: %s", len(all_files), json.dumps(ap["collective"]))


def category_map_for(cfg, mode, gt, category_meta=None):
    """Which contiguous class index the model's detections carry. A ``category_meta.json``-style file when given
    (the reference hard-codes configs/category_objectron.json for *_test / *_novel datasets, omni3d_evaluation.py:1015-1020);
    else the evaluated category names of the mode ranked by their dataset id in the annotation file's own category table
    (the rule of datasets.py:294-320 - for Objectron_test this IS the Objectron map: ids 11,14..21 -> 0..8)."""
    if category_meta:
        return CategoryMap.from_meta(category_meta)
    # the evaluated names of the mode; an empty list means "every category of the file" and Omni3DGroundTruth has written that
    # list back into the settings (datasets.py:218-226)
    names = list(gt.filter_settings["category_names"]) if gt.filter_settings else [c["name"] for c in gt.all_categories]
    known = {c["name"] for c in gt.all_categories}
    return CategoryMap.from_names([n for n in names if n in known], gt.all_categories)


def main(args):
    logging.basicConfig(level=logging.INFO)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    cfg = make_cfg(args.config_file, args.opts)
    if not args.eval_only:
        raise SystemExit("only --eval-only is supported by the native inference path (training is out of scope)")
    model = build_model(cfg, device=torch.device("cuda", local_rank))
    if world > 1:
        # the record gather goes through libovm3d's own RCCL communicator (ovm_gather_records), as for a non-Python host
        from ovmono3d_amd.evaluation.distributed import NativeComm, set_native_comm
        set_native_comm(NativeComm.from_torch_distributed(torch.device("cuda", local_rank)))
    DetectionCheckpointer(model, save_dir=cfg.OUTPUT_DIR).resume_or_load(cfg.MODEL.WEIGHTS, resume=args.resume)
    if cfg.TEST.CAT_MODE == "all":
        do_test(cfg, model, "novel", args.datasets_root, args.image_root, args.depth_dir, args.category_meta)
        do_test(cfg, model, "base", args.datasets_root, args.image_root, args.depth_dir, args.category_meta)
    else:
        do_test(cfg, model, cfg.TEST.CAT_MODE, args.datasets_root, args.image_root, args.depth_dir, args.category_meta)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def default_argument_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--config-file", default="", metavar="FILE")
    p.add_argument("--resume", action="store_true")
    p.add_argument("--eval-only", action="store_true")
    p.add_argument("--num-gpus", type=int, default=1)
    p.add_argument("--num-machines", type=int, default=1)
    p.add_argument("--machine-rank", type=int, default=0)
    p.add_argument("--dist-url", default="tcp://127.0.0.1:29500")
    p.add_argument("--datasets-root", default="datasets/Omni3D", help="(native build) folder of the Omni3D JSON files")
    p.add_argument("--image-root", default="datasets")
    p.add_argument("--depth-dir", default=None, help="(native build) folder of depth-prompt .npz files")
    p.add_argument("--category-meta", default=None, help="(native build) category_meta.json-style file: thing_classes + "
                   "thing_dataset_id_to_contiguous_id of the model's class index; default: derived from the annotation file")
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    return p


if __name__ == "__main__":
    args = default_argument_parser().parse_args()
    if args.num_gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the reference's launch() spawns one process per GPU (tools/train_net.py:563-570); here the same via
        # torch.distributed.run, started as a child process before anything touches the GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.num_gpus}",
               "--master-addr", "127.0.0.1", "--master-port", "29511", os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    main(args)
