#!/usr/bin/env python3
"""Drop-in for the reference's demo/demo.py (same flags, reference demo/demo.py:157-188) on the native path.

Per image: read -> K from the focal-length convention of the reference (:63-76, including its quirk that the
first image's focal length sticks) -> ResizeShortestEdge -> model([{image, height,
width, K, category_list}]) -> threshold on score (:99). Drawing (vis.draw_scene_view, :107-118) is
presentation and out of scope: detections are written as JSON next to where the reference writes its JPEGs.

2D boxes: the reference's demo always goes through ROIHeads3DGDINO (category_list is set, :84). Select it with
``MODEL.ROI_HEADS.NAME ROIHeads3DGDINO``: the native GroundingDINO network (``ovmono3d_amd/gdino``) is built from
``MODEL.AMD.GDINO_WEIGHTS`` (+ ``MODEL.AMD.BERT_VOCAB``) on first use and raises if the checkpoint is missing. Alternatives:
``--boxes-file`` with oracle-2D boxes ({image name: [{bbox xywh, category_id, score}]}), or the default
``MODEL.ROI_HEADS.NAME ROIHeads3D`` = the RPN + box-head path.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ovmono3d_amd.checkpoint import DetectionCheckpointer  # noqa: E402
from ovmono3d_amd.data import ResizeShortestEdge, read_image  # noqa: E402
from ovmono3d_amd.defaults import make_cfg  # noqa: E402
from ovmono3d_amd.modeling import build_model  # noqa: E402


def do_test(args, cfg, model):
    ims = sorted(f for f in os.listdir(args.input_folder) if not f.endswith(".json"))
    with open(args.labels_file) as f:
        cats_per_img = json.load(f)
    boxes_per_img = None
    if args.boxes_file:
        with open(args.boxes_file) as f:
            boxes_per_img = json.load(f)
    model.eval()
    focal_length = args.focal_length
    principal_point = args.principal_point
    thres = args.threshold
    os.makedirs(cfg.OUTPUT_DIR, exist_ok=True)
    resize = ResizeShortestEdge(cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
    gpu_resize = None
    if bool(cfg.MODEL.AMD.get("GPU_RESIZE", False)):
        from ovmono3d_amd.data.gpu_resize import ResizeShortestEdgeGPU
        gpu_resize = ResizeShortestEdgeGPU(cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
    gpu_jpeg = gpu_resize is not None and bool(cfg.MODEL.AMD.get("GPU_JPEG", False))
    use_gdino_head = cfg.MODEL.ROI_HEADS.NAME == "ROIHeads3DGDINO"
    for name in ims:
        im_name = os.path.splitext(name)[0]
        cats = cats_per_img.get(im_name, [])
        if cats == []:
            continue
        if gpu_jpeg:                                                         # baseline JPEG: host entropy decode, pixels born on the device
            from ovmono3d_amd.data.gpu_jpeg import read_image_device
            im = read_image_device(os.path.join(args.input_folder, name), "BGR", torch.device("cuda"))
        else:
            im = read_image(os.path.join(args.input_folder, name), "BGR")   # cv2.imread order, demo.py:52
        h, w = im.shape[:2]
        if focal_length == 0:
            focal_length = 4.0 * h / 2                                       # demo.py:63-65
        px, py = (w / 2, h / 2) if len(principal_point) == 0 else principal_point
        K = np.array([[focal_length, 0.0, px], [0.0, focal_length, py], [0.0, 0.0, 1.0]])
        if gpu_resize is not None:                                           # device-side ResizeShortestEdge, bit-identical to Pillow's
            image_t = gpu_resize(im if torch.is_tensor(im) else torch.from_numpy(np.ascontiguousarray(im)).cuda()).permute(2, 0, 1)
        else:
            image_t = torch.as_tensor(np.ascontiguousarray(resize(im).transpose(2, 0, 1)))
        d = {"image": image_t, "height": h, "width": w, "K": K}
        if boxes_per_img is not None:
            inst = boxes_per_img.get(im_name, [])
            d["oracle2D"] = {"gt_bbox2D": torch.tensor([[b["bbox"][0], b["bbox"][1], b["bbox"][0] + b["bbox"][2],
                                                        b["bbox"][1] + b["bbox"][3]] for b in inst], dtype=torch.float32).reshape(-1, 4),
                             "gt_classes": torch.tensor([b["category_id"] for b in inst], dtype=torch.int64),
                             "gt_scores": torch.tensor([b.get("score", 1.0) for b in inst], dtype=torch.float32)}
        elif use_gdino_head:
            d["category_list"] = cats
        dets = model([d])[0]["instances"]
        n_det = len(dets) if dets.get_fields() else 0
        out = []
        if n_det > 0 and dets.has("pred_bbox3D"):
            for idx in range(n_det):
                score = float(dets.scores[idx])
                if score < thres:
                    continue
                ci = int(dets.pred_classes[idx])
                out.append({"category": cats[ci] if ci < len(cats) else ci, "score": score,
                            "bbox3D": dets.pred_center_cam[idx].tolist() + dets.pred_dimensions[idx].tolist(),
                            "pose": dets.pred_pose[idx].tolist(), "corners3D": dets.pred_bbox3D[idx].tolist(),
                            "center_2D": dets.pred_center_2D[idx].tolist(), "bbox2D": dets.pred_boxes.tensor[idx].tolist()})
        print("File: {} with {} dets".format(im_name, len(out)))
        with open(os.path.join(cfg.OUTPUT_DIR, im_name + "_dets.json"), "w") as f:
            json.dump({"K": K.tolist(), "detections": out}, f)


def setup(args):
    return make_cfg(args.config_file, args.opts)


def main(args):
    cfg = setup(args)
    model = build_model(cfg)
    DetectionCheckpointer(model, save_dir=cfg.OUTPUT_DIR).resume_or_load(cfg.MODEL.WEIGHTS, resume=True)
    with torch.no_grad():
        do_test(args, cfg, model)


if __name__ == "__main__":
    parser = argparse.ArgumentParser(epilog=None, formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("--config-file", default="", metavar="FILE", help="path to config file")
    parser.add_argument("--input-folder", type=str, help="list of image folders to process", required=True)
    parser.add_argument("--labels-file", type=str, help="path to labels file", required=True)
    parser.add_argument("--boxes-file", type=str, default="", help="(native build) oracle-2D boxes per image")
    parser.add_argument("--focal-length", type=float, default=0, help="focal length for image inputs (in px)")
    parser.add_argument("--principal-point", type=float, default=[], nargs=2, help="principal point for image inputs (in px)")
    parser.add_argument("--threshold", type=float, default=0.25, help="threshold on score for visualizing")
    parser.add_argument("--display", default=False, action="store_true", help="unused (no drawing in the native build)")
    parser.add_argument("--eval-only", default=True, action="store_true", help="perform evaluation only")
    parser.add_argument("--num-gpus", type=int, default=1, help="number of gpus *per machine*")
    parser.add_argument("--num-machines", type=int, default=1, help="total number of machines")
    parser.add_argument("--machine-rank", type=int, default=0, help="the rank of this machine (unique per machine)")
    parser.add_argument("--dist-url", default="tcp://127.0.0.1:29500")
    parser.add_argument("opts", default=None, nargs=argparse.REMAINDER,
                        help="Modify config options by adding 'KEY VALUE' pairs at the end of the command.")
    args = parser.parse_args()
    print("Command Line Args:", args)
    main(args)
