#!/usr/bin/env python3
"""Offline producer of the oracle-2D files the eval configs read ("next" row 4 of SURVEY.md 8f).

Runs the text-prompted 2D detector (native GroundingDINO + the reference's phrase-logit glue) over an Omni3D annotation file and
writes the on-disk format ``merge_oracle2d_to_detection_dicts`` consumes (reference cubercnn/data/build.py:45-54; file names in
cubercnn/config/config.py:69-76, README.md:70-74): a list aligned with the dataset order of
``{"image_id": int, "instances": [{"bbox": [x, y, w, h] (original resolution), "category_id": int, "score": float}]}``.

  python tools/make_oracle2d.py --config-file configs/OVMono3D_dinov2_SFP.yaml --dataset datasets/Omni3D/SUNRGBD_test.json \\
      --image-root datasets --output datasets/Omni3D/gdino_sunrgbd_novel_oracle_2d.json \\
      MODEL.AMD.GDINO_WEIGHTS checkpoints/groundingdino_swinb_cogcoor.pth MODEL.AMD.BERT_VOCAB bert-base-uncased/vocab.txt

Categories prompted per image: ``--categories a,b,c`` (one list for the whole file) or, by default, the ``categories`` of the
annotation file in id order. ``category_id`` in the output is what the cube head passes on as ``pred_classes``
(reference roi_heads_gdino.py:108-121), i.e. the model's CONTIGUOUS class index: the rank of the matched category among the
prompted ones ordered by dataset id (datasets.py:294-320) - the id space the evaluator un-maps (tools/train_net.py here,
omni3d_evaluation.py:1029-1093 there). ``--dataset-ids`` writes the annotation file's own category ids instead.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ovmono3d_amd.data.feeding import ResizeShortestEdge, load_omni3d_json, read_image  # noqa: E402
from ovmono3d_amd.defaults import make_cfg  # noqa: E402
from ovmono3d_amd.modeling.roi_heads.gdino_glue import build_caption, gdino_postprocess, phrase_spans  # noqa: E402


def build_detector(cfg, device):
    from ovmono3d_amd.checkpoint import load_state_dict_file
    from ovmono3d_amd.gdino.detector import HashTokenizer, NativeGroundingDino
    from ovmono3d_amd.modeling.roi_heads.gdino_glue import WordPieceTokenizer
    path = cfg.MODEL.AMD.GDINO_WEIGHTS
    if path.startswith("synthetic://"):
        from ovmono3d_amd.util.synth_gdino_weights import synth_gdino_state_dict
        sd, tok = synth_gdino_state_dict(int(path.split("seed=")[1]) if "seed=" in path else 0), HashTokenizer()
    else:
        sd, tok = load_state_dict_file(path), WordPieceTokenizer(cfg.MODEL.AMD.BERT_VOCAB)
    return NativeGroundingDino(device, sd, tok, cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD,
                               precision=3 if cfg.MODEL.AMD.GEMM_PRECISION == "f16x3" else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-file", required=True)
    ap.add_argument("--dataset", required=True, help="Omni3D annotation json")
    ap.add_argument("--image-root", default="datasets")
    ap.add_argument("--output", required=True)
    ap.add_argument("--categories", default="", help="comma separated prompt list (default: the file's categories)")
    ap.add_argument("--dataset-ids", action="store_true", help="write the annotation file's category ids, not class indices")
    ap.add_argument("--box-threshold", type=float, default=0.001)
    ap.add_argument("--nms-threshold", type=float, default=0.5)
    ap.add_argument("opts", nargs=argparse.REMAINDER)
    args = ap.parse_args()
    cfg = make_cfg(args.config_file, args.opts)
    device = torch.device("cuda", torch.cuda.current_device())
    det = build_detector(cfg, device)
    with open(args.dataset) as f:
        meta = json.load(f)
    if args.categories:
        names = [c.strip() for c in args.categories.split(",") if c.strip()]
        ids = list(range(len(names)))
    else:
        cats = sorted(meta.get("categories", []), key=lambda c: c["id"])
        names = [c["name"] for c in cats]
        ids = [c["id"] for c in cats] if args.dataset_ids else list(range(len(cats)))
    if not names:
        raise SystemExit("no categories to prompt: pass --categories or use an annotation file with a categories section")
    caption, cap_list = build_caption(names)
    resize = ResizeShortestEdge(cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
    out = []
    for d in load_omni3d_json(args.dataset, args.image_root):
        im = read_image(d["file_name"], cfg.INPUT.FORMAT)
        net = resize(im)
        image = torch.as_tensor(np.ascontiguousarray(net.transpose(2, 0, 1))).to(device)
        r = det(image, caption)
        boxes, scores, cls = gdino_postprocess(r["pred_logits"], r["pred_boxes"], phrase_spans(r["input_ids"], r["phrase_ids"]), net.shape[:2],
                                               box_threshold=args.box_threshold, nms_threshold=args.nms_threshold)
        sy, sx = d["height"] / net.shape[0], d["width"] / net.shape[1]                     # network resolution -> original resolution
        b = boxes.cpu().numpy() * np.array([sx, sy, sx, sy], np.float32)
        inst = [{"bbox": [float(x1), float(y1), float(x2 - x1), float(y2 - y1)], "category_id": int(ids[int(c)]), "score": float(s)}
                for (x1, y1, x2, y2), s, c in zip(b, scores.cpu().tolist(), cls.cpu().tolist())]
        out.append({"image_id": d["image_id"], "instances": inst})
    os.makedirs(os.path.dirname(os.path.abspath(args.output)), exist_ok=True)
    with open(args.output, "w") as f:
        json.dump(out, f)
    print(f"wrote {args.output}: {len(out)} images, {sum(len(o['instances']) for o in out)} boxes")


if __name__ == "__main__":
    main()
