#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the CPU oracle.

The reference holds no golden vectors, result files or value-asserting tests for this path, and its
code cannot be imported in the build container, so these fixtures are produced by
the fp32 restatement in oracle/ (itself cross-checked in tests/test_oracle_crosscheck.py). They pin
(1) the oracle against drift and (2) the HIP path on the GPU box, where the oracle also runs.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from common import build_cfg, oracle_params, synth_inputs  # noqa: E402
from oracle.pipeline import inference  # noqa: E402
from ovmono3d_amd.util.synth_weights import synth_state_dict  # noqa: E402

FIELDS = ("pred_boxes", "scores", "pred_classes", "pred_bbox3D", "pred_center_cam", "pred_center_2D",
          "pred_dimensions", "pred_pose")


def weight_fingerprint(sd):
    keys = sorted(sd)[:: max(1, len(sd) // 16)]
    return np.array([float(sd[k].double().sum()) for k in keys])


def pack_inputs(inputs, out, prefix):
    for i, d in enumerate(inputs):
        out[f"{prefix}img{i}"] = d["image"].numpy()
        out[f"{prefix}meta{i}"] = np.array([d["height"], d["width"]], dtype=np.int64)
        out[f"{prefix}K{i}"] = np.asarray(d["K"], dtype=np.float32)
        if "oracle2D" in d:
            out[f"{prefix}box{i}"] = d["oracle2D"]["gt_bbox2D"].numpy()
            out[f"{prefix}cls{i}"] = d["oracle2D"]["gt_classes"].numpy()
            out[f"{prefix}sc{i}"] = d["oracle2D"]["gt_scores"].numpy()
        if "depth" in d:
            out[f"{prefix}depth{i}"] = d["depth"].numpy()


def main():
    torch.manual_seed(0)
    # ---- case 1: oracle-2D, two images of different size, tiny ViT, canvas 224 ----
    cfg = build_cfg("vittest14", 224, "f16x3", max_batch=2)
    sd = synth_state_dict("vittest14", seed=21)
    inputs = synth_inputs(2, hw=((140, 196), (224, 168)), n_boxes=10, seed=22)
    res, aux = inference(sd, inputs, oracle_params(cfg), return_aux=True)
    out = {"weights_seed": np.array(21), "weights_fp": weight_fingerprint(sd), "n_images": np.array(2)}
    pack_inputs(inputs, out, "in_")
    for i, r in enumerate(res):
        for f in FIELDS:
            out[f"out{i}_{f}"] = r[f].numpy()
    for k in ("p2", "p3", "p4"):
        out[f"feat_{k}_corner"] = aux["features"][k][:, :8, :6, :6].numpy()
    np.savez_compressed(os.path.join(HERE, "e2e_oracle2d_vittest14.npz"), **out)

    # ---- case 2: depth prompt ----
    inputs = synth_inputs(1, hw=((150, 200),), n_boxes=6, seed=23, depth=True)
    depth = torch.stack([x["depth"] for x in inputs])
    res = inference(sd, inputs, oracle_params(cfg), prompt_depth=depth)
    out = {"weights_seed": np.array(21), "weights_fp": weight_fingerprint(sd), "n_images": np.array(1)}
    pack_inputs(inputs, out, "in_")
    for f in FIELDS:
        out[f"out0_{f}"] = res[0][f].numpy()
    np.savez_compressed(os.path.join(HERE, "e2e_depth_vittest14.npz"), **out)

    # ---- case 3: RPN + box head + cube head end to end (no oracle boxes) ----
    inputs = synth_inputs(1, hw=((168, 210),), n_boxes=0, seed=24, oracle2d=False)
    res, aux = inference(sd, inputs, oracle_params(cfg), return_aux=True)
    out = {"weights_seed": np.array(21), "weights_fp": weight_fingerprint(sd), "n_images": np.array(1)}
    pack_inputs(inputs, out, "in_")
    for f in FIELDS:
        out[f"out0_{f}"] = res[0][f].numpy()
    out["proposal_boxes"] = aux["proposals"][0][0].numpy()
    out["proposal_logits"] = aux["proposals"][0][1].numpy()
    np.savez_compressed(os.path.join(HERE, "e2e_rpn_vittest14.npz"), **out)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
