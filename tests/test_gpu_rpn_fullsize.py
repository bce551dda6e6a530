"""The RPN -> box head -> Fast R-CNN inference -> cube head route at its REAL size (VERDICT r2 missing #1): the route the fork's
`--eval-only` takes when no oracle boxes are merged in (reference cubercnn/data/build.py:281-311, rcnn3d.py:105-111) and the only one with
a published number (nohup.out:939-940, ViT-B). Canvas 896: p2 / p3 / p4 = 128 / 64 / 32 -> 3 x (16384 + 4096 + 1024) = 64,512 anchors,
top-1000 per level, NMS 0.7, 1000 proposals, fc1 with M = 1000, 50-class NMS at 0.5, <= 100 detections - against oracle/rpn.py +
oracle/heads.py::forward_box (+ forward_cube), through the C ABI (ovm_rpn_box_forward, ovm_cube_forward).

Discrete decisions sit between the float tensors (three top-k, two NMS, a 0.01 threshold on softmax scores): where two scores agree
to ~1e-6 the HIP and CPU summation orders may rank them differently. The test therefore pairs by box, bounds the number of flips and
prints them; everything paired is held to 1e-3 with exact class ids. Parity against the reference itself is unpinned."""
import pytest
import torch

from common import build_cfg, oracle_params, synth_inputs
from parity import parity_ok, parity_report

pytestmark = pytest.mark.gpu


def _pair_boxes(a, b, tol_px):
    """greedy one-to-one pairing of two box sets by max corner distance; returns (#paired, max distance among pairs)."""
    d = (a[:, None, :] - b[None, :, :]).abs().amax(-1)
    used = torch.zeros(b.shape[0], dtype=torch.bool)
    n, worst = 0, 0.0
    for i in torch.argsort(d.min(1)[0]).tolist():
        row = d[i].clone()
        row[used] = float("inf")
        j = int(torch.argmin(row))
        if row[j] <= tol_px:
            used[j] = True
            n += 1
            worst = max(worst, float(row[j]))
    return n, worst


@pytest.mark.parametrize("arch,seed,hw", [("vitb14", 2, (532, 709)), ("vitl14", 0, (532, 532))])
def test_rpn_boxhead_fullsize_canvas896(device, arch, seed, hw):
    from oracle.pipeline import inference
    from ovmono3d_amd.modeling import build_model
    from ovmono3d_amd.util.synth_weights import synth_state_dict
    cfg = build_cfg(arch, 896, "f16x3", max_batch=1, max_rois=1000)     # configs/OVMono3D_dinov2_SFP.yaml:30 names vitb14
    sd = synth_state_dict(arch, num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES, seed=seed)
    model = build_model(cfg)
    model.load_state_dict(sd)
    inputs = synth_inputs(1, hw=(hw,), orig_scale=480.0 / 532.0, n_boxes=0, seed=91, oracle2d=False)
    out = model(inputs)[0]["instances"]
    eng = model.engine
    R = min(cfg.MODEL.RPN.POST_NMS_TOPK_TEST, cfg.MODEL.AMD.MAX_ROIS)
    n_prop = int(eng.debug_tensor("rpn_counts", 1).view(torch.int32)[0])
    got_prop = eng.debug_tensor("rpn_boxes", R * 4).view(R, 4)[:n_prop].cpu()
    got_logit = eng.debug_tensor("rpn_scores", R)[:n_prop].cpu()
    torch.set_num_threads(16)
    ref, aux = inference(sd, inputs, oracle_params(cfg), return_aux=True)
    ref = ref[0]
    rp_box, rp_logit = aux["proposals"][0]
    # ---- proposals: 64,512 anchors -> per-level top-1000 -> NMS 0.7 -> top 1000 by objectness
    assert sum(int(f.shape[-1]) ** 2 * 3 for f in aux["features"].values()) == 64512
    assert 500 <= len(rp_box) <= cfg.MODEL.RPN.POST_NMS_TOPK_TEST == 1000, len(rp_box)      # NMS 0.7 may leave fewer than 1000 on a small image
    n_pair, worst_px = _pair_boxes(rp_box, got_prop, 1e-2)
    flips = max(len(rp_box), n_prop) - n_pair
    same_order = n_prop == len(rp_box) and bool(((got_prop - rp_box).abs().amax(1) <= 1e-2).all())
    print(f"{arch}: proposals HIP {n_prop} / oracle {len(rp_box)}, paired {n_pair} (worst {worst_px:.1e} px), flipped {flips}, identical order {same_order}")
    assert abs(n_prop - len(rp_box)) <= 2 and flips <= 10, "more than 1 % of the proposals differ between the HIP path and the oracle"
    if same_order:
        assert float((got_logit - rp_logit).abs().max()) <= 1e-3 * float(rp_logit.abs().max())
    # ---- detections (<= 100 after the 0.01 threshold and the per-class NMS at 0.5) and their cubes
    rep = parity_report(out, ref)
    print(f"{arch}: detections", {k: v for k, v in rep.items() if k != "pose"}, "pose:", rep.get("pose"))
    assert 0 < rep["n_det_oracle"] <= cfg.TEST.DETECTIONS_PER_IMAGE
    lim = max(2, rep["n_det_oracle"] // 50)
    assert rep["unmatched_oracle"] <= lim and rep["unmatched_hip"] <= lim, rep
    assert rep["class_id_mismatches"] == 0, rep
    assert parity_ok(dict(rep, unmatched_oracle=0, unmatched_hip=0), 1e-3, pose_by_conditioning=True), rep
    # ---- no discrete decision in between: the oracle's 2D detections through the HIP cube branch on the HIP features, identity pairing
    from ovmono3d_amd.structures import Boxes, Instances
    b2 = aux["instances_2d"][0]
    images = model.preprocess_image(inputs)
    model.backbone(images)
    t = Instances(images.image_sizes[0])
    t.pred_boxes, t.scores, t.pred_classes = Boxes(b2["pred_boxes"].to(device)), b2["scores"].to(device), b2["pred_classes"].to(device)
    got_b = model.roi_heads._forward_cube(None, [t], None, list(images.image_sizes), [inputs[0]["height"] / images.image_sizes[0][0]],
                                          images=images, postprocess=True)[0]
    rep_b = parity_report(got_b, ref, box_tol=1e-4)
    print(f"{arch}: same boxes through the HIP cube branch", {k: v for k, v in rep_b.items() if k != "pose"})
    assert rep_b["same_order"] and parity_ok(rep_b, 1e-3), rep_b
