"""End-to-end GPU parity of the native path (through the plugin surface and the C ABI) against the
CPU oracle on seeded synthetic checkpoints.

Tolerance (BASELINE.json north_star): 1e-3 relative on float tensors, measured as
max|a-b| / max|b| per output tensor; category indices bit-exact. Default precision f16x3.
Parity is "unpinned" against the reference itself (no reference fixtures exist): the
oracle is the fp32 restatement in oracle/."""
import pytest
import torch

from common import assert_close, build_cfg, build_clip_cfg, build_mae_cfg, build_midas_cfg, build_sam_cfg, oracle_params, synth_inputs

pytestmark = pytest.mark.gpu

TOL = 1e-3
FIELDS = ("pred_boxes", "scores", "pred_bbox3D", "pred_center_cam", "pred_center_2D", "pred_dimensions", "pred_pose")


def _build(cfg, seed=1):
    from ovmono3d_amd.modeling import build_model
    from ovmono3d_amd.util.synth_weights import synth_state_dict
    name = {"build_clip_backbone": cfg.MODEL.CLIP.ARCH, "build_mae_backbone": cfg.MODEL.MAE.CHECKPOINT,
            "build_midas_backbone": cfg.MODEL.MIDAS.ARCH, "build_sam_backbone": cfg.MODEL.SAM.ARCH}.get(cfg.MODEL.BACKBONE.NAME,
                                                                                                           cfg.MODEL.DINO.MODEL_NAME)
    sd = synth_state_dict(name, num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES, seed=seed)
    model = build_model(cfg)
    model.load_state_dict(sd)
    return model, sd


def _compare(out, ref, tol=TOL):
    assert len(out) == len(ref)
    for o, r in zip(out, ref):
        inst = o["instances"]
        assert len(inst) == len(r["scores"])
        assert torch.equal(inst.pred_classes.cpu(), r["pred_classes"].to(torch.int64)), "category indices differ"
        for f in FIELDS:
            got = inst.get(f)
            got = got.tensor if hasattr(got, "tensor") else got
            if r[f].numel():
                assert_close(got, r[f], tol, f)


@pytest.mark.parametrize("precision", ["f16x3"])
def test_oracle2d_tiny_vit(device, precision):
    from oracle.pipeline import inference
    cfg = build_cfg("vittest14", 224, precision, max_batch=2)
    model, sd = _build(cfg)
    inputs = synth_inputs(2, hw=((140, 196), (224, 168)), n_boxes=12, seed=3)
    out = model(inputs)
    ref, aux = inference(sd, inputs, oracle_params(cfg), return_aux=True)
    # intermediate: pyramid levels (NHWC storage exposed as NCHW views)
    model.backbone.export_features = True
    feats = model.backbone(model.preprocess_image(inputs))
    for k in ("p2", "p3", "p4"):
        assert_close(feats[k], aux["features"][k], 2e-4, k)
    _compare(out, ref)


def test_corun_mode_is_scheduling_only(device):
    """ovm_set_corun (4-wave attention workgroups held to one per CU, for running beside the GroundingDINO stream) must not
    change results beyond the fp32 summation order of the leftover (cls) query's dot-product path."""
    cfg = build_cfg("vittest14", 224, "f16x3", max_batch=2)
    model, sd = _build(cfg)
    inputs = synth_inputs(2, hw=((140, 196), (224, 168)), n_boxes=12, seed=3)
    model.backbone.export_features = True
    base = {k: v.clone() for k, v in model.backbone(model.preprocess_image(inputs)).items()}
    model.engine.set_corun(True)
    try:
        got = model.backbone(model.preprocess_image(inputs))
        for k in ("p2", "p3", "p4"):
            assert_close(got[k], base[k], 1e-5, f"corun {k}")
    finally:
        model.engine.set_corun(False)


def test_oracle2d_vitb_canvas518(device):
    from oracle.pipeline import inference
    cfg = build_cfg("vitb14", 518, "f16x3", max_batch=1)
    model, sd = _build(cfg, seed=2)
    inputs = synth_inputs(1, hw=((512, 384),), orig_scale=1.25, n_boxes=32, seed=4)
    out = model(inputs)
    ref = inference(sd, inputs, oracle_params(cfg))
    _compare(out, ref)


def test_depth_prompt(device):
    from oracle.pipeline import inference
    cfg = build_cfg("vittest14", 224, "f16x3", max_batch=2)
    model, sd = _build(cfg, seed=5)
    inputs = synth_inputs(2, hw=((150, 200),), n_boxes=6, seed=6, depth=True)
    depth = torch.stack([x["depth"] for x in inputs])           # omni3d_evaluation.py:661-665
    out = model(inputs, prompt_depth=depth)
    ref = inference(sd, inputs, oracle_params(cfg), prompt_depth=depth)
    _compare(out, ref)
    out2 = model(inputs)                                         # and differs from the no-depth result
    assert (out2[0]["instances"].pred_center_cam - out[0]["instances"].pred_center_cam).abs().max() > 0


@pytest.mark.parametrize("tower", ["dinov2", "clip", "mae", "midas", "sam"])
def test_fast_precision_runs(device, tower):
    """One-pass fp16 mode on every tower: same categories / box set, floats within the looser documented band."""
    from oracle.pipeline import inference
    cfg = {"dinov2": lambda: build_cfg("vittest14", 224, "f16", max_batch=1), "clip": lambda: build_clip_cfg("ViT-test-16", 256, "f16", max_batch=1),
           "mae": lambda: build_mae_cfg("test/vit-mae-test", 256, "f16", max_batch=1), "midas": lambda: build_midas_cfg("DPT_test", 256, "f16", max_batch=1),
           "sam": lambda: build_sam_cfg("vit_test", 256, "f16", max_batch=1)}[tower]()
    model, sd = _build(cfg)
    inputs = synth_inputs(1, n_boxes=8, seed=7)
    out = model(inputs)
    ref = inference(sd, inputs, oracle_params(cfg))
    _compare(out, ref, tol=5e-2)


def test_empty_and_degenerate_boxes(device):
    cfg = build_cfg("vittest14", 224, "f16x3", max_batch=1)
    model, sd = _build(cfg)
    inputs = synth_inputs(1, n_boxes=4, seed=8)
    inputs[0]["oracle2D"] = {"gt_bbox2D": torch.zeros(0, 4), "gt_classes": torch.zeros(0, dtype=torch.int64)}
    out = model(inputs)
    inst = out[0]["instances"]
    assert len(inst) == 0 and not inst.has("pred_bbox3D")        # roi_heads.py:371-372: 3D fields absent
    inputs = synth_inputs(1, n_boxes=4, seed=8)
    inputs[0]["oracle2D"]["gt_bbox2D"][1] = torch.tensor([10.0, 10.0, 10.0, 40.0])   # empty after postprocess
    out = model(inputs)
    assert len(out[0]["instances"]) == 3


def test_rpn_boxhead_path_tiny_vit(device):
    """No oracle boxes -> RPN -> box head -> Fast R-CNN inference -> cube head (reference rcnn3d.py:105-111)."""
    from oracle.pipeline import inference
    cfg = build_cfg("vittest14", 224, "f16x3", max_batch=2)
    model, sd = _build(cfg, seed=21)
    inputs = synth_inputs(2, hw=((168, 210), (224, 224)), n_boxes=0, seed=24, oracle2d=False)
    out = model(inputs)
    ref = inference(sd, inputs, oracle_params(cfg))
    assert len(out[0]["instances"]) > 0
    _compare(out, ref)
    assert out[0]["instances"].has("pred_bbox3D")


def test_oracle2d_vitl_canvas896_headline_size(device):
    """The bench's ViT configuration at its own size: ViT-L/14 (24 layers, D = 1024), SQUARE_PAD 896 (T = 4097), one 532x532
    image, 32 given boxes, against the CPU oracle (rcnn3d.py:79-117). Measures the f16x3 error
    growth over 24 layers on the kernels themselves instead of arguing it from an emulation."""
    from oracle.pipeline import inference
    cfg = build_cfg("vitl14", 896, "f16x3", max_batch=1, max_rois=64)
    model, sd = _build(cfg, seed=0)
    inputs = synth_inputs(1, hw=((532, 532),), orig_scale=512.0 / 532.0, n_boxes=32, seed=11)
    out = model(inputs)
    torch.set_num_threads(16)
    ref, aux = inference(sd, inputs, oracle_params(cfg), return_aux=True)
    model.backbone.export_features = True
    feats = model.backbone(model.preprocess_image(inputs))
    for k in ("p2", "p3", "p4"):
        e = assert_close(feats[k], aux["features"][k], 1e-3, k)
        print(f"ViT-L/896 {k}: scale-relative error {e:.2e}")
    _compare(out, ref)


def test_config5_geometry_vitl_width_canvas1036_batch2(device):
    """BASELINE config 5 geometry: 1024x1024 inputs need SQUARE_PAD 1036 -> 74x74 patches,
    T = 5477 (= 21 x 256 + 101 queries: a partial last attention block), p2/p3/p4 = 148/74/37 (odd p4), batch 2, ViT-L width at
    depth 2, two image shapes, against the CPU oracle."""
    from oracle.pipeline import inference
    cfg = build_cfg("vitl14_d2", 1036, "f16x3", max_batch=2, max_rois=64)
    model, sd = _build(cfg, seed=4)
    inputs = synth_inputs(2, hw=((1024, 1024), (768, 1024)), orig_scale=1.0, n_boxes=24, seed=12)
    out = model(inputs)
    torch.set_num_threads(16)
    ref, aux = inference(sd, inputs, oracle_params(cfg), return_aux=True)
    model.backbone.export_features = True
    feats = model.backbone(model.preprocess_image(inputs))
    assert tuple(feats["p4"].shape[-2:]) == (37, 37) and tuple(feats["p2"].shape[-2:]) == (148, 148)
    for k in ("p2", "p3", "p4"):
        assert_close(feats[k], aux["features"][k], 2e-4, k)
    _compare(out, ref)


def test_inference_on_dataset_empty_image_between_nonempty(device):
    """An image without detections between two with detections, through the per-rank loop (reference
    omni3d_evaluation.py:626-734): the records of all three concatenate on the model's device and keep dataset order."""
    from ovmono3d_amd.evaluation import inference_on_dataset
    cfg = build_cfg("vittest14", 224, "f16x3", max_batch=1)
    model, sd = _build(cfg)
    inputs = synth_inputs(3, n_boxes=4, seed=8)
    inputs[1]["oracle2D"] = {"gt_bbox2D": torch.zeros(0, 4), "gt_classes": torch.zeros(0, dtype=torch.int64)}
    res = inference_on_dataset(model, [[d] for d in inputs])
    assert [len(r["instances"]) for r in res] == [4, 0, 4] and [r["image_id"] for r in res] == [0, 1, 2]
    solo = model([inputs[2]])[0]["instances"]
    assert abs(res[2]["instances"][0]["score"] - float(solo.scores[0])) < 1e-7


# ------------------------------------------------------------------------------------------ CLIP tower (BASELINE config 4)
def test_clip_tower_tiny_oracle2d_and_rpn_paths(device):
    """build_clip_backbone: open_clip-style tower (conv1 without bias, ln_pre, QuickGELU, LN eps 1e-5, antialiased pos-embed
    resize 7 -> 16) + the 4-level pyramid (scale-4 stage ConvT . LN . GELU . ConvT) + 4-level ROIPooler / RPN, against the CPU
    oracle (reference backbone/clip.py:62-166). Two image shapes; given boxes, then the RPN + box-head route."""
    from oracle.pipeline import inference
    cfg = build_clip_cfg("ViT-test-16", 256, "f16x3", max_batch=2)
    model, sd = _build(cfg, seed=5)
    inputs = synth_inputs(2, hw=((160, 224), (256, 192)), n_boxes=12, seed=31)
    out = model(inputs)
    ref, aux = inference(sd, inputs, oracle_params(cfg), return_aux=True)
    model.backbone.export_features = True
    feats = model.backbone(model.preprocess_image(inputs))
    assert sorted(feats) == ["p2", "p3", "p4", "p5"] and tuple(feats["p2"].shape[-2:]) == (64, 64) and tuple(feats["p5"].shape[-2:]) == (8, 8)
    for k in ("p2", "p3", "p4", "p5"):
        assert_close(feats[k], aux["features"][k], 2e-4, k)
    model.backbone.export_features = False
    _compare(out, ref)
    # RPN over four levels -> box head -> cube head
    inputs2 = synth_inputs(2, hw=((192, 256), (256, 256)), n_boxes=0, seed=33, oracle2d=False)
    out2 = model(inputs2)
    ref2 = inference(sd, inputs2, oracle_params(cfg))
    assert len(out2[0]["instances"]) > 0
    _compare(out2, ref2)


def test_backbone_level_c_abi(device):
    """ovm_backbone_num_levels / ovm_backbone_level hand out the pyramid the handle holds (how a non-Python host reads p5 of the
    4-level towers): sides, strides and contents equal the exported copies; bad arguments are refused."""
    import ctypes as C
    for cfg, n in ((build_cfg("vittest14", 224, "f16x3", max_batch=1), 3), (build_clip_cfg("ViT-test-16", 256, "f16x3", max_batch=1), 4)):
        model, sd = _build(cfg, seed=5)
        inputs = synth_inputs(1, hw=((160, 224),), n_boxes=4, seed=31)
        model.backbone.export_features = True
        feats = model.backbone(model.preprocess_image(inputs))
        L, h = model.engine._lib, model.engine._h
        assert L.ovm_backbone_num_levels(h) == n == len(feats)
        for i, (name, side, stride) in enumerate(model.engine.levels):
            ptr, s_, st = C.c_void_p(), C.c_int32(), C.c_float()
            assert L.ovm_backbone_level(h, i, C.byref(ptr), C.byref(s_), C.byref(st)) == 0
            assert s_.value == side == feats[name].shape[-1] and abs(st.value - stride) < 1e-6
            got = torch.empty(side * side * model.engine.C, dtype=torch.float32, device=device)
            torch.cuda.synchronize()
            C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(got.data_ptr()), ptr, C.c_size_t(got.numel() * 4), 3)   # device to device
            assert torch.equal(got.view(side, side, -1), feats[name][0].permute(1, 2, 0))
        assert L.ovm_backbone_level(h, n, None, None, None) != 0 and L.ovm_backbone_level(h, -1, None, None, None) != 0
        del model


def test_clip_tower_refuses_prompt_depth(device):
    """detectron2's SimpleFeaturePyramid.forward takes no depth; the fork would raise a TypeError at rcnn3d.py:97."""
    cfg = build_clip_cfg("ViT-test-16", 256, "f16x3", max_batch=1)
    model, sd = _build(cfg, seed=5)
    inputs = synth_inputs(1, hw=((160, 224),), n_boxes=4, seed=31, depth=True)
    with pytest.raises(TypeError):
        model(inputs, prompt_depth=torch.stack([x["depth"] for x in inputs]))
    from ovmono3d_amd.lib import OvmError
    native, _keep = model.engine.make_images(inputs)
    with pytest.raises(OvmError):
        model.engine.backbone_forward(native, 1, inputs[0]["depth"][None])


def test_clip_vitb16_canvas1024_config4_size(device):
    """BASELINE config 4 at its own size: CLIP ViT-B/16 (12 layers, D = 768), SQUARE_PAD 1024 -> 64 x 64 patches (T = 4097, the
    same token count as the headline ViT-L/896 run), p2..p5 = 256 / 128 / 64 / 32, one 608 x 1024-bounded image with 32 given
    boxes, against the CPU oracle."""
    from oracle.pipeline import inference
    cfg = build_clip_cfg("ViT-B-16", 1024, "f16x3", max_batch=1, max_rois=64)
    model, sd = _build(cfg, seed=0)
    inputs = synth_inputs(1, hw=((608, 800),), orig_scale=1.0, n_boxes=32, seed=13)
    # make sure every pyramid level is pooled from: ROIPooler level = floor(4 + log2(sqrt(area) / 224)) clamped to [2, 5]
    o = inputs[0]["oracle2D"]
    o["gt_bbox2D"][:4] = torch.tensor([[10.0, 10.0, 60.0, 50.0], [100.0, 80.0, 300.0, 260.0], [50.0, 40.0, 420.0, 400.0], [20.0, 10.0, 780.0, 600.0]])
    import math
    lv = {min(5, max(2, int(math.floor(4 + math.log2(math.sqrt(float((b[2] - b[0]) * (b[3] - b[1]))) / 224 + 1e-8))))) for b in o["gt_bbox2D"]}
    assert lv == {2, 3, 4, 5}, lv
    out = model(inputs)
    torch.set_num_threads(16)
    ref, aux = inference(sd, inputs, oracle_params(cfg), return_aux=True)
    model.backbone.export_features = True
    feats = model.backbone(model.preprocess_image(inputs))
    assert tuple(feats["p2"].shape) == (1, 256, 256, 256) and tuple(feats["p5"].shape) == (1, 256, 32, 32)
    for k in ("p2", "p3", "p4", "p5"):
        e = assert_close(feats[k], aux["features"][k], 1e-3, k)
        print(f"CLIP ViT-B/16 @1024 {k}: scale-relative error {e:.2e}")
    _compare(out, ref)


# ------------------------------------------------------------------------------------------ MAE tower ("next" row 3 analogue)
def test_mae_tower_tiny_and_vitb16_canvas1024(device):
    """build_mae_backbone: Hugging Face ViTMAE encoder keys (separate q / k / v linears, LN eps 1e-12, erf-GELU), 2-D sin-cos position
    table built on the host for the canvas grid, tap = the state before the last block (reference backbone/mae.py:43-118), behind
    the 4-level pyramid and heads - against the CPU oracle: a tiny tower (two image shapes, given boxes, then the RPN route) and
    facebook/vit-mae-base's architecture at canvas 1024."""
    from oracle.pipeline import inference
    cfg = build_mae_cfg("test/vit-mae-test", 256, "f16x3", max_batch=2)
    model, sd = _build(cfg, seed=6)
    inputs = synth_inputs(2, hw=((160, 224), (256, 192)), n_boxes=12, seed=41)
    out = model(inputs)
    ref, aux = inference(sd, inputs, oracle_params(cfg), return_aux=True)
    model.backbone.export_features = True
    feats = model.backbone(model.preprocess_image(inputs))
    for k in ("p2", "p3", "p4", "p5"):
        assert_close(feats[k], aux["features"][k], 2e-4, k)
    model.backbone.export_features = False
    _compare(out, ref)
    inputs2 = synth_inputs(2, hw=((192, 256), (256, 256)), n_boxes=0, seed=43, oracle2d=False)
    out2 = model(inputs2)
    assert len(out2[0]["instances"]) > 0
    _compare(out2, inference(sd, inputs2, oracle_params(cfg)))
    with pytest.raises(TypeError):
        model(synth_inputs(1, hw=((160, 224),), n_boxes=2, seed=1), prompt_depth=torch.zeros(1, 1, 8, 8))
    del model
    cfg = build_mae_cfg("facebook/vit-mae-base", 1024, "f16x3", max_batch=1, max_rois=64)
    model, sd = _build(cfg, seed=1)
    inputs = synth_inputs(1, hw=((608, 800),), orig_scale=1.0, n_boxes=32, seed=14)
    out = model(inputs)
    torch.set_num_threads(16)
    ref, aux = inference(sd, inputs, oracle_params(cfg), return_aux=True)
    model.backbone.export_features = True
    feats = model.backbone(model.preprocess_image(inputs))
    for k in ("p2", "p3", "p4", "p5"):
        e = assert_close(feats[k], aux["features"][k], 1e-3, k)
        print(f"MAE ViT-B/16 @1024 {k}: scale-relative error {e:.2e}")
    _compare(out, ref)


# ------------------------------------------------------------------------------------------ MiDaS tower ("next" row 3 analogue)
def test_midas_tower_tiny_and_dpt_large_canvas1024(device):
    """build_midas_backbone: MiDaS DPT_Large's timm ViT-L/16 keys (fused qkv, no LayerScale), position table of the 24 x 24 grid resized
    with antialiased bicubic (up to 64 x 64 and, in the tiny tower, 6 -> 16), behind the 4-level pyramid and heads - against the
    CPU oracle: a tiny tower on two image shapes and ViT-L/16 (24 layers, D = 1024, T = 4097) at canvas 1024."""
    from oracle.pipeline import inference
    cfg = build_midas_cfg("DPT_test", 256, "f16x3", max_batch=2)
    model, sd = _build(cfg, seed=8)
    inputs = synth_inputs(2, hw=((160, 224), (256, 192)), n_boxes=12, seed=51)
    out = model(inputs)
    ref, aux = inference(sd, inputs, oracle_params(cfg), return_aux=True)
    model.backbone.export_features = True
    feats = model.backbone(model.preprocess_image(inputs))
    for k in ("p2", "p3", "p4", "p5"):
        assert_close(feats[k], aux["features"][k], 2e-4, k)
    model.backbone.export_features = False
    _compare(out, ref)
    del model
    cfg = build_midas_cfg("DPT_Large", 1024, "f16x3", max_batch=1, max_rois=64)
    model, sd = _build(cfg, seed=2)
    inputs = synth_inputs(1, hw=((608, 800),), orig_scale=1.0, n_boxes=32, seed=15)
    out = model(inputs)
    torch.set_num_threads(16)
    ref, aux = inference(sd, inputs, oracle_params(cfg), return_aux=True)
    model.backbone.export_features = True
    feats = model.backbone(model.preprocess_image(inputs))
    for k in ("p2", "p3", "p4", "p5"):
        e = assert_close(feats[k], aux["features"][k], 1e-3, k)
        print(f"MiDaS ViT-L/16 @1024 {k}: scale-relative error {e:.2e}")
    _compare(out, ref)


# ------------------------------------------------------------------------------------------ SAM tower ("next" row 3 analogue)
def test_sam_tower_tiny(device):
    """build_sam_backbone at test size: no class token, position table bicubic-resized 8 -> 16, windowed blocks on a zero-padded grid
    (16 -> 18 = 3 x 3 windows of 6), global blocks with linearly resized relative-position tables (15 -> 31 entries), the decomposed
    bias from the unscaled query - against the CPU oracle (reference backbone/sam.py:73-112), batch 2, two image shapes."""
    from oracle.pipeline import inference
    cfg = build_sam_cfg("vit_test", 256, "f16x3", max_batch=2)
    model, sd = _build(cfg, seed=9)
    inputs = synth_inputs(2, hw=((160, 224), (256, 192)), n_boxes=12, seed=61)
    out = model(inputs)
    ref, aux = inference(sd, inputs, oracle_params(cfg), return_aux=True)
    model.backbone.export_features = True
    feats = model.backbone(model.preprocess_image(inputs))
    for k in ("p2", "p3", "p4", "p5"):
        assert_close(feats[k], aux["features"][k], 2e-4, k)
    model.backbone.export_features = False
    _compare(out, ref)


def test_sam_vitb_canvas1024(device):
    """segment_anything vit_b's architecture at its own size: 64 x 64 grid (no table resize), 14 x 14 windows on the grid padded to 70
    (25 windows, 196 tokens each), global attention over 4096 tokens in blocks 2 / 5 / 8 / 11, against the CPU oracle."""
    from oracle.pipeline import inference
    cfg = build_sam_cfg("vit_b", 1024, "f16x3", max_batch=1, max_rois=64)
    model, sd = _build(cfg, seed=3)
    inputs = synth_inputs(1, hw=((608, 800),), orig_scale=1.0, n_boxes=32, seed=16)
    out = model(inputs)
    torch.set_num_threads(16)
    ref, aux = inference(sd, inputs, oracle_params(cfg), return_aux=True)
    model.backbone.export_features = True
    feats = model.backbone(model.preprocess_image(inputs))
    for k in ("p2", "p3", "p4", "p5"):
        e = assert_close(feats[k], aux["features"][k], 1e-3, k)
        print(f"SAM ViT-B @1024 {k}: scale-relative error {e:.2e}")
    _compare(out, ref)
    torch.cuda.synchronize()
    import time
    t0 = time.time()
    for _ in range(5):
        model.backbone(model.preprocess_image(inputs))
    torch.cuda.synchronize()
    print(f"SAM ViT-B @1024 backbone: {(time.time() - t0) / 5 * 1e3:.2f} ms per image")


# ------------------------------------------------------------------------------------------ BASELINE configs 3 / 4 / 5 at their per-GPU batch
def _batch8_every_image(device, cfg, sd_seed, shapes, orig_scale, n_boxes, seed, label):
    """A real batch of 8 images per GPU (BASELINE configs[2..4]: 16 / 2, 32 / 4, 64 / 8) of >= 3 distinct network shapes through
    one call, every image checked (reference omni3d_evaluation.py:652-667 feeds the model whole batches):
      * images 0 and 7 against the CPU oracle: ids exact, every float field within 1e-3;
      * every image against ITS OWN batch-1 run on the HIP path: ids exact, floats within 1e-5 (scale-relative; pred_pose 1e-4). A batch never changes
        the arithmetic, only fp32 summation order: whether a thin GEMM grid is split along K (the pyramid's convolutions: <= 96 tiles at
        B = 1, more at B = 8), and which rows of the token matrix fall into the leftover-row dot-product workgroups (M = B x 4097 =
        whole tiles + 1 row at B = 1, + 8 rows at B = 8) - ~1e-7 on the features, amplified to 2-4e-6 on pred_pose by the
        random-init 6-D head (measured; every field is reported below) - the batch-1 route is oracle-anchored by
        test_oracle2d_vitl_canvas896_headline_size / test_clip_vitb16_canvas1024_config4_size and by images 0 and 7 here."""
    from oracle.pipeline import inference
    model, sd = _build(cfg, seed=sd_seed)
    inputs = synth_inputs(8, hw=shapes, orig_scale=orig_scale, n_boxes=n_boxes, seed=seed)
    assert len({tuple(d["image"].shape[1:]) for d in inputs}) >= 3
    out = model(inputs)
    assert len(out) == 8
    worst = {f: 0.0 for f in FIELDS}
    for i in range(8):
        solo = model([inputs[i]])[0]["instances"]
        inst = out[i]["instances"]
        assert len(inst) == len(solo) == n_boxes, (i, len(inst), len(solo))
        assert torch.equal(inst.pred_classes, solo.pred_classes), f"image {i}: category indices differ between batch 8 and batch 1"
        for f in FIELDS:
            a, b = inst.get(f), solo.get(f)
            a, b = (a.tensor if hasattr(a, "tensor") else a), (b.tensor if hasattr(b, "tensor") else b)
            # pred_pose passes through rotation_6d_to_matrix, which amplifies by up to ~50x on this random-init checkpoint where a2 is
            # nearly parallel to a1 (tests/parity.py:pose_conditioning; measured 1.4e-5 on one box of 128): 1e-4 there, 1e-5 elsewhere
            worst[f] = max(worst[f], assert_close(a, b, 1e-4 if f == "pred_pose" else 1e-5, f"image {i} {f} (batch 8 vs batch 1)"))
    print(f"{label}: batch 8 vs batch 1 on the HIP path, worst error over 8 images: " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    torch.set_num_threads(16)
    ref = inference(sd, [inputs[0], inputs[7]], oracle_params(cfg))
    _compare([out[0], out[7]], ref)
    errs = {f: max(float((((o["instances"].get(f).tensor if hasattr(o["instances"].get(f), "tensor") else o["instances"].get(f)).cpu().double()
                           - r[f].double()).abs().max() / r[f].double().abs().max())) for o, r in zip([out[0], out[7]], ref)) for f in FIELDS}
    print(f"{label}: images 0 and 7 vs the CPU oracle: " + ", ".join(f"{k} {v:.1e}" for k, v in errs.items()))


def test_config3_vitl_canvas896_batch8_every_image(device):
    """BASELINE configs[2] (Omni3D novel-split eval, batch 16 over 2 GPUs = 8 per GPU): DINOv2 ViT-L/14, 24 layers, canvas 896
    (T = 4097), 8 images of 4 network shapes as ResizeShortestEdge(532, 896) emits them, 16 given boxes each."""
    cfg = build_cfg("vitl14", 896, "f16x3", max_batch=8, max_rois=64)
    _batch8_every_image(device, cfg, 0, ((532, 532), (532, 709), (709, 532), (504, 896)), 512.0 / 532.0, 16, 71, "config 3 (ViT-L/14 @896)")


def test_config4_clip_vitb16_canvas1024_batch8_every_image(device):
    """BASELINE configs[3] (OVMono3D_clip_SFP.yaml, batch 32 over 4 GPUs = 8 per GPU): CLIP ViT-B/16 at canvas 1024 (T = 4097),
    4-level pyramid, 8 images of 3 shapes."""
    cfg = build_clip_cfg("ViT-B-16", 1024, "f16x3", max_batch=8, max_rois=64)
    _batch8_every_image(device, cfg, 0, ((608, 800), (608, 1024), (800, 608)), 1.0, 16, 72, "config 4 (CLIP ViT-B/16 @1024)")


def test_config5_vitl_24layers_canvas1036_batch8_every_image(device):
    """BASELINE configs[4] (1024 x 1024 inputs, batch 64 over 8 GPUs = 8 per GPU) at its full depth: DINOv2 ViT-L/14, all 24
    layers, SQUARE_PAD 1036 -> 74 x 74 patches, T = 5477 (partial last attention block, odd p4 = 37), 8 images of 3 shapes -
    replaces the depth-2 stand-in of test_config5_geometry_vitl_width_canvas1036_batch2 for the full network."""
    cfg = build_cfg("vitl14", 1036, "f16x3", max_batch=8, max_rois=64)
    _batch8_every_image(device, cfg, 0, ((1024, 1024), (768, 1024), (1024, 683)), 1.0, 16, 73, "config 5 (ViT-L/14 x 24 @1036, T=5477)")
