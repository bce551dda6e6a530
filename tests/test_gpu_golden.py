"""GPU: the HIP path reproduces the committed golden vectors (1e-3 scale-relative on floats, exact categories)."""
import pytest
import torch

from common import assert_close, build_cfg, golden_weights, load_golden

pytestmark = pytest.mark.gpu
FLOAT_FIELDS = ("pred_boxes", "scores", "pred_bbox3D", "pred_center_cam", "pred_center_2D", "pred_dimensions", "pred_pose")


def _run(name, device):
    from ovmono3d_amd.modeling import build_model
    inputs, expected, z = load_golden(name)
    sd, ok = golden_weights(z)
    if not ok:
        pytest.skip("torch CPU RNG stream differs from the one the fixtures were generated with")
    cfg = build_cfg("vittest14", 224, "f16x3", max_batch=2)
    model = build_model(cfg)
    model.load_state_dict(sd)
    depth = torch.stack([d["depth"] for d in inputs]) if "depth" in inputs[0] else None
    out = model(inputs, prompt_depth=depth)
    return out, expected, z, model, inputs


@pytest.mark.parametrize("name", ["e2e_oracle2d_vittest14.npz", "e2e_depth_vittest14.npz", "e2e_rpn_vittest14.npz"])
def test_native_matches_golden(device, name):
    out, expected, z, model, inputs = _run(name, device)
    for o, e in zip(out, expected):
        inst = o["instances"]
        assert torch.equal(inst.pred_classes.cpu(), e["pred_classes"])
        for f in FLOAT_FIELDS:
            got = inst.get(f)
            got = got.tensor if hasattr(got, "tensor") else got
            assert_close(got, e[f], 1e-3, f)
    if "feat_p3_corner" in z:
        model.backbone.export_features = True
        feats = model.backbone(model.preprocess_image(inputs))
        for k in ("p2", "p3", "p4"):
            assert_close(feats[k][:, :8, :6, :6], torch.from_numpy(z[f"feat_{k}_corner"]), 5e-4, k)
