"""Pins the oracle's third-party arithmetic against independent implementations available offline:
HF transformers' Dinov2 layer, scipy rotations, torch.nn.functional primitives and closed-form
properties. (The reference itself cannot be imported here and ships no value-asserting tests:
SURVEY.md §4/§8c - anything not covered below is "parity unpinned".)"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import heads as OH
from oracle import roi_ops, rpn, sfp, vit
from ovmono3d_amd.util.synth_weights import synth_state_dict


def test_vit_block_matches_hf_dinov2_layer():
    from transformers import Dinov2Config, Dinov2Model
    D, heads = 128, 2
    sd = synth_state_dict("vittest14", seed=4)
    m = Dinov2Model(Dinov2Config(hidden_size=D, num_hidden_layers=2, num_attention_heads=heads, image_size=224,
                                 patch_size=14, mlp_ratio=4, layer_norm_eps=1e-6, hidden_act="gelu")).eval()
    hf = {}
    for i in range(2):
        p, q = f"backbone.net.vit.blocks.{i}.", f"encoder.layer.{i}."
        wq, wk, wv = sd[p + "attn.qkv.weight"].chunk(3, 0)
        bq, bk, bv = sd[p + "attn.qkv.bias"].chunk(3, 0)
        hf.update({q + "norm1.weight": sd[p + "norm1.weight"], q + "norm1.bias": sd[p + "norm1.bias"],
                   q + "attention.attention.query.weight": wq, q + "attention.attention.query.bias": bq,
                   q + "attention.attention.key.weight": wk, q + "attention.attention.key.bias": bk,
                   q + "attention.attention.value.weight": wv, q + "attention.attention.value.bias": bv,
                   q + "attention.output.dense.weight": sd[p + "attn.proj.weight"],
                   q + "attention.output.dense.bias": sd[p + "attn.proj.bias"],
                   q + "layer_scale1.lambda1": sd[p + "ls1.gamma"], q + "layer_scale2.lambda1": sd[p + "ls2.gamma"],
                   q + "norm2.weight": sd[p + "norm2.weight"], q + "norm2.bias": sd[p + "norm2.bias"],
                   q + "mlp.fc1.weight": sd[p + "mlp.fc1.weight"], q + "mlp.fc1.bias": sd[p + "mlp.fc1.bias"],
                   q + "mlp.fc2.weight": sd[p + "mlp.fc2.weight"], q + "mlp.fc2.bias": sd[p + "mlp.fc2.bias"]})
    missing, unexpected = m.load_state_dict(hf, strict=False)
    assert not unexpected
    x = torch.randn(2, 50, D, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        y_hf = x
        for lyr in m.encoder.layer:
            o = lyr(y_hf)
            y_hf = o[0] if isinstance(o, (tuple, list)) else o
        y = x
        for i in range(2):
            y = vit.block(y, sd, f"backbone.net.vit.blocks.{i}.", heads)
    assert (y - y_hf).abs().max() < 2e-5 * y_hf.abs().max()


def test_patch_embed_and_cls_token_layout():
    sd = synth_state_dict("vittest14", seed=1)
    img = torch.randn(1, 3, 28, 42)
    x = vit.prepare_tokens(sd, img, "backbone.net.vit.")
    assert x.shape == (1, 1 + 2 * 3, 128)
    pos = vit.interpolate_pos_encoding(sd["backbone.net.vit.pos_embed"], 2, 3)
    w = sd["backbone.net.vit.patch_embed.proj.weight"]
    manual = (img[0, :, 14:28, 28:42] * w[5]).sum() + sd["backbone.net.vit.patch_embed.proj.bias"][5]
    assert abs(float(x[0, 1 + 1 * 3 + 2, 5] - pos[0, 1 + 5, 5]) - float(manual)) < 1e-4
    assert torch.allclose(x[0, 0], sd["backbone.net.vit.cls_token"][0, 0] + pos[0, 0])


def test_channel_layernorm_equals_functional():
    x = torch.randn(2, 16, 5, 7)
    w, b = torch.rand(16) + 0.5, torch.randn(16)
    ref = F.layer_norm(x.permute(0, 2, 3, 1), (16,), w, b, 1e-6).permute(0, 3, 1, 2)
    assert torch.allclose(sfp.channel_layer_norm(x, w, b), ref, atol=1e-5)


def test_roi_align_closed_form_on_ramps():
    H = W = 32
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    feat = torch.stack([xx, yy, torch.ones_like(xx)])
    for box in ([131.3, 50.86, 150.33, 58.82], [30.0, 40.0, 166.0, 137.0]):
        b = torch.tensor(box)
        o = roi_ops.roi_align_single(feat, b, 1 / 7, 7)
        x1, y1 = b[0] / 7 - 0.5, b[1] / 7 - 0.5
        w, h = (b[2] - b[0]) / 7, (b[3] - b[1]) / 7
        ex = torch.tensor([float(x1 + (i + 0.5) * w / 7) for i in range(7)])
        ey = torch.tensor([float(y1 + (i + 0.5) * h / 7) for i in range(7)])
        assert torch.allclose(o[0, 0], ex, atol=1e-4) and torch.allclose(o[1, :, 0], ey, atol=1e-4)
        assert torch.allclose(o[2], torch.ones(7, 7), atol=1e-6)
    # empty / inverted boxes give zeros (sampling grid of size <= 0)
    assert roi_ops.roi_align_single(feat, torch.tensor([5.0, 5.0, 5.0, 9.0]), 1 / 7).abs().sum() == 0


def test_level_assignment_rule():
    b = torch.tensor([[0, 0, 10, 10], [0, 0, 224, 224], [0, 0, 111, 111], [0, 0, 112, 112], [0, 0, 900, 900]], dtype=torch.float32)
    assert roi_ops.assign_boxes_to_levels(b, 2, 4).tolist() == [0, 2, 0, 1, 2]


def _nms_bruteforce(boxes, scores, thr):
    order = sorted(range(len(scores)), key=lambda i: -float(scores[i]))
    keep = []
    for i in order:
        ok = True
        for j in keep:
            xx1, yy1 = max(boxes[i][0], boxes[j][0]), max(boxes[i][1], boxes[j][1])
            xx2, yy2 = min(boxes[i][2], boxes[j][2]), min(boxes[i][3], boxes[j][3])
            inter = max(0.0, xx2 - xx1) * max(0.0, yy2 - yy1)
            a = (boxes[i][2] - boxes[i][0]) * (boxes[i][3] - boxes[i][1])
            c = (boxes[j][2] - boxes[j][0]) * (boxes[j][3] - boxes[j][1])
            if inter / (a + c - inter) > thr:
                ok = False
                break
        if ok:
            keep.append(i)
    return keep


def test_nms_matches_bruteforce():
    g = torch.Generator().manual_seed(0)
    xy = torch.rand(200, 2, generator=g) * 100
    wh = torch.rand(200, 2, generator=g) * 40 + 2
    boxes = torch.cat([xy, xy + wh], 1)
    scores = torch.rand(200, generator=g)
    for thr in (0.3, 0.5, 0.7):
        assert roi_ops.nms(boxes, scores, thr).tolist() == _nms_bruteforce(boxes.tolist(), scores.tolist(), thr)
    idxs = torch.randint(0, 3, (200,), generator=g)
    k = roi_ops.batched_nms(boxes, scores, idxs, 0.5)
    exp = []
    for c in range(3):
        sel = torch.where(idxs == c)[0]
        exp += [int(sel[i]) for i in _nms_bruteforce(boxes[sel].tolist(), scores[sel].tolist(), 0.5)]
    assert sorted(k.tolist()) == sorted(exp)
    assert all(scores[k[i]] >= scores[k[i + 1]] for i in range(len(k) - 1))


def test_rotations_against_scipy():
    from scipy.spatial.transform import Rotation
    g = torch.Generator().manual_seed(1)
    aa = torch.randn(64, 3, generator=g)
    aa[0] = torch.tensor([1e-9, 0.0, 0.0])                      # small-angle branch
    M = OH.axis_angle_to_matrix(aa)
    ref = torch.from_numpy(Rotation.from_rotvec(aa.numpy().astype(np.float64)).as_matrix()).float()
    assert (M - ref).abs().max() < 2e-6
    d6 = torch.randn(32, 6, generator=g)
    R = OH.rotation_6d_to_matrix(d6)
    assert (R @ R.transpose(1, 2) - torch.eye(3)).abs().max() < 1e-5
    assert torch.allclose(torch.linalg.det(R), torch.ones(32), atol=1e-5)
    assert torch.allclose(R[:, 0], F.normalize(d6[:, :3], dim=-1), atol=1e-6)


def test_allocentric_is_identity_on_the_optical_axis_and_rotates_the_view_ray():
    K = torch.tensor([[[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]]]).repeat(2, 1, 1)
    R = OH.rotation_6d_to_matrix(torch.randn(2, 6, generator=torch.Generator().manual_seed(2)))
    u, v = torch.tensor([320.0, 500.0]), torch.tensor([240.0, 100.0])
    out = OH.R_from_allocentric(K, R, u, v)
    assert torch.allclose(out[0], R[0])                          # angle == 0 -> untouched (math_util.py:672-679)
    Mrot = out[1] @ R[1].T                                       # the applied rotation maps +z onto the object ray
    ray = torch.tensor([(500.0 - 320) / 500, (100.0 - 240) / 500, 1.0])
    ray = ray / ray.norm()
    assert torch.allclose(Mrot @ torch.tensor([0.0, 0.0, 1.0]), ray, atol=1e-5)


def test_cuboid_vertices_geometry():
    box = torch.tensor([[0.5, -0.2, 4.0, 1.0, 2.0, 3.0]])       # X Y Z W H L
    v = OH.get_cuboid_verts(box, torch.eye(3).unsqueeze(0))[0]
    assert torch.allclose(v.mean(0), box[0, :3], atol=1e-6)
    assert torch.allclose(v[1] - v[0], torch.tensor([3.0, 0, 0]))    # L along x (verts 0->1)
    assert torch.allclose(v[3] - v[0], torch.tensor([0, 2.0, 0]))    # H along y (0->3)
    assert torch.allclose(v[4] - v[0], torch.tensor([0, 0, 1.0]))    # W along z (0->4)


def test_box2box_and_anchors():
    base = rpn.cell_anchors(64.0, (0.5, 1.0, 2.0))
    assert torch.allclose((base[:, 2] - base[:, 0]) * (base[:, 3] - base[:, 1]), torch.full((3,), 4096.0), rtol=1e-5)
    assert torch.allclose((base[:, 3] - base[:, 1]) / (base[:, 2] - base[:, 0]), torch.tensor([0.5, 1.0, 2.0]), rtol=1e-5)
    a = rpn.grid_anchors(2, 3, 7, base)
    assert a.shape == (18, 4) and torch.allclose(a[3:6], base + torch.tensor([7.0, 0, 7.0, 0]))
    boxes = torch.tensor([[10.0, 20.0, 50.0, 80.0]])
    assert torch.allclose(rpn.apply_deltas(torch.zeros(1, 4), boxes), boxes)
    big = rpn.apply_deltas(torch.tensor([[0.0, 0.0, 100.0, 100.0]]), boxes)       # clamp at log(1000/16)
    assert torch.allclose(big[0, 2] - big[0, 0], torch.tensor(40.0 * 1000 / 16), rtol=1e-5)
    d = rpn.apply_deltas(torch.tensor([[1.0, -1.0, 0.0, 0.0]]), boxes, (10.0, 10.0, 5.0, 5.0))
    assert torch.allclose(d, boxes + torch.tensor([4.0, -6.0, 4.0, -6.0]))


def test_virtual_depth_scale():
    # same focal / height as the virtual camera -> scale 1 (math_util.py:581-592)
    assert OH.compute_virtual_scale_from_focal_spaces(512.0, 512.0, 512.0, 512.0) == 1.0
    assert OH.compute_virtual_scale_from_focal_spaces(1024.0, 512.0, 512.0, 532.0) == pytest.approx(2 * 532 / 512)
