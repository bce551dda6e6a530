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


def _hf_clip_from_open_clip_keys(sd, D, L, heads, patch, grid, pos_table):
    """Hugging Face CLIPVisionModel (random-init, no download) carrying an open_clip-keyed tower; the position table is given
    already resized so that both sides see the same embedding (the resize itself is tested against F.interpolate in
    test_host.py)."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfg = CLIPVisionConfig(hidden_size=D, intermediate_size=4 * D, num_hidden_layers=L, num_attention_heads=heads, num_channels=3,
                           image_size=grid * patch, patch_size=patch, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                           attn_implementation="eager")
    m = CLIPVisionModel(cfg).eval()
    V = "backbone.net.visual."
    hf = {"vision_model.embeddings.class_embedding": sd[V + "class_embedding"],
          "vision_model.embeddings.patch_embedding.weight": sd[V + "conv1.weight"],
          "vision_model.embeddings.position_embedding.weight": pos_table,
          "vision_model.pre_layrnorm.weight": sd[V + "ln_pre.weight"], "vision_model.pre_layrnorm.bias": sd[V + "ln_pre.bias"],
          "vision_model.post_layernorm.weight": sd[V + "ln_post.weight"], "vision_model.post_layernorm.bias": sd[V + "ln_post.bias"]}
    for i in range(L):
        s, d = V + f"transformer.resblocks.{i}.", f"vision_model.encoder.layers.{i}."
        wq, wk, wv = sd[s + "attn.in_proj_weight"].chunk(3, 0)
        bq, bk, bv = sd[s + "attn.in_proj_bias"].chunk(3, 0)
        for n, w, b in (("q", wq, bq), ("k", wk, bk), ("v", wv, bv)):
            hf[d + f"self_attn.{n}_proj.weight"], hf[d + f"self_attn.{n}_proj.bias"] = w, b
        hf[d + "self_attn.out_proj.weight"], hf[d + "self_attn.out_proj.bias"] = sd[s + "attn.out_proj.weight"], sd[s + "attn.out_proj.bias"]
        hf[d + "layer_norm1.weight"], hf[d + "layer_norm1.bias"] = sd[s + "ln_1.weight"], sd[s + "ln_1.bias"]
        hf[d + "layer_norm2.weight"], hf[d + "layer_norm2.bias"] = sd[s + "ln_2.weight"], sd[s + "ln_2.bias"]
        hf[d + "mlp.fc1.weight"], hf[d + "mlp.fc1.bias"] = sd[s + "mlp.c_fc.weight"], sd[s + "mlp.c_fc.bias"]
        hf[d + "mlp.fc2.weight"], hf[d + "mlp.fc2.bias"] = sd[s + "mlp.c_proj.weight"], sd[s + "mlp.c_proj.bias"]
    if not any(k.startswith("vision_model.") for k in m.state_dict()):          # key prefix differs between transformers releases
        hf = {k[len("vision_model."):]: v for k, v in hf.items()}
    missing, unexpected = m.load_state_dict(hf, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    return m


def test_clip_tower_matches_hf_clip_vision_model():
    """oracle/clip_vit.py (the reference's CLIPBackbone.forward over open_clip's tower) against an independent implementation of
    the same architecture: conv1 without bias, class token, ln_pre, pre-norm blocks with QuickGELU, last block's patch tokens
    before any final norm. Same weights, same (already resized) position table."""
    from oracle import clip_vit
    from ovmono3d_amd.util.synth_weights import CLIP_ARCH, synth_clip_state_dict
    arch = "ViT-test-16"
    D, L, heads, patch, M = CLIP_ARCH[arch]
    sd = synth_clip_state_dict(arch, seed=11)
    grid = 12
    x = torch.randn(2, 3, grid * patch, grid * patch, generator=torch.Generator().manual_seed(0))
    dense = clip_vit.clip_backbone_forward(sd, x, heads, L)
    assert dense.shape == (2, D, grid, grid)
    pos = clip_vit.resize_pos_embed(sd["backbone.net.visual.positional_embedding"], (grid, grid))
    m = _hf_clip_from_open_clip_keys(sd, D, L, heads, patch, grid, pos)
    with torch.no_grad():
        hs = m(pixel_values=x, output_hidden_states=True).hidden_states[-1]   # last encoder layer, before post_layernorm
    ref = hs[:, 1:].reshape(2, grid, grid, D).permute(0, 3, 1, 2)
    assert float((dense - ref).abs().max() / ref.abs().max()) < 2e-5
    # native grid: the table is used as it is
    x14 = torch.randn(1, 3, M * patch, M * patch, generator=torch.Generator().manual_seed(1))
    assert clip_vit.resize_pos_embed(sd["backbone.net.visual.positional_embedding"], (M, M)) is sd["backbone.net.visual.positional_embedding"]
    d14 = clip_vit.clip_backbone_forward(sd, x14, heads, L)
    m14 = _hf_clip_from_open_clip_keys(sd, D, L, heads, patch, M, sd["backbone.net.visual.positional_embedding"])
    with torch.no_grad():
        h14 = m14(pixel_values=x14, output_hidden_states=True).hidden_states[-1]
    assert float((d14 - h14[:, 1:].reshape(1, M, M, D).permute(0, 3, 1, 2)).abs().max() / h14.abs().max()) < 2e-5


def test_sfp4_stage_shapes_and_scale4_branch():
    """The 4-level pyramid of the CLIP config: strides P/4, P/2, P, 2P; the scale-4 branch is ConvT . LN . GELU . ConvT, checked
    against torch modules wired the way detectron2's SimpleFeaturePyramid builds them."""
    from ovmono3d_amd.util.synth_weights import CLIP_ARCH, synth_clip_state_dict
    sd = synth_clip_state_dict("ViT-test-16", seed=4, fpn_channels=64)
    D = CLIP_ARCH["ViT-test-16"][0]
    feat = torch.randn(1, D, 6, 6, generator=torch.Generator().manual_seed(2))
    out = sfp.sfp4_forward(sd, feat)
    assert {k: tuple(v.shape) for k, v in out.items()} == {"p2": (1, 64, 24, 24), "p3": (1, 64, 12, 12), "p4": (1, 64, 6, 6), "p5": (1, 64, 3, 3)}
    up1 = torch.nn.ConvTranspose2d(D, D // 2, 2, 2); up2 = torch.nn.ConvTranspose2d(D // 2, D // 4, 2, 2)
    c1 = torch.nn.Conv2d(D // 4, 64, 1, bias=False); c3 = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False)
    with torch.no_grad():
        up1.weight.copy_(sd["backbone.simfp_2.0.weight"]); up1.bias.copy_(sd["backbone.simfp_2.0.bias"])
        up2.weight.copy_(sd["backbone.simfp_2.3.weight"]); up2.bias.copy_(sd["backbone.simfp_2.3.bias"])
        c1.weight.copy_(sd["backbone.simfp_2.4.weight"]); c3.weight.copy_(sd["backbone.simfp_2.5.weight"])
        y = up1(feat)
        y = F.layer_norm(y.permute(0, 2, 3, 1), (D // 2,), sd["backbone.simfp_2.1.weight"], sd["backbone.simfp_2.1.bias"], 1e-6).permute(0, 3, 1, 2)
        y = up2(F.gelu(y))
        y = F.layer_norm(c1(y).permute(0, 2, 3, 1), (64,), sd["backbone.simfp_2.4.norm.weight"], sd["backbone.simfp_2.4.norm.bias"], 1e-6).permute(0, 3, 1, 2)
        y = F.layer_norm(c3(y).permute(0, 2, 3, 1), (64,), sd["backbone.simfp_2.5.norm.weight"], sd["backbone.simfp_2.5.norm.bias"], 1e-6).permute(0, 3, 1, 2)
    assert float((out["p2"] - y).abs().max()) < 1e-4


def test_mae_tower_matches_hf_vitmae_encoder_and_sincos_table():
    """oracle/mae_vit.py against Hugging Face's ViTMAEModel: the 2-D sin-cos position table (x half first - the layout the
    pretrained weights rely on) against the table the HF model initialises itself with, and the reference's own recipe
    (embeddings without masking -> encoder with output_hidden_states -> hidden_states[num_layers - 1]) with the same weights."""
    from transformers import ViTMAEConfig, ViTMAEModel
    from oracle import mae_vit
    from ovmono3d_amd import lib
    from ovmono3d_amd.util.synth_weights import MAE_ARCH, synth_mae_state_dict
    name = "test/vit-mae-test"
    D, L, heads, patch = MAE_ARCH[name]
    grid = 10
    sd = synth_mae_state_dict(name, seed=7)
    cfg = ViTMAEConfig(hidden_size=D, num_hidden_layers=L, num_attention_heads=heads, intermediate_size=4 * D, image_size=grid * patch,
                       patch_size=patch, mask_ratio=0.0, layer_norm_eps=1e-12, hidden_act="gelu", attn_implementation="eager")
    torch.manual_seed(0)
    m = ViTMAEModel(cfg).eval()
    table = torch.from_numpy(mae_vit.sincos_2d(D, grid, grid)).float()
    from transformers.models.vit_mae import modeling_vit_mae as hf_mae
    if hasattr(hf_mae, "build_2d_sinusoidal_position_embedding"):             # transformers 5.x: canonical [h | w] halves, which its
        t5 = hf_mae.build_2d_sinusoidal_position_embedding(height=grid, width=grid, embed_dim=D, cls_token=True)   # embeddings rotate back
        t5 = torch.cat([t5[..., D // 2:], t5[..., :D // 2]], dim=-1).reshape(-1, D)
        assert float((t5 - table).abs().max()) < 1e-6
    host = np.empty((1 + grid * grid, D), np.float32)
    assert lib.load().ovm_host_sincos_pos_embed(D, grid, host.ctypes.data) == 0
    assert np.abs(host - table.numpy()).max() < 1e-6 and not host[0].any()
    rect = mae_vit.sincos_2d(D, 3, 5)                                          # non-square grid: x varies fastest
    assert rect.shape == (16, D) and np.allclose(rect[1 + 1, :D // 2], rect[1 + 5 + 1, :D // 2]) and not np.allclose(rect[2], rect[3])
    hf = {k[len("backbone.net.vit."):]: v for k, v in sd.items() if k.startswith("backbone.net.vit.")}
    hf["embeddings.position_embeddings"] = table[None]
    if hasattr(m, "layers"):                                                   # transformers 5.x renamed the 4.46 parameter tree the checkpoint uses
        ren = (("encoder.layer.", "layers."), (".attention.attention.query.", ".attention.q_proj."), (".attention.attention.key.", ".attention.k_proj."),
               (".attention.attention.value.", ".attention.v_proj."), (".attention.output.dense.", ".attention.o_proj."),
               (".intermediate.dense.", ".mlp.fc1."), (".output.dense.", ".mlp.fc2."))
        def new_name(k):
            for a, b in ren:
                k = k.replace(a, b)
            return k
        hf = {new_name(k): v for k, v in hf.items()}
    missing, unexpected = m.load_state_dict(hf, strict=False)
    assert not unexpected and not [k for k in missing if "position_ids" not in k], (missing, unexpected)
    x = torch.randn(2, 3, grid * patch, grid * patch, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        emb = m.embeddings.patch_embeddings(x) + m.embeddings.position_embeddings[:, 1:, :]            # mae.py:80-95, no masking
        cls = (m.embeddings.cls_token + m.embeddings.position_embeddings[:, :1, :]).expand(2, -1, -1)
        h = torch.cat((cls, emb), dim=1)
        states = [h]
        for layer in (m.layers if hasattr(m, "layers") else m.encoder.layer):    # = encoder(..., output_hidden_states=True).hidden_states
            o = layer(h)
            h = o[0] if isinstance(o, (tuple, list)) else o
            states.append(h)
        ref = states[L - 1][:, 1:].reshape(2, grid, grid, D).permute(0, 3, 1, 2)
    got = mae_vit.mae_backbone_forward(sd, x, heads, L)
    assert got.shape == ref.shape and float((got - ref).abs().max() / ref.abs().max()) < 2e-5
    assert float((got - states[L][:, 1:].reshape(2, grid, grid, D).permute(0, 3, 1, 2)).abs().max()) > 1e-3   # NOT the last block's output


def test_midas_tower_block_matches_hf_vit_layer():
    """oracle/midas_vit.py's timm-style block (fused qkv, no LayerScale, erf-GELU, LN eps 1e-6) against Hugging Face ViT layers of the
    same architecture with the fused qkv split into query / key / value; and the whole forward's geometry (class token, resized
    position table, last block's patch tokens, no final norm)."""
    from transformers import ViTConfig, ViTModel
    from oracle import clip_vit, midas_vit
    from ovmono3d_amd.util.synth_weights import MIDAS_ARCH, synth_midas_state_dict
    arch = "DPT_test"
    D, L, heads, patch, M = MIDAS_ARCH[arch]
    sd = synth_midas_state_dict(arch, seed=9)
    m = ViTModel(ViTConfig(hidden_size=D, num_hidden_layers=L, num_attention_heads=heads, intermediate_size=4 * D, image_size=M * patch,
                           patch_size=patch, layer_norm_eps=1e-6, hidden_act="gelu", attn_implementation="eager"), add_pooling_layer=False).eval()
    layers = m.encoder.layer if hasattr(m, "encoder") else m.layers
    names = dict(layers[0].named_parameters())
    new = "attention.q_proj.weight" in names
    with torch.no_grad():
        for i, lyr in enumerate(layers):
            p = f"backbone.net.vit.blocks.{i}."
            wq, wk, wv = sd[p + "attn.qkv.weight"].chunk(3, 0)
            bq, bk, bv = sd[p + "attn.qkv.bias"].chunk(3, 0)
            mapping = {("attention.q_proj" if new else "attention.attention.query"): (wq, bq),
                       ("attention.k_proj" if new else "attention.attention.key"): (wk, bk),
                       ("attention.v_proj" if new else "attention.attention.value"): (wv, bv),
                       ("attention.o_proj" if new else "attention.output.dense"): (sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"]),
                       "layernorm_before": (sd[p + "norm1.weight"], sd[p + "norm1.bias"]), "layernorm_after": (sd[p + "norm2.weight"], sd[p + "norm2.bias"]),
                       ("mlp.fc1" if new else "intermediate.dense"): (sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]),
                       ("mlp.fc2" if new else "output.dense"): (sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])}
            params = dict(lyr.named_parameters())
            assert set(params) == {k + s for k in mapping for s in (".weight", ".bias")}, sorted(params)
            for k, (w, b) in mapping.items():
                params[k + ".weight"].copy_(w); params[k + ".bias"].copy_(b)
        x = torch.randn(2, 37, D, generator=torch.Generator().manual_seed(0))
        y_hf = x
        for lyr in layers:
            o = lyr(y_hf)
            y_hf = o[0] if isinstance(o, (tuple, list)) else o
        y = x
        for i in range(L):
            y = midas_vit.timm_block(y, sd, f"backbone.net.vit.blocks.{i}.", heads)
    assert float((y - y_hf).abs().max() / y_hf.abs().max()) < 2e-5
    grid = 10
    img = torch.randn(1, 3, grid * patch, grid * patch, generator=torch.Generator().manual_seed(1))
    dense = midas_vit.midas_backbone_forward(sd, img, heads, L)
    assert dense.shape == (1, D, grid, grid)
    # by hand: tokens = [cls | patches] + resized table, then the blocks
    V = "backbone.net.vit."
    t = torch.nn.functional.conv2d(img, sd[V + "patch_embed.proj.weight"], sd[V + "patch_embed.proj.bias"], stride=patch).flatten(2).transpose(1, 2)
    t = torch.cat([sd[V + "cls_token"], t], 1) + clip_vit.resize_pos_embed(sd[V + "pos_embed"][0], (grid, grid))[None]
    for i in range(L):
        t = midas_vit.timm_block(t, sd, V + f"blocks.{i}.", heads)
    assert torch.equal(dense, t[:, 1:].reshape(1, grid, grid, D).permute(0, 3, 1, 2))


def test_sam_blocks_match_hf_sam_vision_layers():
    """oracle/sam_vit.py's blocks (zero-padded windows, decomposed relative-position bias from the unscaled query, linearly resized
    tables) against Hugging Face ``SamVisionLayer``s of the same architecture - a windowed and a global block, on a grid (16) that is
    neither a multiple of the window (6) nor the checkpoint's grid (8), so padding and the table resize both take part."""
    from transformers import SamVisionConfig
    from transformers.models.sam.modeling_sam import SamVisionLayer
    from oracle import sam_vit
    from ovmono3d_amd.util.synth_weights import SAM_ARCH, synth_sam_state_dict
    arch = "vit_test"
    D, L, heads, patch, M, ws, glob = SAM_ARCH[arch]
    sd = synth_sam_state_dict(arch, seed=12)
    cfg = SamVisionConfig(hidden_size=D, num_hidden_layers=L, num_attention_heads=heads, image_size=M * patch, patch_size=patch, window_size=ws,
                          global_attn_indexes=list(glob), layer_norm_eps=1e-6, hidden_act="gelu", mlp_dim=4 * D, attn_implementation="eager")
    x = torch.randn(2, 16, 16, D, generator=torch.Generator().manual_seed(0))
    for i in (0, 1):                                                           # block 0 windowed, block 1 global
        lyr = SamVisionLayer(cfg, window_size=0 if i in glob else ws).eval()
        p = f"backbone.net.vit.blocks.{i}."
        mapping = {"layer_norm1.weight": "norm1.weight", "layer_norm1.bias": "norm1.bias", "layer_norm2.weight": "norm2.weight",
                   "layer_norm2.bias": "norm2.bias", "attn.qkv.weight": "attn.qkv.weight", "attn.qkv.bias": "attn.qkv.bias",
                   "attn.proj.weight": "attn.proj.weight", "attn.proj.bias": "attn.proj.bias", "attn.rel_pos_h": "attn.rel_pos_h",
                   "attn.rel_pos_w": "attn.rel_pos_w", "mlp.lin1.weight": "mlp.lin1.weight", "mlp.lin1.bias": "mlp.lin1.bias",
                   "mlp.lin2.weight": "mlp.lin2.weight", "mlp.lin2.bias": "mlp.lin2.bias"}
        params = dict(lyr.named_parameters())
        assert set(params) == set(mapping), sorted(params)
        with torch.no_grad():
            for k, src in mapping.items():
                assert params[k].shape == sd[p + src].shape, (k, params[k].shape, sd[p + src].shape)
                params[k].copy_(sd[p + src])
            o = lyr(x)
            y_hf = o[0] if isinstance(o, (tuple, list)) else o
        y = sam_vit.block(x, sd, p, heads, 0 if i in glob else ws)
        assert float((y - y_hf).abs().max() / y_hf.abs().max()) < 2e-5, i
    img = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(1))
    out = sam_vit.sam_backbone_forward(sd, img, heads, L, ws, glob)
    assert out.shape == (1, D, 16, 16) and torch.isfinite(out).all()
