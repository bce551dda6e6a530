"""Seeded synthetic checkpoints with the reference's state_dict key tree.

There is no network and no checkpoint in the build/test environment, so tests and the bench
use random-init weights of the exact architecture. Keys follow the module tree printed at
reference nohup.out:563-684 (``backbone.net.vit.blocks.N...``, ``backbone.simfp_2.0.weight``,
``roi_heads.cube_head.feature_generator.fc1.weight`` ...), which is what
``DetectionCheckpointer`` loads at reference demo/demo.py:148.

Value ranges follow SURVEY.md §8d: N(0, 0.02)-like fan-in scaled weights, LN gamma ~ U(0.5,1.5),
LayerScale gamma ~ U(0.5,1.5)*scale, cube-head output layers std 0.01 so decoded boxes are
non-degenerate.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

VIT_ARCH = {
    # name: (embed dim, depth, heads)   -- reference cubercnn/modeling/backbone/dino.py:17-24
    "vits14": (384, 12, 6),
    "vitb14": (768, 12, 12),
    "vitl14": (1024, 24, 16),
    "vitg14": (1536, 40, 24),
    # tiny architecture for fast CPU tests (not a hub model)
    "vittest14": (128, 2, 2),
    # ViT-L width at depth 2: full-size token / channel geometry (T = 4097 or 5477, D = 1024, 16 heads) at a cost the CPU
    # oracle finishes in seconds (not a hub model)
    "vitl14_d2": (1024, 2, 16),
}


CLIP_ARCH = {
    # open_clip name: (width, layers, heads, patch, pretrained position grid)   -- reference backbone/clip.py:19 (arch), open_clip model configs
    "ViT-B-16": (768, 12, 12, 16, 14),
    "ViT-L-16-d2": (1024, 2, 16, 16, 14),     # ViT-L width at depth 2 (not an open_clip model): wide-tower geometry for tests
    "ViT-test-16": (256, 2, 4, 16, 7),        # tiny (not an open_clip model)
}


MAE_ARCH = {
    # Hugging Face checkpoint: (hidden size, layers, heads, patch)   -- reference backbone/mae.py:22 (facebook/vit-mae-base), ViTMAEConfig
    "facebook/vit-mae-base": (768, 12, 12, 16),
    "test/vit-mae-test": (256, 3, 4, 16),     # tiny (not a published checkpoint)
}


MIDAS_ARCH = {
    # name: (width, layers, heads, patch, pretrained position grid)   -- reference backbone/midas_final.py:23-31 (DPT_Large = timm vit_large_patch16_384)
    "DPT_Large": (1024, 24, 16, 16, 24),
    "DPT_test": (256, 2, 4, 16, 6),           # tiny (not a hub model)
}


SAM_ARCH = {
    # name: (width, layers, heads, patch, checkpoint grid, window, global-attention blocks)   -- reference backbone/sam.py:39-40
    # (segment_anything sam_model_registry['vit_b']: image 1024 -> 64 x 64 grid, window 14, global blocks 2 5 8 11)
    "vit_b": (768, 12, 12, 16, 64, 14, (2, 5, 8, 11)),
    "vit_test": (256, 4, 4, 16, 8, 6, (1, 3)),      # tiny (not a published model)
}


def _lin(g, out_f, in_f, std=None, bias_std=0.02):
    std = (1.0 / math.sqrt(in_f)) if std is None else std
    w = torch.randn(out_f, in_f, generator=g) * std
    b = torch.randn(out_f, generator=g) * bias_std
    return w, b


def synth_state_dict(model_name: str = "vitl14", num_classes: int = 50, fpn_channels: int = 256,
                     fc_dim: int = 1024, pooler_res: int = 7, seed: int = 0,
                     depth_fusion: bool = True, pos_grid: int = 37,
                     num_anchors: int = 3) -> Dict[str, torch.Tensor]:
    if model_name in CLIP_ARCH:
        return synth_clip_state_dict(model_name, num_classes, fpn_channels, fc_dim, pooler_res, seed, num_anchors)
    if model_name in MAE_ARCH:
        return synth_mae_state_dict(model_name, num_classes, fpn_channels, fc_dim, pooler_res, seed, num_anchors)
    if model_name in MIDAS_ARCH:
        return synth_midas_state_dict(model_name, num_classes, fpn_channels, fc_dim, pooler_res, seed, num_anchors)
    if model_name in SAM_ARCH:
        return synth_sam_state_dict(model_name, num_classes, fpn_channels, fc_dim, pooler_res, seed, num_anchors)
    D, L, _ = VIT_ARCH[model_name]
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    V = "backbone.net.vit."
    sd[V + "cls_token"] = torch.randn(1, 1, D, generator=g) * 0.02
    sd[V + "pos_embed"] = torch.randn(1, 1 + pos_grid * pos_grid, D, generator=g) * 0.02
    sd[V + "mask_token"] = torch.zeros(1, D)
    w = torch.randn(D, 3, 14, 14, generator=g) * (1.0 / math.sqrt(588.0))
    sd[V + "patch_embed.proj.weight"] = w
    sd[V + "patch_embed.proj.bias"] = torch.randn(D, generator=g) * 0.02
    for i in range(L):
        B = V + f"blocks.{i}."
        for n in ("norm1", "norm2"):
            sd[B + n + ".weight"] = 0.5 + torch.rand(D, generator=g)
            sd[B + n + ".bias"] = torch.randn(D, generator=g) * 0.05
        sd[B + "attn.qkv.weight"], sd[B + "attn.qkv.bias"] = _lin(g, 3 * D, D, std=2.0 / math.sqrt(D))
        sd[B + "attn.proj.weight"], sd[B + "attn.proj.bias"] = _lin(g, D, D)
        sd[B + "ls1.gamma"] = (0.5 + torch.rand(D, generator=g)) * 0.5
        sd[B + "mlp.fc1.weight"], sd[B + "mlp.fc1.bias"] = _lin(g, 4 * D, D)
        sd[B + "mlp.fc2.weight"], sd[B + "mlp.fc2.bias"] = _lin(g, D, 4 * D)
        sd[B + "ls2.gamma"] = (0.5 + torch.rand(D, generator=g)) * 0.5
    sd[V + "norm.weight"] = torch.ones(D)
    sd[V + "norm.bias"] = torch.zeros(D)
    if depth_fusion:
        w, b = _lin(g, D, D + 1)
        # keep the fusion close to identity-plus-perturbation so features stay well scaled
        sd["backbone.net.depth_fusion.weight"] = w.view(D, D + 1, 1, 1).contiguous()
        sd["backbone.net.depth_fusion.bias"] = b

    _synth_neck_dino(sd, g, D, fpn_channels)
    _synth_heads(sd, g, fpn_channels, num_classes, fc_dim, pooler_res, num_anchors)
    return sd


def _conv(g, cout, cin, k, std=None):
    std = (1.0 / math.sqrt(cin * k * k)) if std is None else std
    return torch.randn(cout, cin, k, k, generator=g) * std


def _ln(sd, g, prefix, C):
    sd[prefix + ".weight"] = 0.5 + torch.rand(C, generator=g)
    sd[prefix + ".bias"] = torch.randn(C, generator=g) * 0.05


def _synth_neck_dino(sd, g, D, C):
    conv = lambda cout, cin, k, std=None: _conv(g, cout, cin, k, std)
    ln = lambda prefix: _ln(sd, g, prefix, C)

    # p2: ConvT(D->D/2,k2,s2,bias) -> 1x1(D/2->C)+LN -> 3x3+LN   (nohup.out:565-575)
    sd["backbone.simfp_2.0.weight"] = torch.randn(D, D // 2, 2, 2, generator=g) * (1.0 / math.sqrt(D))
    sd["backbone.simfp_2.0.bias"] = torch.randn(D // 2, generator=g) * 0.02
    sd["backbone.simfp_2.1.weight"] = conv(C, D // 2, 1)
    ln("backbone.simfp_2.1.norm")
    sd["backbone.simfp_2.2.weight"] = conv(C, C, 3)
    ln("backbone.simfp_2.2.norm")
    # p3: 1x1+LN -> 3x3+LN   (nohup.out:577-585)
    sd["backbone.simfp_3.0.weight"] = conv(C, D, 1)
    ln("backbone.simfp_3.0.norm")
    sd["backbone.simfp_3.1.weight"] = conv(C, C, 3)
    ln("backbone.simfp_3.1.norm")
    # p4: MaxPool -> 1x1+LN -> 3x3+LN   (nohup.out:586-596)
    sd["backbone.simfp_4.1.weight"] = conv(C, D, 1)
    ln("backbone.simfp_4.1.norm")
    sd["backbone.simfp_4.2.weight"] = conv(C, C, 3)
    ln("backbone.simfp_4.2.norm")



def _synth_heads(sd, g, C, num_classes, fc_dim, pooler_res, num_anchors):
    conv = lambda cout, cin, k, std=None: _conv(g, cout, cin, k, std)
    # RPN head (nohup.out:632-639)
    R = "proposal_generator.rpn_head."
    sd[R + "conv.weight"] = conv(C, C, 3)
    sd[R + "conv.bias"] = torch.randn(C, generator=g) * 0.02
    sd[R + "objectness_logits.weight"] = conv(num_anchors, C, 1, std=0.3)
    sd[R + "objectness_logits.bias"] = torch.randn(num_anchors, generator=g) * 0.1
    sd[R + "anchor_deltas.weight"] = conv(4 * num_anchors, C, 1, std=0.02)
    sd[R + "anchor_deltas.bias"] = torch.randn(4 * num_anchors, generator=g) * 0.02

    # box head + predictor (nohup.out:652-662)
    K = C * pooler_res * pooler_res
    H = "roi_heads."
    sd[H + "box_head.fc1.weight"], sd[H + "box_head.fc1.bias"] = _lin(g, fc_dim, K)
    sd[H + "box_head.fc2.weight"], sd[H + "box_head.fc2.bias"] = _lin(g, fc_dim, fc_dim)
    # spread class logits so softmax has a clear winner for a good share of RoIs
    sd[H + "box_predictor.cls_score.weight"], sd[H + "box_predictor.cls_score.bias"] = \
        _lin(g, num_classes + 1, fc_dim, std=0.25)
    sd[H + "box_predictor.bbox_pred.weight"], sd[H + "box_predictor.bbox_pred.bias"] = \
        _lin(g, num_classes * 4, fc_dim, std=0.02)

    # cube head (nohup.out:663-675; init cube_head.py:109-145, widened so outputs are non-degenerate)
    Q = H + "cube_head."
    sd[Q + "feature_generator.fc1.weight"], sd[Q + "feature_generator.fc1.bias"] = _lin(g, fc_dim, K)
    sd[Q + "feature_generator.fc2.weight"], sd[Q + "feature_generator.fc2.bias"] = _lin(g, fc_dim, fc_dim)
    for name, nout, bias0 in (("bbox_3D_dims", 3, 0.0), ("bbox_3D_center_deltas", 2, 0.0),
                              ("bbox_3D_pose", 6, 0.0), ("bbox_3D_center_depth", 1, 2.0),
                              ("bbox_3D_uncertainty", 1, 0.3)):
        w = torch.randn(nout, fc_dim, generator=g) * 0.01
        b = torch.full((nout,), bias0) + torch.randn(nout, generator=g) * 0.01
        sd[Q + name + ".weight"], sd[Q + name + ".bias"] = w, b
    # parameters registered by ROIHeads3D.__init__ (roi_heads.py:118-129); unused with priors disabled
    sd[H + "priors_dims_per_cat"] = torch.ones(1, num_classes, 2, 3)
    sd[H + "priors_z_scales"] = torch.ones(num_classes, 1)


def synth_clip_state_dict(arch: str = "ViT-B-16", num_classes: int = 50, fpn_channels: int = 256, fc_dim: int = 1024,
                          pooler_res: int = 7, seed: int = 0, num_anchors: int = 3) -> Dict[str, torch.Tensor]:
    """Random-init checkpoint with the key tree of the reference's CLIP variant: ``backbone.net.visual.*`` is open_clip's
    VisionTransformer (held as ``self.visual``, reference backbone/clip.py:28; ln_post / proj exist in the module but the dense
    tap never reaches them), ``backbone.simfp_{2..5}`` detectron2's SimpleFeaturePyramid with scale factors (4, 2, 1, 0.5)
    (:155-166), then the same RPN / box / cube heads."""
    D, L, _, P, M = CLIP_ARCH[arch]
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    V = "backbone.net.visual."
    sd[V + "class_embedding"] = torch.randn(D, generator=g) * 0.02
    sd[V + "positional_embedding"] = torch.randn(1 + M * M, D, generator=g) * 0.02
    sd[V + "conv1.weight"] = torch.randn(D, 3, P, P, generator=g) * (1.0 / math.sqrt(3.0 * P * P))
    sd[V + "ln_pre.weight"] = 0.5 + torch.rand(D, generator=g)
    sd[V + "ln_pre.bias"] = torch.randn(D, generator=g) * 0.05
    for i in range(L):
        B = V + f"transformer.resblocks.{i}."
        for n in ("ln_1", "ln_2"):
            sd[B + n + ".weight"] = 0.5 + torch.rand(D, generator=g)
            sd[B + n + ".bias"] = torch.randn(D, generator=g) * 0.05
        sd[B + "attn.in_proj_weight"], sd[B + "attn.in_proj_bias"] = _lin(g, 3 * D, D, std=2.0 / math.sqrt(D))
        sd[B + "attn.out_proj.weight"], sd[B + "attn.out_proj.bias"] = _lin(g, D, D, std=0.5 / math.sqrt(D))
        sd[B + "mlp.c_fc.weight"], sd[B + "mlp.c_fc.bias"] = _lin(g, 4 * D, D)
        sd[B + "mlp.c_proj.weight"], sd[B + "mlp.c_proj.bias"] = _lin(g, D, 4 * D, std=0.5 / math.sqrt(4 * D))
    sd[V + "ln_post.weight"] = torch.ones(D)
    sd[V + "ln_post.bias"] = torch.zeros(D)
    sd[V + "proj"] = torch.randn(D, 512, generator=g) * 0.02
    _synth_neck4(sd, g, D, fpn_channels)
    C = fpn_channels
    _synth_heads(sd, g, C, num_classes, fc_dim, pooler_res, num_anchors)
    return sd


def _synth_neck4(sd, g, D, C):
    """detectron2 SimpleFeaturePyramid with scale factors (4, 2, 1, 0.5): backbone.simfp_2 .. simfp_5."""
    conv = lambda cout, cin, k, std=None: _conv(g, cout, cin, k, std)
    ln = lambda prefix: _ln(sd, g, prefix, C)
    # p2 (scale 4): ConvT(D->D/2) . LN(D/2) . GELU . ConvT(D/2->D/4) . 1x1+LN . 3x3+LN
    sd["backbone.simfp_2.0.weight"] = torch.randn(D, D // 2, 2, 2, generator=g) * (1.0 / math.sqrt(D))
    sd["backbone.simfp_2.0.bias"] = torch.randn(D // 2, generator=g) * 0.02
    sd["backbone.simfp_2.1.weight"] = 0.5 + torch.rand(D // 2, generator=g)
    sd["backbone.simfp_2.1.bias"] = torch.randn(D // 2, generator=g) * 0.05
    sd["backbone.simfp_2.3.weight"] = torch.randn(D // 2, D // 4, 2, 2, generator=g) * (1.0 / math.sqrt(D // 2))
    sd["backbone.simfp_2.3.bias"] = torch.randn(D // 4, generator=g) * 0.02
    sd["backbone.simfp_2.4.weight"] = conv(C, D // 4, 1)
    ln("backbone.simfp_2.4.norm")
    sd["backbone.simfp_2.5.weight"] = conv(C, C, 3)
    ln("backbone.simfp_2.5.norm")
    # p3 (scale 2): ConvT(D->D/2) . 1x1+LN . 3x3+LN
    sd["backbone.simfp_3.0.weight"] = torch.randn(D, D // 2, 2, 2, generator=g) * (1.0 / math.sqrt(D))
    sd["backbone.simfp_3.0.bias"] = torch.randn(D // 2, generator=g) * 0.02
    sd["backbone.simfp_3.1.weight"] = conv(C, D // 2, 1)
    ln("backbone.simfp_3.1.norm")
    sd["backbone.simfp_3.2.weight"] = conv(C, C, 3)
    ln("backbone.simfp_3.2.norm")
    # p4 (scale 1): 1x1+LN . 3x3+LN
    sd["backbone.simfp_4.0.weight"] = conv(C, D, 1)
    ln("backbone.simfp_4.0.norm")
    sd["backbone.simfp_4.1.weight"] = conv(C, C, 3)
    ln("backbone.simfp_4.1.norm")
    # p5 (scale 0.5): MaxPool . 1x1+LN . 3x3+LN
    sd["backbone.simfp_5.1.weight"] = conv(C, D, 1)
    ln("backbone.simfp_5.1.norm")
    sd["backbone.simfp_5.2.weight"] = conv(C, C, 3)
    ln("backbone.simfp_5.2.norm")


def synth_mae_state_dict(checkpoint: str = "facebook/vit-mae-base", num_classes: int = 50, fpn_channels: int = 256, fc_dim: int = 1024,
                         pooler_res: int = 7, seed: int = 0, num_anchors: int = 3) -> Dict[str, torch.Tensor]:
    """Random-init checkpoint with the key tree of the reference's MAE variant: ``backbone.net.vit.*`` is Hugging Face's ViTMAEModel
    (``ViTMAEForPreTraining.from_pretrained(...).vit``, reference backbone/mae.py:27), then the 4-level pyramid and the heads. The stored
    position table is the 224-pixel one; the reference rebuilds it for the canvas (:62-78), so its values never matter."""
    D, L, _, P = MAE_ARCH[checkpoint]
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    V = "backbone.net.vit."
    sd[V + "embeddings.cls_token"] = torch.randn(1, 1, D, generator=g) * 0.02
    sd[V + "embeddings.position_embeddings"] = torch.zeros(1, 1 + 14 * 14, D)
    sd[V + "embeddings.patch_embeddings.projection.weight"] = torch.randn(D, 3, P, P, generator=g) * (1.0 / math.sqrt(3.0 * P * P))
    sd[V + "embeddings.patch_embeddings.projection.bias"] = torch.randn(D, generator=g) * 0.02
    for i in range(L):
        B = V + f"encoder.layer.{i}."
        for n in ("layernorm_before", "layernorm_after"):
            sd[B + n + ".weight"] = 0.5 + torch.rand(D, generator=g)
            sd[B + n + ".bias"] = torch.randn(D, generator=g) * 0.05
        for n in ("query", "key", "value"):
            sd[B + f"attention.attention.{n}.weight"], sd[B + f"attention.attention.{n}.bias"] = _lin(g, D, D, std=2.0 / math.sqrt(D))
        sd[B + "attention.output.dense.weight"], sd[B + "attention.output.dense.bias"] = _lin(g, D, D, std=0.5 / math.sqrt(D))
        sd[B + "intermediate.dense.weight"], sd[B + "intermediate.dense.bias"] = _lin(g, 4 * D, D)
        sd[B + "output.dense.weight"], sd[B + "output.dense.bias"] = _lin(g, D, 4 * D, std=0.5 / math.sqrt(4 * D))
    sd[V + "layernorm.weight"] = torch.ones(D)
    sd[V + "layernorm.bias"] = torch.zeros(D)
    _synth_neck4(sd, g, D, fpn_channels)
    _synth_heads(sd, g, fpn_channels, num_classes, fc_dim, pooler_res, num_anchors)
    return sd


def synth_midas_state_dict(arch: str = "DPT_Large", num_classes: int = 50, fpn_channels: int = 256, fc_dim: int = 1024, pooler_res: int = 7,
                           seed: int = 0, num_anchors: int = 3) -> Dict[str, torch.Tensor]:
    """Random-init checkpoint with the key tree of the reference's MiDaS variant: ``backbone.net.vit.*`` is the timm VisionTransformer MiDaS
    keeps as ``midas.pretrained.model`` (reference backbone/midas_final.py:23-24), then the 4-level pyramid and the heads."""
    D, L, _, P, M = MIDAS_ARCH[arch]
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    V = "backbone.net.vit."
    sd[V + "cls_token"] = torch.randn(1, 1, D, generator=g) * 0.02
    sd[V + "pos_embed"] = torch.randn(1, 1 + M * M, D, generator=g) * 0.02
    sd[V + "patch_embed.proj.weight"] = torch.randn(D, 3, P, P, generator=g) * (1.0 / math.sqrt(3.0 * P * P))
    sd[V + "patch_embed.proj.bias"] = torch.randn(D, generator=g) * 0.02
    for i in range(L):
        B = V + f"blocks.{i}."
        for n in ("norm1", "norm2"):
            sd[B + n + ".weight"] = 0.5 + torch.rand(D, generator=g)
            sd[B + n + ".bias"] = torch.randn(D, generator=g) * 0.05
        sd[B + "attn.qkv.weight"], sd[B + "attn.qkv.bias"] = _lin(g, 3 * D, D, std=2.0 / math.sqrt(D))
        sd[B + "attn.proj.weight"], sd[B + "attn.proj.bias"] = _lin(g, D, D, std=0.5 / math.sqrt(D))
        sd[B + "mlp.fc1.weight"], sd[B + "mlp.fc1.bias"] = _lin(g, 4 * D, D)
        sd[B + "mlp.fc2.weight"], sd[B + "mlp.fc2.bias"] = _lin(g, D, 4 * D, std=0.5 / math.sqrt(4 * D))
    sd[V + "norm.weight"] = torch.ones(D)
    sd[V + "norm.bias"] = torch.zeros(D)
    _synth_neck4(sd, g, D, fpn_channels)
    _synth_heads(sd, g, fpn_channels, num_classes, fc_dim, pooler_res, num_anchors)
    return sd


def synth_sam_state_dict(arch: str = "vit_b", num_classes: int = 50, fpn_channels: int = 256, fc_dim: int = 1024, pooler_res: int = 7,
                         seed: int = 0, num_anchors: int = 3) -> Dict[str, torch.Tensor]:
    """Random-init checkpoint with the key tree of the reference's SAM variant: ``backbone.net.vit.*`` is segment_anything's
    ``ImageEncoderViT`` (``sam.image_encoder``, reference backbone/sam.py:39-40,49; its neck exists but the dense tap never reaches it), then
    the 4-level pyramid and the heads. Relative-position tables are given real values (they are zero-initialised upstream and learned)."""
    D, L, H, P, M, ws, glob = SAM_ARCH[arch]
    dh = D // H
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    V = "backbone.net.vit."
    sd[V + "pos_embed"] = torch.randn(1, M, M, D, generator=g) * 0.02
    sd[V + "patch_embed.proj.weight"] = torch.randn(D, 3, P, P, generator=g) * (1.0 / math.sqrt(3.0 * P * P))
    sd[V + "patch_embed.proj.bias"] = torch.randn(D, generator=g) * 0.02
    for i in range(L):
        B = V + f"blocks.{i}."
        for n in ("norm1", "norm2"):
            sd[B + n + ".weight"] = 0.5 + torch.rand(D, generator=g)
            sd[B + n + ".bias"] = torch.randn(D, generator=g) * 0.05
        sd[B + "attn.qkv.weight"], sd[B + "attn.qkv.bias"] = _lin(g, 3 * D, D, std=2.0 / math.sqrt(D))
        sd[B + "attn.proj.weight"], sd[B + "attn.proj.bias"] = _lin(g, D, D, std=0.5 / math.sqrt(D))
        side = M if i in glob else ws
        sd[B + "attn.rel_pos_h"] = torch.randn(2 * side - 1, dh, generator=g) * 0.05
        sd[B + "attn.rel_pos_w"] = torch.randn(2 * side - 1, dh, generator=g) * 0.05
        sd[B + "mlp.lin1.weight"], sd[B + "mlp.lin1.bias"] = _lin(g, 4 * D, D)
        sd[B + "mlp.lin2.weight"], sd[B + "mlp.lin2.bias"] = _lin(g, D, 4 * D, std=0.5 / math.sqrt(4 * D))
    sd[V + "neck.0.weight"] = torch.randn(256, D, 1, 1, generator=g) * 0.02
    _synth_neck4(sd, g, D, fpn_channels)
    _synth_heads(sd, g, fpn_channels, num_classes, fc_dim, pooler_res, num_anchors)
    return sd
