"""Host logic: config tree, containers, registries, sharding, pos-embed interpolation, engine guards."""
import numpy as np
import pytest
import torch

from common import build_cfg
from ovmono3d_amd.config import CfgNode, get_cfg, get_cfg_defaults
from ovmono3d_amd.defaults import make_cfg


def test_config_base_inheritance_and_overrides():
    cfg = make_cfg("OVMono3D_dinov2_SFP.yaml", ["MODEL.ROI_HEADS.NAME", "ROIHeads3DGDINO", "MODEL.FPN.SQUARE_PAD", "518"])
    assert cfg.MODEL.DINO.MODEL_NAME == "vitb14"                      # reference OVMono3D_dinov2_SFP.yaml:30
    assert cfg.MODEL.FPN.SQUARE_PAD == 518
    assert cfg.MODEL.ROI_HEADS.NAME == "ROIHeads3DGDINO"              # CLI override, reference README.md:59
    assert cfg.MODEL.ROI_HEADS.NUM_CLASSES == 50
    assert cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST == 0.01              # from Base.yaml through _BASE_
    assert cfg.MODEL.ROI_CUBE_HEAD.DIMS_PRIORS_ENABLED is False
    assert cfg.MODEL.PIXEL_MEAN == [123.675, 116.28, 103.53]
    assert cfg.MODEL.ANCHOR_GENERATOR.SIZES == [[64], [256], [512]]
    assert cfg.INPUT.MIN_SIZE_TEST == 532 and cfg.INPUT.MAX_SIZE_TEST == 896
    assert cfg.TEST.DETECTIONS_PER_IMAGE == 100
    with pytest.raises(AttributeError):
        cfg.MODEL.DEVICE = "cpu"                                      # frozen
    with pytest.raises(KeyError):
        make_cfg(None, ["MODEL.NOPE", 1])
    vl = make_cfg("OVMono3D_dinov2L_SFP.yaml")
    assert vl.MODEL.DINO.MODEL_NAME == "vitl14" and vl.MODEL.FPN.SQUARE_PAD == 896


def test_config_to_native_validation():
    from ovmono3d_amd.native import config_to_native
    c = config_to_native(build_cfg("vitl14", 896))
    assert (c.embed_dim, c.depth, c.heads, c.canvas, c.precision) == (1024, 24, 16, 896, 3)
    assert list(c.anchor_sizes)[:3] == [64.0, 256.0, 512.0] and c.rpn_pre_topk == 1000 and c.tower == 0
    from common import build_clip_cfg
    k = config_to_native(build_clip_cfg("ViT-B-16", 1024))
    assert (k.tower, k.embed_dim, k.depth, k.heads, k.canvas, k.pos_grid) == (1, 768, 12, 12, 1024, 14)
    assert list(k.anchor_sizes) == [64.0, 128.0, 256.0, 512.0] and (k.pooler_min_level, k.pooler_max_level) == (2, 5)
    assert k.use_depth_fusion == 0
    with pytest.raises(ValueError):
        config_to_native(build_clip_cfg("ViT-B-16", 1000))             # not a multiple of 16
    from common import build_mae_cfg
    m = config_to_native(build_mae_cfg("facebook/vit-mae-base", 1024))
    assert (m.tower, m.embed_dim, m.depth, m.heads, m.pooler_max_level) == (2, 768, 11, 12, 5)      # 11 of the 12 blocks run (mae.py:43-55)
    from common import build_midas_cfg
    d = config_to_native(build_midas_cfg("DPT_Large", 1024))
    assert (d.tower, d.embed_dim, d.depth, d.heads, d.pos_grid, d.pooler_max_level) == (3, 1024, 24, 16, 24, 5)
    from common import build_sam_cfg
    a = config_to_native(build_sam_cfg("vit_b", 1024))
    assert (a.tower, a.embed_dim, a.depth, a.heads, a.pos_grid, a.sam_window, a.sam_global_mask) == (4, 768, 12, 12, 64, 14, 0b100100100100)
    with pytest.raises(ValueError):
        config_to_native(build_cfg(extra=["MODEL.BACKBONE.NAME", "build_dla_from_vision_fpn_backbone"]))
    with pytest.raises(ValueError):
        config_to_native(build_cfg("vitl14", 900))                    # not a multiple of 14
    with pytest.raises(ValueError):
        config_to_native(build_cfg(extra=["MODEL.ROI_CUBE_HEAD.Z_TYPE", "log"]))
    with pytest.raises(ValueError):
        config_to_native(build_cfg(extra=["MODEL.AMD.GEMM_PRECISION", "bf16"]))


def test_engine_refuses_cpu_device():
    from ovmono3d_amd.native import Engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine(build_cfg(), device="cpu")


def test_registries_hold_reference_names():
    import ovmono3d_amd.modeling  # noqa: F401
    from ovmono3d_amd import registry as R
    assert "build_dino_backbone" in R.BACKBONE_REGISTRY
    assert "RCNN3D" in R.META_ARCH_REGISTRY
    assert "ROIHeads3D" in R.ROI_HEADS_REGISTRY and "ROIHeads3DGDINO" in R.ROI_HEADS_REGISTRY
    assert "RPNWithIgnore" in R.PROPOSAL_GENERATOR_REGISTRY
    assert "CubeHead" in R.ROI_CUBE_HEAD_REGISTRY
    with pytest.raises(KeyError):
        R.ROI_HEADS_REGISTRY.get("StandardROIHeads")


def test_instances_semantics():
    from ovmono3d_amd.structures import Boxes, Instances
    i = Instances((10, 20))
    i.pred_boxes = Boxes(torch.arange(12.0).view(3, 4))
    i.scores = torch.tensor([0.1, 0.9, 0.5])
    assert len(i) == 3 and i.has("scores") and not i.has("pred_bbox3D")
    j = i[torch.tensor([True, False, True])]
    assert len(j) == 2 and j.image_size == (10, 20) and j.scores.tolist() == pytest.approx([0.1, 0.5])
    with pytest.raises(AssertionError):
        i.bad = torch.zeros(2)
    assert len(Boxes(torch.zeros(0))) == 0


def test_shard_range_matches_inference_sampler():
    """reference cubercnn/data/build.py:320 (InferenceSampler): 9314 images on 2 ranks -> 4657 + 4657
    (reference nohup.out:704)."""
    from ovmono3d_amd import lib
    assert lib.shard_range(9314, 0, 2) == (0, 4657) and lib.shard_range(9314, 1, 2) == (4657, 9314)
    n, w = 23, 8
    cover = []
    for r in range(w):
        b, e = lib.shard_range(n, r, w)
        assert e - b == n // w + (1 if r < n % w else 0)
        cover += list(range(b, e))
    assert cover == list(range(n))
    assert lib.shard_range(3, 7, 8) == (3, 3)                         # more ranks than items -> empty shard


@pytest.mark.parametrize("G", [64, 37, 16, 74])
def test_pos_embed_interpolation_matches_torch(G):
    from oracle.vit import interpolate_pos_encoding
    from ovmono3d_amd import lib
    pos = torch.randn(1, 1 + 37 * 37, 48, generator=torch.Generator().manual_seed(G))
    ref = interpolate_pos_encoding(pos, G, G)[0].numpy()
    got = lib.interp_pos_embed(pos[0].numpy(), G)
    assert np.abs(ref - got).max() < 5e-5


def test_gdino_head_requires_category_list_and_detector():
    """Error behaviour of the reference: NameError without category_list (roi_heads_gdino.py:130-134)."""
    from ovmono3d_amd.modeling.roi_heads.roi_heads_gdino import ROIHeads3DGDINO
    from ovmono3d_amd.structures import ImageList

    class FakeEngine:
        device = torch.device("cpu")
    h = ROIHeads3DGDINO.__new__(ROIHeads3DGDINO)
    h.training, h.detector, h.loss_w_3d, h.engine = False, None, 1.0, FakeEngine()
    h._gdino_cfg = build_cfg()                       # default GDINO_WEIGHTS path does not exist here -> loud failure
    il = ImageList(None, [(10, 10)])
    with pytest.raises(NameError):
        h.forward(il, {}, None, [], [1.0], None, category_list=None)
    with pytest.raises(NotImplementedError):
        h.forward(il, {}, None, [], [1.0], None, category_list=["chair"])


def test_wordpiece_tokenizer_and_caption(tmp_path):
    from ovmono3d_amd.modeling.roi_heads.gdino_glue import WordPieceTokenizer, build_caption, phrase_spans
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", ".", "chair", "din", "##ing", "table", "cereal", "box", "shoe", "##s"]
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(vocab) + "\n")
    tok = WordPieceTokenizer(str(vf))
    caption, caps = build_caption(["Chair", "dining table", "cereal box", "shoes"])
    assert caption == "chair . dining table . cereal box . shoes ."           # reference roi_heads_gdino.py:176-181
    assert tok.tokenize(caption) == ["chair", ".", "din", "##ing", "table", ".", "cereal", "box", ".", "shoe", "##s", "."]
    ids = tok.encode(caption)
    phrases = [tok.encode(c.lower(), add_special_tokens=False) for c in caps]
    assert phrase_spans(ids, phrases) == [(1, 2), (3, 6), (7, 9), (10, 12)]
    assert tok.tokenize("zebra") == ["[UNK]"]
    with pytest.raises(AssertionError):
        phrase_spans(ids, [[5], [99]])


@pytest.mark.parametrize("M,G", [(14, 64), (14, 16), (7, 16), (14, 10), (14, 7), (14, 14)])
def test_clip_pos_embed_resize_matches_torch_antialiased_bicubic(M, G):
    """ovm_host_resize_pos_embed_aa == F.interpolate(size, bicubic, align_corners=False, antialias=True) on the patch rows
    (reference clip.py:98-133), for up-sampling (the shipped case 14 -> 64), down-sampling and the no-op."""
    from oracle.clip_vit import resize_pos_embed
    from ovmono3d_amd import lib
    L = lib.load()
    D = 24
    pos = torch.randn(1 + M * M, D, generator=torch.Generator().manual_seed(M * 100 + G))
    ref = resize_pos_embed(pos, (G, G))
    out = np.empty((1 + G * G, D), np.float32)
    src = np.ascontiguousarray(pos.numpy())
    assert L.ovm_host_resize_pos_embed_aa(src.ctypes.data, M, D, G, out.ctypes.data) == 0
    assert np.abs(out - ref.numpy()).max() < 2e-6
    assert np.array_equal(out[0], src[0])                                     # class row untouched
