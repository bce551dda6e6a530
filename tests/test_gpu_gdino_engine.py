"""GPU parity of libovm3d's C++ GroundingDINO engine (ovm_gdino_forward: the whole network behind one C-ABI call, no Python op
sequencing) against the Hugging Face port, at test size and at the real Swin-B / BERT-base / 900-query size, plus the plan /
HIP-graph machinery (replay == eager; plans of different shapes own their scratch). Parity vs upstream itself is unpinned."""
import time

import pytest
import torch

from common import assert_close
from hf_gdino_patches import patch_hf_to_upstream
from test_gpu_gdino import _small_hf_gdino

pytestmark = pytest.mark.gpu

MEAN, STD = [103.53, 116.28, 123.675], [57.375, 57.12, 58.395]
SMALL = dict(d_model=64, enc_layers=2, dec_layers=2, heads=4, ffn_dim=128, num_queries=30, bert_heads=2, swin_embed=32,
             swin_depths=(2, 2, 2, 2), swin_heads=(1, 2, 4, 8), swin_window=12)


def _normalised(img_u8):
    """what the engine feeds the network: (x - mean) / std per channel, channels flipped (roi_heads_gdino.py:146)"""
    mean, std = torch.tensor(MEAN).view(3, 1, 1), torch.tensor(STD).view(3, 1, 1)
    return ((img_u8.float() - mean) / std)[[2, 1, 0]]


def _engine(device, sd, cfg_kwargs, **kw):
    from ovmono3d_amd.gdino.engine import GdinoEngine
    from ovmono3d_amd.gdino.config import GDinoConfig
    return GdinoEngine(device, sd, GDinoConfig(**cfg_kwargs), pixel_mean=MEAN, pixel_std=STD, flip_channels=True, **kw)


def test_engine_matches_hf_small(device):
    from transformers.models.grounding_dino.modeling_grounding_dino import generate_masks_with_special_tokens_and_transfer_map
    hf, _ = _small_hf_gdino()
    H, W = 96, 132
    g = torch.Generator().manual_seed(2)
    img = torch.randint(0, 256, (3, H, W), dtype=torch.uint8, generator=g)
    ids = torch.tensor([101, 500, 1012, 600, 601, 1012, 700, 701, 702, 1012, 102])
    with torch.no_grad():
        out = hf(pixel_values=_normalised(img)[None], input_ids=ids[None], return_dict=True)
    _, p_hf = generate_masks_with_special_tokens_and_transfer_map(ids[None])
    eng = _engine(device, hf.state_dict(), SMALL, use_graphs=False)
    logits, boxes = eng.forward(img.to(device), ids.tolist(), p_hf[0].tolist())
    T, S = len(ids), out.encoder_last_hidden_state_vision.shape[1]
    assert_close(eng.debug("enc_text", (T, 64)), out.encoder_last_hidden_state_text[0], 1e-4, "encoder text")
    assert_close(eng.debug("enc_vision", (S, 64)), out.encoder_last_hidden_state_vision[0], 1e-4, "encoder vision")
    hf_topk = torch.topk(out.enc_outputs_class[0].max(-1)[0], 30)[1]
    assert sorted(eng.debug("topk", (30,), torch.int32).cpu().tolist()) == sorted(hf_topk.tolist())
    assert torch.isinf(logits[:, T:]).all() and (logits[:, T:] < 0).all()
    assert_close(boxes, out.pred_boxes[0], 2e-4, "pred_boxes")
    assert_close(logits[:, :T], out.logits[0][:, :T], 2e-4, "pred_logits")
    # default position ids = upstream numbering (phrase-relative, delimiter included): differs from HF's ids, runs, stays finite
    l2, b2 = eng.forward(img.to(device), ids.tolist())
    assert torch.isfinite(b2).all() and torch.isfinite(l2[:, :T]).all()


def test_engine_matches_python_sequenced_path_small(device):
    """Same weights through the round-1 Python-sequenced generic-op path (itself checked against HF): intermediate taps agree."""
    from pyref_gdino.model import GDinoConfig, GroundingDinoNative
    from pyref_gdino.ops import Ops
    hf, _ = _small_hf_gdino()
    H, W = 100, 130                                             # not multiples of the patch / window sizes
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (3, H, W), dtype=torch.uint8, generator=g)
    ids = [101, 500, 1012, 600, 601, 1012, 102]
    eng = _engine(device, hf.state_dict(), SMALL, use_graphs=False)
    logits, boxes = eng.forward(img.to(device), ids)
    net = GroundingDinoNative(Ops(device, 3), hf.state_dict(), GDinoConfig(**SMALL))
    x = _normalised(img).permute(1, 2, 0).reshape(H * W, 3).contiguous().to(device)
    lg, bx, aux = net.forward(x, H, W, torch.tensor(ids), return_aux=True)
    T = len(ids)
    assert_close(eng.debug("text_features", (T, 64)), aux["text_features"], 2e-5, "text features")
    assert_close(eng.debug("source_flatten", tuple(aux["source_flatten"].shape)), aux["source_flatten"], 5e-5, "projected image features")
    assert_close(eng.debug("enc_vision", tuple(aux["enc_vision"].shape)), aux["enc_vision"], 5e-5, "encoder vision")
    assert_close(eng.debug("enc_text", (T, 64)), aux["enc_text"], 5e-5, "encoder text")
    assert sorted(eng.debug("topk", (30,), torch.int32).cpu().tolist()) == sorted(aux["topk"].cpu().tolist())
    assert_close(boxes, bx, 2e-4, "pred_boxes")
    assert_close(logits[:, :T], lg[:, :T], 2e-4, "pred_logits")


def test_engine_graph_replay_and_plans_own_their_scratch(device):
    """Each (image size, caption) plan owns its arena, split-K workspace and sort keys and is captured into a HIP graph after its
    first run. Replays on new pixels equal the eager forward bit for bit; running a LARGER shape in between (which re-sized the
    process-global scratch of the round-1 path under captured graphs) does not disturb a smaller plan's replay."""
    hf, _ = _small_hf_gdino()
    sd = hf.state_dict()
    eager = _engine(device, sd, SMALL, use_graphs=False)
    graphed = _engine(device, sd, SMALL, use_graphs=True)
    g = torch.Generator().manual_seed(3)
    ids = [101, 500, 1012, 600, 601, 1012, 102]
    shapes = [(96, 132), (96, 132), (160, 200), (96, 132), (160, 200), (96, 132)]
    for i, (h, w) in enumerate(shapes):
        im = torch.randint(0, 256, (3, h, w), dtype=torch.uint8, generator=g).to(device)
        a_l, a_b = eager.forward(im, ids)
        b_l, b_b = graphed.forward(im, ids)
        assert torch.equal(a_b, b_b) and torch.equal(a_l, b_l), (i, h, w)
    other = [101, 700, 701, 1012, 102]                            # another caption on a known shape: its own plan
    im = torch.randint(0, 256, (3, 96, 132), dtype=torch.uint8, generator=g).to(device)
    assert torch.equal(eager.forward(im, other)[1], graphed.forward(im, other)[1])


def test_engine_text_branch_on_its_own_stream_is_bit_identical(device):
    """The text side of the network (BERT, then each encoder layer's text enhancer) runs on a second stream / graph branch beside
    the image side (Swin + neck, deformable attention). Same kernels, same operands, another schedule: outputs must equal the
    single-stream order bit for bit, eagerly and under graph replay, on repeated calls with new pixels."""
    from ovmono3d_amd import lib
    L = lib.load()
    hf, _ = _small_hf_gdino()
    sd = hf.state_dict()
    try:
        L.ovm_tune_set(b"gdino_branches", 0)
        single = _engine(device, sd, SMALL, use_graphs=False)
        L.ovm_tune_set(b"gdino_branches", 1)
        two_eager = _engine(device, sd, SMALL, use_graphs=False)
        two_graph = _engine(device, sd, SMALL, use_graphs=True)
    finally:
        L.ovm_tune_set(b"gdino_branches", 1)
    g = torch.Generator().manual_seed(9)
    ids = [101, 500, 1012, 600, 601, 1012, 700, 701, 702, 1012, 102]
    for i, (h, w) in enumerate([(96, 132), (96, 132), (160, 200), (96, 132), (96, 132)]):
        im = torch.randint(0, 256, (3, h, w), dtype=torch.uint8, generator=g).to(device)
        a_l, a_b = single.forward(im, ids)
        for eng in (two_eager, two_graph):
            b_l, b_b = eng.forward(im, ids)
            assert torch.equal(a_b, b_b) and torch.equal(a_l, b_l), (i, h, w)
    T = len(ids)
    assert torch.equal(single.debug("enc_text", (T, 64)), two_graph.debug("enc_text", (T, 64)))


def test_engine_full_size_swinb_matches_hf(device):
    """The real architecture (Swin-B 384 / window 12, BERT-base, 6 + 6 layers, 900 queries) at a non-square network resolution,
    random weights; HF runs in fp32 on the same GPU. Also prints the engine's kernel launches per forward and its time."""
    from transformers.models.grounding_dino.modeling_grounding_dino import generate_masks_with_special_tokens_and_transfer_map
    from ovmono3d_amd.gdino.detector import HashTokenizer
    from synth_gdino import synth_gdino_model
    hf, sd = synth_gdino_model(5)
    patch_hf_to_upstream(hf)
    hf = hf.to(device)
    H, W = 532, 708
    g = torch.Generator().manual_seed(2)
    img = torch.randint(0, 256, (3, H, W), dtype=torch.uint8, generator=g)
    ids = torch.tensor(HashTokenizer().encode("chair . dining table . sofa . potted plant . television . bookcase ."))
    with torch.no_grad():
        out = hf(pixel_values=_normalised(img)[None].to(device), input_ids=ids[None].to(device), return_dict=True)
    _, p_hf = generate_masks_with_special_tokens_and_transfer_map(ids[None])
    eng = _engine(device, sd, {}, use_graphs=True)
    imd = img.to(device)
    eng.forward(imd, ids.tolist(), p_hf[0].tolist())
    T, S = len(ids), out.encoder_last_hidden_state_vision.shape[1]
    assert_close(eng.debug("enc_vision", (S, 256)), out.encoder_last_hidden_state_vision[0], 1e-4, "encoder vision")
    assert_close(eng.debug("enc_text", (T, 256)), out.encoder_last_hidden_state_text[0], 1e-4, "encoder text")
    sc = out.enc_outputs_class[0].max(-1)[0]
    hf_topk = torch.topk(sc, 900)[1]
    mine, theirs = eng.debug("topk", (900,), torch.int32).cpu().long(), hf_topk.cpu()
    assert sorted(mine.tolist()) == sorted(theirs.tolist())
    swapped = mine != theirs
    assert int(swapped.sum()) <= 20
    assert ((sc[mine[swapped].to(device)] - sc[theirs[swapped].to(device)]).abs() <= 1e-4 * sc.abs().max()).all()
    eng.set_force_topk(hf_topk)                                  # pin HF's order of near-ties for the decoder comparison
    logits, boxes = eng.forward(imd, ids.tolist(), p_hf[0].tolist())
    assert_close(boxes, out.pred_boxes[0], 3e-4, "pred_boxes")
    assert_close(logits[:, :T], out.logits[0][:, :T], 3e-4, "pred_logits")
    eng.set_force_topk(None)
    for _ in range(3):
        eng.forward(imd, ids.tolist(), p_hf[0].tolist())
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        eng.forward(imd, ids.tolist(), p_hf[0].tolist())
    torch.cuda.synchronize()
    print(f"C++ GroundingDINO engine, Swin-B {H}x{W}: {(time.time() - t0) / 10 * 1e3:.2f} ms per forward (graph replay), "
          f"{eng.launches()} kernel launches per forward")


@pytest.mark.parametrize("tower", ["dinov2", "clip"])
def test_ovm_infer_one_call_equals_staged_path(device, tower):
    """ovm_infer (the whole text-prompted path of one image behind ONE C-ABI call - backbone, GroundingDINO engine on
    an internal side stream, output glue, cube head, postprocess) against the same stages sequenced by the Python host
    (ROIHeads3DGDINO.prefetch / forward): identical records. Replaces reference rcnn3d.py:79-117 with category_list."""
    from common import build_cfg, build_clip_cfg, synth_inputs
    from ovmono3d_amd.gdino.detector import HashTokenizer, NativeGroundingDino
    from ovmono3d_amd.gdino.config import GDinoConfig
    from ovmono3d_amd.modeling import build_model
    from ovmono3d_amd.util.synth_weights import synth_state_dict
    hf, _ = _small_hf_gdino()

    class Tok(HashTokenizer):
        def _id(self, w):
            return super()._id(w) % 1900 + 50 if w not in (".", "?") else super()._id(w)
    sd = synth_state_dict("vittest14" if tower == "dinov2" else "ViT-test-16", seed=3)
    outs = []
    for fused in (True, False):
        if tower == "dinov2":
            cfg = build_cfg("vittest14", 280, "f16x3", max_batch=1, max_rois=64, roi_heads="ROIHeads3DGDINO", extra=["MODEL.AMD.FUSED_INFER", fused])
        else:                                                    # the 4-level towers go through the same one-call path
            cfg = build_clip_cfg("ViT-test-16", 288, "f16x3", max_batch=1, max_rois=64, roi_heads="ROIHeads3DGDINO",
                                 extra=["MODEL.AMD.FUSED_INFER", fused])
        model = build_model(cfg, device=device)
        model.load_state_dict(sd)
        model.roi_heads.detector = NativeGroundingDino(device, hf.state_dict(), Tok(), cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD,
                                                       cfg=GDinoConfig(**SMALL))
        res = []
        for seed in (5, 6, 5):                                   # second and third call replay the detector's graph
            inp = synth_inputs(1, hw=((210, 280),), oracle2d=False, seed=seed)
            inp[0]["category_list"] = ["chair", "dining table", "sofa"]
            inp[0]["image"] = inp[0]["image"].to(device)
            res.append(model(inp)[0]["instances"])
        outs.append(res)
    for a, b in zip(*outs):
        assert len(a) == len(b) and len(a) >= 5
        assert torch.equal(a.pred_classes.cpu(), b.pred_classes.cpu())
        for f in ("scores", "pred_bbox3D", "pred_center_cam", "pred_center_2D", "pred_dimensions", "pred_pose"):
            assert torch.equal(a.get(f), b.get(f)), f
        assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor)
    assert torch.equal(outs[0][0].pred_bbox3D, outs[0][2].pred_bbox3D)      # same input -> same output across graph replays


def test_ovm_infer_error_after_fork_leaves_the_handle_usable(device):
    """An error behind ovm_infer's fork (here: an image larger than the backbone's canvas, refused by ovm_backbone_forward while the
    detector already runs on the side stream) must drain both streams before it returns - the caller may free the image right away -
    and must leave the handle usable: the next call gives the same records as a handle that never saw the error."""
    from common import build_cfg, synth_inputs
    from ovmono3d_amd.gdino.detector import HashTokenizer, NativeGroundingDino
    from ovmono3d_amd.gdino.engine import GDinoConfig
    from ovmono3d_amd.lib import OvmError
    from ovmono3d_amd.modeling import build_model
    from ovmono3d_amd.util.synth_weights import synth_state_dict
    hf, _ = _small_hf_gdino()

    class Tok(HashTokenizer):
        def _id(self, w):
            return super()._id(w) % 1900 + 50 if w not in (".", "?") else super()._id(w)
    sd = synth_state_dict("vittest14", seed=3)
    cfg = build_cfg("vittest14", 280, "f16x3", max_batch=1, max_rois=64, roi_heads="ROIHeads3DGDINO")

    def make():
        m = build_model(cfg, device=device)
        m.load_state_dict(sd)
        m.roi_heads.detector = NativeGroundingDino(device, hf.state_dict(), Tok(), cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, cfg=GDinoConfig(**SMALL))
        return m

    def inp(hw, seed):
        d = synth_inputs(1, hw=(hw,), oracle2d=False, seed=seed)
        d[0]["category_list"] = ["chair", "dining table", "sofa"]
        d[0]["image"] = d[0]["image"].to(device)
        return d
    clean, hit = make(), make()
    want = clean(inp((210, 280), 5))[0]["instances"]
    for _ in range(2):
        big = inp((294, 350), 9)                                  # wider than the 280 canvas: the detector takes it, the backbone refuses
        with pytest.raises(OvmError):
            hit(big)
        del big                                                   # the image may be freed as soon as the call has returned
        got = hit(inp((210, 280), 5))[0]["instances"]
        assert len(got) == len(want) >= 5
        for f in ("scores", "pred_bbox3D", "pred_pose"):
            assert torch.equal(got.get(f), want.get(f)), f


def test_vectorised_deformable_sampling_is_bit_identical(device):
    """msdeform_fused4_kernel (round 3: one thread per query x head x FOUR channels, the softmax over the 16 sample logits evaluated once
    into registers, float4 taps) against the one-thread-per-channel kernel it replaces (ovm_tune_set msdeform_vec = 0): every output
    element sees the same operations in the same order, so the whole forward - encoder memories, selected proposals, logits, boxes -
    must agree bit for bit (encoder mode: reference points; decoder mode: reference boxes)."""
    from ovmono3d_amd import lib
    L = lib.load()
    hf, _ = _small_hf_gdino()
    g = torch.Generator().manual_seed(7)
    img = torch.randint(0, 256, (3, 120, 168), dtype=torch.uint8, generator=g).to(device)
    ids = [101, 500, 1012, 600, 601, 1012, 102]
    outs = []
    try:
        for vec in (0, 1):
            assert L.ovm_tune_set(b"msdeform_vec", vec) == 0
            eng = _engine(device, hf.state_dict(), SMALL, use_graphs=False)
            logits, boxes = eng.forward(img, ids)
            outs.append((logits.clone(), boxes.clone(), eng.debug("topk", (SMALL["num_queries"],), torch.int32).clone()))
            del eng
    finally:
        L.ovm_tune_set(b"msdeform_vec", 1)
    from ovmono3d_amd.gdino.config import GDinoConfig
    cfg = GDinoConfig(**SMALL)
    assert cfg.n_levels * cfg.n_points == 16, "the vectorised kernel covers L * P = 16 (else this test compares a kernel with itself)"
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("size", ["small", "full"])
def test_decoder_row_chain_matches_launch_per_op_decoder(device, size):
    """dec_chain.hip (round 3): a decoder layer as two row-chain kernels around the query self-attention - 16 query rows per workgroup
    resident in LDS through fourteen linears, four LayerNorms, the text cross-attention, the deformable sampling and the box update -
    against the launch-per-op sequence it replaces (ovm_tune_set gdino_dec_chain = 0; itself checked against the HF port above).
    Same operands, same three-pass products, fp32 everywhere else; what differs is summation order inside LayerNorm / softmax /
    the FFN's chunked second layer: logits and boxes within 2e-5, the launch count drops by ~30 per layer."""
    from ovmono3d_amd import lib
    L = lib.load()
    g = torch.Generator().manual_seed(11)
    if size == "small":
        hf, _ = _small_hf_gdino()
        sd, cfgk, hw = hf.state_dict(), SMALL, (120, 168)
        ids = [101, 500, 1012, 600, 601, 1012, 102]
    else:
        from ovmono3d_amd.util.synth_gdino_weights import synth_gdino_state_dict
        sd, cfgk, hw = synth_gdino_state_dict(3), {}, (532, 709)
        ids = [101, 2000 + 17, 1012, 2000 + 29, 2000 + 31, 1012, 2000 + 5, 1012, 102]
    img = torch.randint(0, 256, (3,) + hw, dtype=torch.uint8, generator=g).to(device)
    outs, launches = [], []
    try:
        for chain in (0, 1):
            assert L.ovm_tune_set(b"gdino_dec_chain", chain) == 0
            eng = _engine(device, sd, cfgk, use_graphs=False)
            logits, boxes = eng.forward(img, ids)
            outs.append((logits[:, :len(ids)].clone(), boxes.clone()))
            launches.append(eng.launches())
            del eng
    finally:
        L.ovm_tune_set(b"gdino_dec_chain", 1)
    print(f"decoder row chain ({size}): {launches[0]} -> {launches[1]} kernel launches per forward")
    assert launches[1] <= launches[0] - 15 * (2 if size == "small" else 6)          # (op wrappers counted, split-K reduce launches not included)
    assert torch.isfinite(outs[1][0]).all() and torch.isfinite(outs[1][1]).all()
    assert_close(outs[1][1], outs[0][1], 2e-5, "pred_boxes (row chain vs launch per op)")
    assert_close(outs[1][0], outs[0][0], 2e-5, "pred_logits (row chain vs launch per op)")


def test_wide_contractions_on_256_tiles_match_128_tiles(device):
    """Round 3: the detector's wide contractions over K <= 256 (encoder vision q|v and FFN up-projection, decoder value projection, Swin
    stage 1 FFN up-projection) run on the 256 x 256 GEMM, their LayerNorm writing interleaved split rows (Run::split_for256). Against the
    128-tile route (ovm_tune_set gdino_gemm256 = 0) on the full-size detector: same three-pass products, fp32 accumulation; the k-order
    inside a tile and the old route's split-K differ - logits and boxes to 2e-5."""
    from ovmono3d_amd import lib
    from ovmono3d_amd.util.synth_gdino_weights import synth_gdino_state_dict
    L = lib.load()
    g = torch.Generator().manual_seed(13)
    sd, hw = synth_gdino_state_dict(3), (532, 532)
    ids = [101, 2000 + 17, 1012, 2000 + 29, 2000 + 31, 1012, 2000 + 5, 1012, 102]
    img = torch.randint(0, 256, (3,) + hw, dtype=torch.uint8, generator=g).to(device)
    outs = []
    try:
        for val in (0, 1):
            assert L.ovm_tune_set(b"gdino_gemm256", val) == 0
            eng = _engine(device, sd, {}, use_graphs=False)
            logits, boxes = eng.forward(img, ids)
            outs.append((logits[:, :len(ids)].clone(), boxes.clone()))
            del eng
    finally:
        L.ovm_tune_set(b"gdino_gemm256", 1)
    assert torch.isfinite(outs[1][0]).all() and torch.isfinite(outs[1][1]).all()
    assert_close(outs[1][1], outs[0][1], 2e-5, "pred_boxes (256 x 256 tiles vs 128 x 128)")
    assert_close(outs[1][0], outs[0][0], 2e-5, "pred_logits (256 x 256 tiles vs 128 x 128)")


@pytest.mark.parametrize("size", ["small", "full"])
def test_decoder_ffn_chunks_on_separate_workgroups_is_bit_identical(device, size):
    """Round 3: chain B of a decoder layer runs as (row blocks) x (FFN chunks of 512 hidden columns) - every workgroup of a row block
    repeats the front of the layer and computes one chunk's partial second-layer product - and chain C adds the partials to
    (x + bias) in chunk order, the accumulation order of the one-workgroup form, then finishes the layer. Nothing else changes, so
    logits and boxes must agree in every bit with ovm_tune_set gdino_ffn_split = 0."""
    from ovmono3d_amd import lib
    L = lib.load()
    g = torch.Generator().manual_seed(17)
    if size == "small":
        hf, _ = _small_hf_gdino()
        sd, cfgk, hw = hf.state_dict(), SMALL, (120, 168)
        ids = [101, 500, 1012, 600, 601, 1012, 102]
    else:
        from ovmono3d_amd.util.synth_gdino_weights import synth_gdino_state_dict
        sd, cfgk, hw = synth_gdino_state_dict(3), {}, (532, 620)
        ids = [101, 2000 + 17, 1012, 2000 + 29, 2000 + 31, 1012, 2000 + 5, 1012, 102]
    img = torch.randint(0, 256, (3,) + hw, dtype=torch.uint8, generator=g).to(device)
    outs, launches = [], []
    try:
        for split in (0, 1):
            assert L.ovm_tune_set(b"gdino_ffn_split", split) == 0
            eng = _engine(device, sd, cfgk, use_graphs=False)
            logits, boxes = eng.forward(img, ids)
            outs.append((logits.clone(), boxes.clone()))
            launches.append(eng.launches())
            del eng
    finally:
        L.ovm_tune_set(b"gdino_ffn_split", 0)
    from ovmono3d_amd.gdino.config import GDinoConfig
    cfg = GDinoConfig(**cfgk)
    if cfg.ffn_dim > 512:
        assert launches[1] == launches[0] + cfg.dec_layers, launches       # one finishing kernel per layer (else this test compares a kernel with itself)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_swin_qkv_projection_inside_the_window_kernel(device):
    """Round 3: swin_qkv_attn_kernel - one workgroup per (window, head) projects its own q | k | v from the window's split LayerNorm rows
    (three-pass MFMA over fragment-ordered weights, k-steps of 32 in the 128-tile GEMM's order) into LDS and runs attn_f32_kernel's loop
    on them. Same products in the same order and the same softmax code (the compiler's contraction choices differ between the two
    kernels, so not the same bits): the full-size Swin-B detector (all four stages, shifted and unshifted windows, padded windows at
    532 x 620) must agree with the two-launch form (ovm_tune_set gdino_swin_fused = 0): boxes to 2e-5 (measured 6.6e-06), logits to 1e-4
    of their largest magnitude (measured 3.2e-04 absolute) after 24 Swin blocks, 6 encoder and 6 decoder layers."""
    from ovmono3d_amd import lib
    from ovmono3d_amd.util.synth_gdino_weights import synth_gdino_state_dict
    L = lib.load()
    g = torch.Generator().manual_seed(19)
    sd, hw = synth_gdino_state_dict(3), (532, 620)
    ids = [101, 2000 + 17, 1012, 2000 + 29, 2000 + 31, 1012, 2000 + 5, 1012, 102]
    img = torch.randint(0, 256, (3,) + hw, dtype=torch.uint8, generator=g).to(device)
    outs, launches = [], []
    try:
        for fused in (0, 1):
            assert L.ovm_tune_set(b"gdino_swin_fused", fused) == 0
            eng = _engine(device, sd, {}, use_graphs=False)
            logits, boxes = eng.forward(img, ids)
            outs.append((logits.clone(), boxes.clone()))
            launches.append(eng.launches())
            del eng
    finally:
        L.ovm_tune_set(b"gdino_swin_fused", 1)
    assert launches[1] == launches[0] - 24, launches           # 2 + 2 + 18 + 2 Swin-B blocks: one launch less each
    nid = len(ids)
    assert torch.isfinite(outs[1][0][:, :nid]).all() and torch.isfinite(outs[1][1]).all()
    d_box = float((outs[0][1] - outs[1][1]).abs().max()); d_log = float((outs[0][0][:, :nid] - outs[1][0][:, :nid]).abs().max())
    print(f"swin fused vs two launches: max |d boxes| {d_box:.3e}, max |d logits| {d_log:.3e}")
    assert_close(outs[1][1], outs[0][1], 2e-5, "pred_boxes (fused Swin window kernel vs two launches)")
    assert_close(outs[1][0][:, :nid], outs[0][0][:, :nid], 1e-4, "pred_logits (fused Swin window kernel vs two launches)")
