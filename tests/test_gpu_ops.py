"""GPU parity of individual HIP kernels against fp32 torch/oracle references, through the C ABI."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from common import assert_close, rel_err

pytestmark = pytest.mark.gpu


def _lib():
    from ovmono3d_amd import lib
    return lib.load()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _split(x):
    L = _lib()
    hi = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    lo = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    rc = L.ovm_op_split_f16(x.data_ptr(), x.numel(), hi.data_ptr(), lo.data_ptr(), _stream())
    assert rc == 0
    return hi, lo


def _interleave(hi, lo):
    """[rows][K] hi, lo -> the interleaved operand image [rows][K/32][hi 32 | lo 32] of the split-precision GEMM."""
    rows, K = hi.shape
    out = torch.empty(rows, 2 * K, dtype=torch.float16, device=hi.device)
    assert _lib().ovm_op_interleave(hi.data_ptr(), lo.data_ptr(), rows, K, out.data_ptr(), _stream()) == 0
    return out


def _gemm_args(ah, al, wh, wl, K, precision, a_interleaved=False):
    """(a_hi, a_lo, lda, w_hi, w_lo, keepalive) for ovm_op_gemm: split weights are always an interleaved image."""
    if precision != 3:
        return ah.data_ptr(), al.data_ptr(), K, wh.data_ptr(), wl.data_ptr(), ()
    wi = _interleave(wh, wl)
    if a_interleaved:
        ai = _interleave(ah, al)
        return ai.data_ptr(), ai.data_ptr() + 64, 2 * K, wi.data_ptr(), wi.data_ptr() + 64, (wi, ai)
    return ah.data_ptr(), al.data_ptr(), K, wi.data_ptr(), wi.data_ptr() + 64, (wi,)


def test_split_f16_roundtrip(device):
    x = torch.randn(10007, device=device) * 3
    hi, lo = _split(x)
    rec = hi.float() + lo.float()
    assert rel_err(rec, x) < 1e-6
    assert torch.equal(hi, x.half())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (257, 300, 192), (1, 13, 1024), (4097, 384, 128), (77, 1024, 12544)])
@pytest.mark.parametrize("precision", [1, 3, 4])
def test_gemm_store(device, M, N, K, precision):
    a_il = precision == 4                      # 4 = split mode with interleaved activations as well
    precision = 3 if a_il else precision
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g).to(device)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(device)
    bias = torch.randn(N, generator=g).to(device)
    Npad = (N + 127) // 128 * 128
    Wp = torch.zeros(Npad, K, device=device)
    Wp[:N] = W
    ah, al = _split(A)
    wh, wl = _split(Wp)
    Cout = torch.full((M, N + 3), 7.0, device=device)
    a_hi, a_lo, lda, w_hi, w_lo, keep = _gemm_args(ah, al, wh, wl, K, precision, a_il)
    rc = _lib().ovm_op_gemm(a_hi, a_lo, lda, w_hi, w_lo, M, N, K, bias.data_ptr(), 1, Cout.data_ptr(), N + 3, precision, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    ref = torch.relu(A.double() @ W.double().T + bias.double()).float()
    # f16x3 leaves fp32 accumulation noise ~ sqrt(K) * 2^-24 (as any fp32 GEMM); f16 leaves operand rounding 2^-11
    tol = max(2e-6, 3e-8 * math.sqrt(K)) if precision == 3 else 3e-3
    assert_close(Cout[:, :N], ref, tol, f"gemm {M}x{N}x{K} p{precision}")
    assert torch.all(Cout[:, N:] == 7.0), "GEMM wrote outside its N range"


def test_gemm_exact_integers(device):
    """A = I-like / asymmetric small integers: catches transposed or permuted fragment maps exactly."""
    M, N, K = 130, 200, 128
    A = torch.zeros(M, K, device=device)
    A[torch.arange(M), torch.arange(M) % K] = 1.0
    A[:, 5] += 2.0
    W = (torch.arange(N * K, device=device).reshape(N, K) % 13 - 6).float()
    Wp = torch.zeros(256, K, device=device); Wp[:N] = W
    ah, al = _split(A); wh, wl = _split(Wp)
    for precision, a_il in ((1, False), (3, False), (3, True)):
        Cout = torch.zeros(M, N, device=device)
        a_hi, a_lo, lda, w_hi, w_lo, keep = _gemm_args(ah, al, wh, wl, K, precision, a_il)
        rc = _lib().ovm_op_gemm(a_hi, a_lo, lda, w_hi, w_lo, M, N, K, None, 0, Cout.data_ptr(), N, precision, _stream())
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(Cout, A @ W.T)


@pytest.mark.parametrize("M,D", [(5, 128), (4097, 1024), (300, 768), (1000, 256), (33, 384)])
def test_layernorm(device, M, D):
    x = torch.randn(M, D, device=device) * 2 + 0.5
    g = torch.rand(D, device=device) + 0.5
    b = torch.randn(D, device=device)
    y = torch.empty_like(x)
    rc = _lib().ovm_op_layernorm(x.data_ptr(), M, D, g.data_ptr(), b.data_ptr(), 1e-6, y.data_ptr(), _stream())
    assert rc == 0
    ref = torch.nn.functional.layer_norm(x.double(), (D,), g.double(), b.double(), 1e-6).float()
    assert_close(y, ref, 2e-6, "layernorm")


def _attn_ref(qkv, B, T, heads):
    D = heads * 64
    q, k, v = qkv.double().view(B, T, 3, heads, 64).permute(2, 0, 3, 1, 4)
    a = ((q * 0.125) @ k.transpose(-1, -2)).softmax(-1)
    return (a @ v).transpose(1, 2).reshape(B * T, D).float()


@pytest.mark.parametrize("B,T,heads", [(1, 64, 1), (1, 257, 2), (2, 130, 2), (1, 1370, 2), (1, 4097, 1), (1, 513, 2)])
@pytest.mark.parametrize("precision", [1, 3])
@pytest.mark.parametrize("waves", [4, 8])
def test_attention(device, B, T, heads, precision, waves):
    """waves = 8: the 256-query workgroup variant the engine uses by default; waves = 4: the co-run variant (one workgroup per CU)."""
    g = torch.Generator().manual_seed(T)
    qkv = (torch.randn(B * T, 3 * heads * 64, generator=g) * 1.5).to(device)
    out = torch.empty(B * T, heads * 64, device=device)
    try:
        assert _lib().ovm_tune_set(b"attn_waves", waves) == 0
        rc = _lib().ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), precision, _stream())
        torch.cuda.synchronize()
    finally:
        _lib().ovm_tune_set(b"attn_waves", 0)
    assert rc == 0
    ref = _attn_ref(qkv, B, T, heads)
    assert_close(out, ref, 5e-6 if precision == 3 else 5e-3, f"attention T={T} p{precision} w{waves}")


def test_attention_max_jump(device):
    """Forces the online-softmax rescale: one key per query block spikes late in the key sequence."""
    B, T, heads = 1, 300, 1
    qkv = torch.randn(B * T, 192, device=device) * 0.3
    qkv[:, 0:64] = 0.0
    qkv[:, 0] = 8.0                       # every query looks along dim 0
    qkv[290, 64] = 40.0                   # key 290 (5th tile) dominates -> running max jumps
    qkv[10, 64] = 12.0                    # an earlier, smaller spike in tile 0
    out = torch.empty(B * T, 64, device=device)
    for precision in (1, 3):
        rc = _lib().ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), precision, _stream())
        assert rc == 0
        assert_close(out, _attn_ref(qkv, B, T, heads), 5e-6 if precision == 3 else 5e-3, "attention max jump")


def test_roi_align(device):
    from oracle.roi_ops import roi_pooler
    g = torch.Generator().manual_seed(3)
    Cc, B = 64, 2
    feats = [torch.randn(B, Cc, s, s, generator=g) for s in (32, 16, 8)]
    scales = [1 / 7, 1 / 14, 1 / 28]
    n = 40
    x1 = torch.rand(n, generator=g) * 150 - 10
    y1 = torch.rand(n, generator=g) * 150 - 10
    w = torch.rand(n, generator=g) * 200 + 1
    h = torch.rand(n, generator=g) * 200 + 1
    boxes = torch.stack([x1, y1, x1 + w, y1 + h], 1)
    boxes[0] = torch.tensor([5.0, 5.0, 5.0, 9.0])           # zero width
    boxes[1] = torch.tensor([-50.0, -50.0, 400.0, 400.0])   # far outside
    boxes[2] = torch.tensor([10.0, 10.0, 10.5, 10.5])       # sub-pixel
    idx = torch.cat([torch.zeros(25, dtype=torch.int32), torch.ones(15, dtype=torch.int32)])
    ref = roi_pooler(feats, [boxes[:25], boxes[25:]], scales, 7, 2, 4)            # [n,C,7,7]
    ref = ref.permute(0, 2, 3, 1).reshape(n, -1)
    nhwc = [f.permute(0, 2, 3, 1).contiguous().to(device) for f in feats]
    hw = (C.c_int32 * 6)(32, 32, 16, 16, 8, 8)
    sc = (C.c_float * 3)(*scales)
    out = torch.empty(n, 49 * Cc, device=device)
    d_boxes, d_idx = boxes.to(device), idx.to(device)            # keep the device copies alive across the call
    rc = _lib().ovm_op_roi_align(nhwc[0].data_ptr(), nhwc[1].data_ptr(), nhwc[2].data_ptr(), hw, sc, Cc, 7, 2, 4,
                                 d_boxes.data_ptr(), d_idx.data_ptr(), n, out.data_ptr(), _stream())
    assert rc == 0
    assert_close(out, ref, 2e-5, "roi_align")


def test_cube_decode(device):
    from ovmono3d_amd.lib import OvmImage
    from oracle import heads as OH
    g = torch.Generator().manual_seed(5)
    n = 50
    head = torch.randn(n, 16, generator=g) * 0.5
    head[:, 11] = 1.0 + torch.rand(n, generator=g) * 3          # z
    head[:, 12] = torch.randn(n, generator=g)                   # uncertainty (clip at 0.01 exercised)
    head[3, 2:5] = 9.0                                          # dims clip at exp(5)
    x1 = torch.rand(n, generator=g) * 300
    y1 = torch.rand(n, generator=g) * 200
    boxes = torch.stack([x1, y1, x1 + 20 + torch.rand(n, generator=g) * 150, y1 + 20 + torch.rand(n, generator=g) * 150], 1)
    boxes[7] = torch.tensor([600.0, 100.0, 700.0, 150.0])       # outside the image after clipping -> dropped
    scores = torch.rand(n, generator=g)
    classes = torch.randint(0, 50, (n,), generator=g, dtype=torch.int32)
    idx = torch.cat([torch.zeros(30, dtype=torch.int32), torch.ones(20, dtype=torch.int32)])
    metas = [dict(h=266, w=355, oh=480, ow=640, K=[[900., 0, 320], [0, 900., 240], [0, 0, 1]]),
             dict(h=532, w=532, oh=512, ow=512, K=[[1024., 0, 256], [0, 1024., 256], [0, 0, 1]])]
    imgs = (OvmImage * 2)()
    for i, m in enumerate(metas):
        imgs[i].height, imgs[i].width, imgs[i].orig_height, imgs[i].orig_width = m["h"], m["w"], m["oh"], m["ow"]
        for j, v in enumerate(np.asarray(m["K"], dtype=np.float32).reshape(-1)):
            imgs[i].K[j] = float(v)
    rec = torch.zeros(n, 48, device=device)
    keep = torch.zeros(n, dtype=torch.int32, device=device)
    d = [t.to(device) for t in (head, boxes, scores, classes, idx)]   # keep the device copies alive across the call
    rc = _lib().ovm_op_cube_decode(d[0].data_ptr(), 16, d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                                   imgs, 2, n, 512.0, 1, rec.data_ptr(), keep.data_ptr(), _stream())
    assert rc == 0
    rec = rec.cpu(); keep = keep.cpu()
    # oracle: replay forward_cube's decode with the same head outputs
    nums = [30, 20]
    Ks = [torch.tensor(m["K"]) for m in metas]
    ratios = [m["oh"] / m["h"] for m in metas]
    Ks_box = torch.cat([(Ks[i] / ratios[i]).unsqueeze(0).repeat(k, 1, 1) for i, k in enumerate(nums)])
    Ks_box[:, -1, -1] = 1
    focal = torch.cat([Ks[i][1, 1].repeat(k) for i, k in enumerate(nums)])
    rat = torch.cat([torch.FloatTensor([ratios[i]]).repeat(k) for i, k in enumerate(nums)])
    ims = torch.cat([torch.FloatTensor([metas[i]["h"]]).repeat(k) for i, k in enumerate(nums)])
    v2r = OH.compute_virtual_scale_from_focal_spaces(focal, ims * rat, 512.0, ims)
    sw, sh = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    cx = boxes[:, 0] + 0.5 * sw + sw * head[:, 0]
    cy = boxes[:, 1] + 0.5 * sh + sh * head[:, 1]
    dims = torch.exp(head[:, 2:5].clip(max=5))
    pose = OH.R_from_allocentric(Ks_box, OH.rotation_6d_to_matrix(head[:, 5:11]), cx, cy)
    z = head[:, 11] * v2r
    x3 = z * (cx - Ks_box[:, 0, 2]) / Ks_box[:, 0, 0]
    y3 = z * (cy - Ks_box[:, 1, 2]) / Ks_box[:, 1, 1]
    conf = torch.exp(-head[:, 12].clip(0.01))
    verts = OH.get_cuboid_verts(torch.cat([torch.stack([x3, y3, z], 1), dims], 1), pose)
    tol = 2e-5
    assert_close(rec[:, 4], (scores * conf).sqrt(), tol, "score")
    assert_close(rec[:, 6:30].reshape(n, 8, 3), verts, tol, "bbox3D")
    assert_close(rec[:, 30:33], torch.stack([x3, y3, z], 1), tol, "center_cam")
    assert_close(rec[:, 33:35], torch.stack([cx, cy], 1) * rat[:, None], tol, "center_2D")
    assert_close(rec[:, 35:38], dims, tol, "dims")
    assert_close(rec[:, 38:47].reshape(n, 3, 3), pose, tol, "pose")
    assert torch.equal(rec[:, 5].contiguous().view(torch.int32), classes)
    assert torch.equal(rec[:, 47].contiguous().view(torch.int32), idx)
    assert keep[7] == 0 and keep.sum() == n - 1


@pytest.mark.parametrize("n", [1, 7, 300, 900, 2500])
def test_nms_op(device, n):
    from oracle.roi_ops import nms
    g = torch.Generator().manual_seed(n)
    xy = torch.rand(n, 2, generator=g) * 300
    wh = torch.rand(n, 2, generator=g) * 120 + 2
    boxes = torch.cat([xy, xy + wh], 1)
    scores = torch.rand(n, generator=g)
    ref = nms(boxes, scores, 0.5)
    db, ds = boxes.to(device), scores.to(device)
    keep = torch.full((n,), -1, dtype=torch.int32, device=device)
    nk = torch.zeros(1, dtype=torch.int32, device=device)
    rc = _lib().ovm_op_nms(db.data_ptr(), ds.data_ptr(), n, 0.5, keep.data_ptr(), nk.data_ptr(), _stream())
    assert rc == 0
    k = int(nk.item())
    assert keep[:k].cpu().tolist() == ref.tolist()


def test_gdino_glue_matches_oracle(device):
    """ROIHeads3DGDINO's output glue (reference roi_heads_gdino.py:186-202,253-254,273-294) on synthetic network outputs."""
    from oracle.gdino_glue import build_caption, gdino_postprocess as ref_post, phrase_spans
    from ovmono3d_amd.modeling.roi_heads.gdino_glue import gdino_postprocess
    cats = ["chair", "dining table", "potted plant"]
    caption, cap_list = build_caption(cats)
    assert caption == "chair . dining table . potted plant ."
    # fake WordPiece ids: [CLS] chair . dining table . potted plant . [SEP]
    phrase_ids = [[11], [21, 22], [31, 32, 33]]
    ids = [101, 11, 1012, 21, 22, 1012, 31, 32, 33, 1012, 102]
    spans = phrase_spans(ids, phrase_ids)
    assert spans == [(1, 2), (3, 5), (6, 9)]
    g = torch.Generator().manual_seed(0)
    nq = 900
    logits = torch.randn(nq, 256, generator=g) * 2 - 6          # most queries far below the 0.001 threshold
    logits[::7, 1:9] += 6
    logits[:, len(ids):] = float("-inf")                         # padded text positions (sigmoid -> 0)
    boxes = torch.rand(nq, 4, generator=g) * torch.tensor([1.0, 1.0, 0.4, 0.4])
    rb, rs, rc = ref_post(logits, boxes, spans, cap_list, [[c] for c in cats], (532, 709))
    b, s, c = gdino_postprocess(logits.to(device), boxes.to(device), spans, (532, 709))
    assert len(rs) > 20 and len(rs) == len(s)
    assert torch.equal(c.cpu(), rc)
    assert_close(b, rb, 1e-5, "gdino boxes")
    assert_close(s, rs, 1e-5, "gdino scores")


@pytest.mark.parametrize("nq,ncat,crowd", [(900, 6, 0.08), (1500, 140, 0.3), (2048, 3, 0.02), (64, 2, 0.5), (3000, 5, 0.1)])
def test_gdino_glue_shapes_and_paths(device, nq, ncat, crowd):
    """The glue's three-launch form (one-workgroup sort, bit-matrix, blocked greedy pass) against the oracle on crowded scenes:
    more than 1024 queries (2048-key sort), more phrases than travel as kernel arguments (device span table), a sort that is
    exactly full, fewer queries than one block, and nq > 2048 (the multi-kernel route). Then: no query above the threshold."""
    from oracle.gdino_glue import gdino_postprocess as ref_post
    from ovmono3d_amd.modeling.roi_heads.gdino_glue import gdino_postprocess
    g = torch.Generator().manual_seed(nq + ncat)
    ld = 2 * ncat + 2
    spans = [(1 + 2 * i, 2 + 2 * i) for i in range(ncat)]          # single-token phrases separated by '.'
    cats = [f"c{i}" for i in range(ncat)]
    logits = torch.randn(nq, ld, generator=g) * 3 - 3
    logits[::5] -= 12                                              # a fifth of the queries fall under the 0.001 threshold
    centres = torch.rand(nq, 2, generator=g)
    boxes = torch.cat([centres, crowd * (0.5 + torch.rand(nq, 2, generator=g))], dim=1)
    rb, rs, rc = ref_post(logits, boxes, spans, cats, [[c] for c in cats], (480, 640))
    b, s, c = gdino_postprocess(logits.to(device), boxes.to(device), spans, (480, 640))
    assert 4 < len(rs) < int(0.8 * nq) and len(rs) == len(s), (len(rs), len(s))
    assert torch.equal(c.cpu(), rc)
    assert_close(b, rb, 1e-6, "gdino boxes")
    assert_close(s, rs, 1e-5, "gdino scores")
    b, s, c = gdino_postprocess((logits - 40).to(device), boxes.to(device), spans, (480, 640))
    assert len(s) == 0 and len(b) == 0


@pytest.mark.parametrize("M,N,K,ksplit", [(256, 256, 64, 1), (300, 512, 96, 1), (4097, 768, 256, 1), (1024, 256, 1024, 4), (700, 512, 2048, 8),
                                          (512, 256, 32, 1), (4097, 512, 2048, 4)])       # last: split-K together with a leftover row
def test_gemm256_two_wave_group_kernel(device, M, N, K, ksplit):
    """The 256 x 256 kernel (gemm256.hip: two wave groups ping-ponging LOAD / COMPUTE segments, re-staged half-tiles, counted
    vmcnt across raw barriers) through ovm_op_gemm: exact on small integers with an asymmetric W (fragment / quadrant maps),
    fp32-class on random data, ragged M (clamped rows, the dot-product tail workgroups at M % 256 <= 8), odd and even numbers
    of k-groups, split-K. Repeated launches must agree bit for bit (the hazards are placed by count, not by luck)."""
    import ctypes as C
    L = _lib()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    try:
        assert L.ovm_tune_set(b"op_gemm256", ksplit) == 0
        # exact integers
        A = torch.zeros(M, K)
        A[torch.arange(M), torch.arange(M) % K] = 1.0
        A[:, 5 % K] += 2.0
        A[:, (K - 1)] -= 3.0
        W = (torch.arange(N * K).reshape(N, K) % 13 - 6).float()
        A, W = A.to(device), W.to(device)
        ah, al = _split(A); wh, wl = _split(W)
        a_hi, a_lo, lda, w_hi, w_lo, keep = _gemm_args(ah, al, wh, wl, K, 3, True)
        Cout = torch.full((M, N), 7.0, device=device)
        assert L.ovm_op_gemm(a_hi, a_lo, lda, w_hi, w_lo, M, N, K, None, 0, Cout.data_ptr(), N, 3, _stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(Cout, A @ W.T)
        # random data
        A = torch.randn(M, K, generator=g).to(device)
        W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(device)
        bias = torch.randn(N, generator=g).to(device)
        ah, al = _split(A); wh, wl = _split(W)
        a_hi, a_lo, lda, w_hi, w_lo, keep = _gemm_args(ah, al, wh, wl, K, 3, True)
        outs = []
        for rep in range(3):
            Cout = torch.full((M, N), 7.0, device=device)
            assert L.ovm_op_gemm(a_hi, a_lo, lda, w_hi, w_lo, M, N, K, bias.data_ptr(), 1, Cout.data_ptr(), N, 3, _stream()) == 0
            torch.cuda.synchronize()
            outs.append(Cout)
        ref = torch.relu(A.double() @ W.double().T + bias.double()).float()
        assert_close(outs[0], ref, max(2e-6, 3e-8 * math.sqrt(K)), f"gemm256 {M}x{N}x{K} split {ksplit}")
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    finally:
        L.ovm_tune_set(b"op_gemm256", 0)


@pytest.mark.parametrize("B,T,heads", [(1, 4097, 2), (2, 1370, 3), (1, 257, 2), (1, 5477, 1), (1, 300, 2)])
def test_attention_two_wave_group_kernel_is_bit_identical(device, B, T, heads):
    """attn_pp_kernel (matrix and softmax segments of the two wave groups in alternating barrier intervals, 4-slot K / V^T rings
    re-staged under counted vmcnt waits) keeps attn_kernel's per-lane arithmetic and accumulation order: same bits, on ragged
    last tiles (T % 64 = 1, 26, 37, 44), short sequences (fewer tiles than ring slots) and repeated launches."""
    g = torch.Generator().manual_seed(T + heads)
    qkv = (torch.randn(B * T, 3 * heads * 64, generator=g) * 1.5).to(device)
    L = _lib()
    outs = {}
    try:
        assert L.ovm_tune_set(b"attn_waves", 8) == 0 and L.ovm_tune_set(b"attn_q64", 0) == 0
        for pp in (0, 1, 1):
            assert L.ovm_tune_set(b"attn_pp", pp) == 0
            out = torch.full((B * T, heads * 64), float("nan"), device=device)
            assert L.ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), 3, _stream()) == 0
            torch.cuda.synchronize()
            outs.setdefault(pp, []).append(out)
    finally:
        L.ovm_tune_set(b"attn_waves", 0)
        L.ovm_tune_set(b"attn_pp", 0)
    assert_close(outs[1][0], _attn_ref(qkv.cpu(), B, T, heads), 3e-6, "two-wave-group attention")
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[1][0], outs[1][1])


@pytest.mark.parametrize("B,T,heads", [(1, 4097, 16), (2, 1370, 3), (1, 257, 2), (1, 5477, 2), (1, 300, 2), (1, 64, 1), (3, 513, 1)])
def test_attention_64_queries_per_wave_kernel_is_bit_identical(device, B, T, heads):
    """attn64_kernel (round 3: 4 waves x 64 queries, every K / V^T fragment read feeds both query sub-tiles; ovm_tune_set attn_q64 = 1 -
    measured slower than the default, kept as the record of that experiment) keeps attn_kernel<3, 8>'s per-query arithmetic and accumulation order: same bits as the 8-wave x 32-query kernel
    (ovm_tune_set attn_q64 = 0), on ragged last key tiles, partial query blocks (T = 5477: 101 queries in the last workgroup),
    leftover-query workgroups (T % 256 <= 8), sequences shorter than the ring, and repeated launches; and fp32-class against fp64."""
    import time
    g = torch.Generator().manual_seed(T + heads)
    qkv = (torch.randn(B * T, 3 * heads * 64, generator=g) * 1.5).to(device)
    L = _lib()
    outs = {}
    try:
        for q64 in (0, 1, 1):
            assert L.ovm_tune_set(b"attn_q64", q64) == 0
            out = torch.full((B * T, heads * 64), float("nan"), device=device)
            assert L.ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), 3, _stream()) == 0
            torch.cuda.synchronize()
            outs.setdefault(q64, []).append(out)
        if T == 4097:                                              # ViT-L's shape: time both kernels (the op includes the qkv head split)
            for q64 in (0, 1):
                L.ovm_tune_set(b"attn_q64", q64)
                for _ in range(3):
                    L.ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), 3, _stream())
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    L.ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), 3, _stream())
                torch.cuda.synchronize()
                print(f"ovm_op_attention T=4097 heads=16 attn_q64={q64}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per call (incl. head split)")
    finally:
        L.ovm_tune_set(b"attn_q64", 0)
    assert_close(outs[1][0], _attn_ref(qkv.cpu(), B, T, heads), 5e-6, "64-query attention")
    assert torch.equal(outs[1][0], outs[1][1])
    # leftover queries (T % 256 <= 8) run in extra dot-product workgroups whose wave-level reduction order depends on the workgroup
    # size (8 waves there, 4 here): those rows agree to fp32 rounding, every tiled row bit for bit
    tail = T % 256 if (T > 256 and 0 < T % 256 <= 8) else 0
    a, b = outs[0][0].view(B, T, -1), outs[1][0].view(B, T, -1)
    assert torch.equal(a[:, :T - tail], b[:, :T - tail])
    if tail:
        assert_close(a[:, T - tail:], b[:, T - tail:], 2e-6, "leftover-query rows")


@pytest.mark.parametrize("B,T,heads", [(1, 4097, 16), (2, 261, 3), (1, 513, 2), (3, 1030, 1)])
def test_attention_leftover_queries_split_over_keys(device, B, T, heads):
    """The leftover queries of a launch (T = 4097: one per head; up to 8) are computed by 16 workgroups per query over runs of key tiles
    + a combine kernel (attn_tail.hpp; round 3: the one-workgroup form cost a quarter of the ViT-L launch), against the one-workgroup
    form (ovm_tune_set attn_tail_split = 0): tiled rows identical, leftover rows equal to fp32 rounding (the partial softmax states are
    merged in a fixed order: repeated launches agree bit for bit), both fp32-class against fp64. Cases: 1, 5, 1 and 6 leftover queries."""
    g = torch.Generator().manual_seed(T * 3 + heads)
    qkv = (torch.randn(B * T, 3 * heads * 64, generator=g) * 1.5).to(device)
    L = _lib()
    outs = {}
    try:
        for split in (0, 1, 1):
            assert L.ovm_tune_set(b"attn_tail_split", split) == 0
            out = torch.full((B * T, heads * 64), float("nan"), device=device)
            assert L.ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), 3, _stream()) == 0
            torch.cuda.synchronize()
            outs.setdefault(split, []).append(out)
    finally:
        L.ovm_tune_set(b"attn_tail_split", 1)
    tail = T % 256
    assert 0 < tail <= 8
    ref = _attn_ref(qkv.cpu(), B, T, heads)
    assert_close(outs[1][0], ref, 5e-6, "split leftover queries vs fp64")
    assert torch.equal(outs[1][0], outs[1][1])
    a, b = outs[0][0].view(B, T, -1), outs[1][0].view(B, T, -1)
    assert torch.equal(a[:, :T - tail], b[:, :T - tail])
    assert_close(b[:, T - tail:], a[:, T - tail:], 2e-6, "leftover rows: split vs one workgroup")
    assert_close(b[:, T - tail:], ref.view(B, T, -1)[:, T - tail:].to(device), 5e-6, "leftover rows vs fp64")


@pytest.mark.parametrize("kernel", ["ws128", "gemm256"])
def test_gemm_32bit_offset_boundary(device, kernel):
    """The GEMM kernels address their operands with 32-bit element offsets (gemm.hpp a_row_offset, gemm256.hip aoff). The
    largest launch that fits - M x lda = 2^32 halves exactly: 524,288 rows of an interleaved K = 4096 image, the fc2 input
    layout - must still read its LAST rows correctly, and one row more must be refused with OVM_ERR_CAPACITY instead of wrapping
    around (ovm_create applies the same bound to max_batch, test_create_refuses_offset_overflow)."""
    L = _lib()
    K, N = 4096, 256
    M = (1 << 32) // (2 * K)
    g = torch.Generator(device="cpu").manual_seed(5)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(device)
    wh, wl = _split(W)
    wi = _interleave(wh, wl)
    img = torch.zeros(M + 1, 2 * K, dtype=torch.float16, device=device)        # 8.6 GB; rows outside the probes stay zero
    probes = {0: 300, M // 2 - 7: 130, M - 256: 256}                            # first rows, a middle block, the last tile
    blocks = {}
    for r0, n in probes.items():
        A = torch.randn(n, K, generator=g).to(device)
        ah, al = _split(A)
        img[r0:r0 + n] = _interleave(ah, al)
        blocks[r0] = A
    Cout = torch.empty(M, N, device=device)
    try:
        if kernel == "gemm256":
            assert L.ovm_tune_set(b"op_gemm256", 1) == 0
        rc = L.ovm_op_gemm(img.data_ptr(), img.data_ptr() + 64, 2 * K, wi.data_ptr(), wi.data_ptr() + 64, M, N, K, None, 0,
                           Cout.data_ptr(), N, 3, _stream())
        torch.cuda.synchronize()
        assert rc == 0
        for r0, A in blocks.items():
            ref = (A.double() @ W.double().T).float()
            assert_close(Cout[r0:r0 + A.shape[0]], ref, 3e-6, f"rows {r0}..")
        assert float(Cout[400:M // 2 - 7].abs().max()) == 0.0 and float(Cout[M // 2 + 200:M - 256].abs().max()) == 0.0
        rc = L.ovm_op_gemm(img.data_ptr(), img.data_ptr() + 64, 2 * K, wi.data_ptr(), wi.data_ptr() + 64, M + 1, N, K, None, 0,
                           Cout.data_ptr(), N, 3, _stream())
        assert rc == -5, "a launch whose last row lies beyond 2^32 elements must be refused (OVM_ERR_CAPACITY)"
    finally:
        L.ovm_tune_set(b"op_gemm256", 0)


def test_create_refuses_offset_overflow(device):
    """ovm_create checks max_batch x T x row length against the kernels' 32-bit offsets before it allocates anything."""
    from common import build_cfg
    from ovmono3d_amd.lib import OvmError
    from ovmono3d_amd.native import Engine
    from ovmono3d_amd.util.synth_weights import synth_state_dict
    sd = synth_state_dict("vittest14", seed=1)
    for mb, ok in ((64, True), (1 << 16, False)):
        cfg = build_cfg("vittest14", 224, "f16x3", max_batch=mb, max_rois=4)
        eng = Engine(cfg, device)
        if ok:
            eng.load_state_dict(sd)
            eng.close()
        else:
            with pytest.raises(OvmError, match="32-bit"):
                eng.load_state_dict(sd)
