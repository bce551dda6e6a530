"""JPEG decode split between the host (entropy decode) and the device (reconstruction): SURVEY.md 8f row 2.

The oracle is the reference's own decoder, live: Pillow = libjpeg-turbo, which is also what cv2.imread links (reference
demo/demo.py:52, cubercnn/data/dataset_mapper.py:38). CPU tests pin the host entropy decoder and the numpy restatement
(oracle/jpeg_ref.py) against Pillow bit for bit; the GPU test pins the device reconstruction the same way."""
import io
import os

import numpy as np
import pytest
import torch
from PIL import Image

from ovmono3d_amd.data import gpu_jpeg

HERE = os.path.dirname(os.path.abspath(__file__))
COCO = os.path.join(HERE, "golden", "coco_000000101762.jpg")          # data fixture: one of the reference's demo inputs


def _scene(h, w, seed):
    """A smooth scene with edges and noise, so every frequency band and the chroma filters get exercised."""
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 120 * np.sin(xx / 9.0 + seed) * np.cos(yy / 13.0), 127 + 120 * np.sin((xx + yy) / 17.0),
                    255.0 * ((xx // 16 + yy // 12) % 2)], axis=-1)
    img += g.normal(0, 12, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def _jpeg(arr, **kw):
    b = io.BytesIO()
    Image.fromarray(arr).save(b, format="JPEG", **kw)
    return b.getvalue()


def _pil(data):
    with Image.open(io.BytesIO(data)) as im:
        return np.asarray(im.convert("RGB"))


CASES = [
    ("444_q90", (61, 83), dict(quality=90, subsampling=0)),
    ("422_q75", (64, 97), dict(quality=75, subsampling=1)),
    ("420_q75", (75, 101), dict(quality=75, subsampling=2)),
    ("420_q30_optimised_tables", (200, 333), dict(quality=30, subsampling=2, optimize=True)),
    ("420_q100", (48, 64), dict(quality=100, subsampling=2)),
    ("420_tiny_width", (9, 3), dict(quality=85, subsampling=2)),          # downsampled width 2: plain replication
    ("422_tiny_width", (5, 4), dict(quality=85, subsampling=1)),
    ("420_one_pixel", (1, 1), dict(quality=85, subsampling=2)),
    ("420_restart_rows", (130, 170), dict(quality=80, subsampling=2, restart_marker_rows=1)),
    ("444_restart_blocks", (40, 90), dict(quality=80, subsampling=0, restart_marker_blocks=3)),
    ("420_q5_16bit_tables", (64, 64), dict(quality=1, subsampling=2)),
    # progressive (SOF2): DC / AC first scans and refinements, EOB runs, spectral bands - Pillow writes libjpeg's standard script
    ("prog_420_q75", (75, 101), dict(quality=75, subsampling=2, progressive=True)),
    ("prog_444_q90", (61, 83), dict(quality=90, subsampling=0, progressive=True)),
    ("prog_422_optimised", (64, 97), dict(quality=60, subsampling=1, progressive=True, optimize=True)),
    ("prog_420_restart_rows", (130, 170), dict(quality=80, subsampling=2, progressive=True, restart_marker_rows=1)),
    ("prog_420_q100", (48, 64), dict(quality=100, subsampling=2, progressive=True)),
    ("prog_420_q5_long_eob_runs", (200, 333), dict(quality=5, subsampling=2, progressive=True)),
    ("prog_tiny", (9, 3), dict(quality=85, subsampling=2, progressive=True)),
]


def _case_bytes(name, hw, kw):
    arr = _scene(hw[0], hw[1], len(name))
    return _jpeg(arr, **kw)


@pytest.mark.parametrize("name,hw,kw", CASES, ids=[c[0] for c in CASES])
def test_host_entropy_decode_and_restatement_equal_pillow(name, hw, kw):
    from oracle import jpeg_ref
    data = _case_bytes(name, hw, kw)
    coef, info = gpu_jpeg.entropy_decode(data)
    assert (info.height, info.width) == hw and info.ncomp == 3
    out = jpeg_ref.reconstruct(coef.numpy(), info)
    assert np.array_equal(out, _pil(data))


def test_grey_and_coco_example_equal_pillow():
    from oracle import jpeg_ref
    for prog in (False, True):
        grey = _jpeg(_scene(77, 53, 3)[:, :, 0], quality=80, progressive=prog)
        coef, info = gpu_jpeg.entropy_decode(grey)
        assert info.ncomp == 1 and info.colorspace == 0
        assert np.array_equal(jpeg_ref.reconstruct(coef.numpy(), info), _pil(grey))
    data = open(COCO, "rb").read()
    coef, info = gpu_jpeg.entropy_decode(data)
    ref = _pil(data)
    assert (info.height, info.width) == ref.shape[:2]
    assert np.array_equal(jpeg_ref.reconstruct(coef.numpy(), info), ref)


def test_scope_and_errors():
    arr = _scene(40, 40, 1)
    # a progressive file whose last scan is missing: libjpeg would smooth across blocks - refused, not approximated
    prog = _jpeg(arr, progressive=True)
    cut = prog[:prog.rindex(b"\xff\xda")] + b"\xff\xd9"
    assert gpu_jpeg.jpeg_info(cut).width == 40
    with pytest.raises(gpu_jpeg.UnsupportedJpeg):
        gpu_jpeg.entropy_decode(cut)
    cmyk = io.BytesIO()
    Image.fromarray(np.dstack([arr, arr[:, :, :1]]), mode="CMYK").save(cmyk, format="JPEG")
    with pytest.raises(gpu_jpeg.UnsupportedJpeg):
        gpu_jpeg.jpeg_info(cmyk.getvalue())
    good = _jpeg(arr, quality=80)
    from ovmono3d_amd.lib import OvmError
    with pytest.raises(OvmError):
        gpu_jpeg.entropy_decode(good[:len(good) // 2])                # truncated entropy-coded segment (every cut: test_truncated_files_*)
    with pytest.raises(OvmError):                                      # a cut inside a marker segment
        gpu_jpeg.jpeg_info(good[:30])
    with pytest.raises(OvmError):
        gpu_jpeg.jpeg_info(b"\x89PNG\r\n\x1a\n" + bytes(64))
    with pytest.raises(RuntimeError):
        gpu_jpeg.decode_jpeg(good, torch.device("cpu"))                # no CPU fallback of the reconstruction


@pytest.mark.parametrize("kind", ["baseline_420", "restart_markers", "progressive", "grey_optimised"])
def test_truncated_files_are_never_accepted(kind):
    """A file cut anywhere must not decode to a partly grey image: Pillow - the reference's reader (dataset_mapper.py:38; cv2 links the same
    libjpeg-turbo) - raises "image file is truncated" on such a file, so an evaluation over a damaged dataset fails there instead of scoring
    garbage. The host decoder tracks when it starts CONSUMING the zero bits that stand in for missing data (BitReader::exhausted) and returns
    OVM_ERR_INVALID. EVERY cut offset of four small files, with Pillow as the live judge: whatever Pillow refuses must be refused; where
    Pillow still accepts (a cut behind the last entropy-coded byte), an accepted decode must equal Pillow's bit for bit."""
    from oracle import jpeg_ref
    from ovmono3d_amd.lib import OvmError
    data = {"baseline_420": lambda: _jpeg(_scene(40, 56, 2), quality=80, subsampling=2),
            "restart_markers": lambda: _jpeg(_scene(33, 47, 3), quality=70, subsampling=0, restart_marker_blocks=2),
            "progressive": lambda: _jpeg(_scene(40, 56, 5), quality=70, subsampling=2, progressive=True),
            "grey_optimised": lambda: _jpeg(_scene(24, 31, 4)[:, :, 0], quality=70, optimize=True)}[kind]()
    assert np.array_equal(jpeg_ref.reconstruct(*(lambda c, i: (c.numpy(), i))(*gpu_jpeg.entropy_decode(data))), _pil(data))
    refused_by_pillow = accepted_wrongly = accepted_both = 0
    for cut in range(2, len(data)):
        part = data[:cut]
        try:
            ref = _pil(part)
        except Exception:
            ref = None
        try:
            coef, info = gpu_jpeg.entropy_decode(part)
            got = jpeg_ref.reconstruct(coef.numpy(), info)
        except (OvmError, gpu_jpeg.UnsupportedJpeg):
            got = None
        if ref is None:
            refused_by_pillow += 1
            accepted_wrongly += got is not None
        elif got is not None:
            accepted_both += 1
            assert np.array_equal(got, ref), f"cut at {cut}: accepted, but differs from Pillow"
    assert refused_by_pillow > len(data) // 2
    assert accepted_wrongly == 0, f"{accepted_wrongly} of {refused_by_pillow} truncations that Pillow refuses were decoded without an error"


def test_exif_orientation_goes_to_the_host_reader_and_is_applied(tmp_path):
    """Both readers of the reference apply the Exif Orientation tag (cv2.imread, demo/demo.py:52; detectron2 read_image ->
    _apply_exif_orientation, dataset_mapper.py:38): a rotated phone JPEG reaches the model upright, with height and width swapped. The
    device decoder leaves such files to the host reader (OVM_ERR_UNSUPPORTED -> UnsupportedJpeg), which transposes like Pillow's
    ImageOps.exif_transpose; orientation 1 / no Exif stays on the device path."""
    from PIL import ImageOps
    from ovmono3d_amd.data.feeding import read_image
    arr = _scene(40, 56, 7)
    for orient in (1, 3, 6, 8):
        exif = Image.Exif()
        exif[0x0112] = orient
        b = io.BytesIO()
        Image.fromarray(arr).save(b, format="JPEG", quality=85, exif=exif.tobytes())
        data = b.getvalue()
        if orient == 1:
            assert gpu_jpeg.jpeg_info(data).width == 56
        else:
            with pytest.raises(gpu_jpeg.UnsupportedJpeg):
                gpu_jpeg.jpeg_info(data)
        f = tmp_path / f"o{orient}.jpg"
        f.write_bytes(data)
        with Image.open(io.BytesIO(data)) as im:
            want = np.asarray(ImageOps.exif_transpose(im).convert("RGB"))
        got = read_image(str(f), "RGB")
        assert got.shape == ((56, 40, 3) if orient in (6, 8) else (40, 56, 3)) and np.array_equal(got, want)
        assert np.array_equal(read_image(str(f), "BGR"), want[:, :, ::-1])


def test_corrupt_streams_are_refused_or_decoded_never_worse():
    """The host half parses untrusted files (baseline and progressive): byte flips, truncations and spliced garbage must end in an error code or in a
    decode of the declared size - never in a crash or an out-of-range write (the coefficient buffer is guarded by canaries)."""
    import ctypes as C
    from ovmono3d_amd import lib as _lib
    L = _lib.load()
    rng = np.random.default_rng(0)
    seeds = [_jpeg(_scene(40, 56, 2), quality=80, subsampling=2), _jpeg(_scene(33, 20, 3), quality=60, subsampling=0, restart_marker_blocks=2),
             _jpeg(_scene(24, 24, 4)[:, :, 0], quality=70, optimize=True), open(COCO, "rb").read()[:6000],
             _jpeg(_scene(40, 56, 5), quality=70, subsampling=2, progressive=True),
             _jpeg(_scene(33, 47, 6), quality=90, subsampling=0, progressive=True, restart_marker_rows=1)]
    outcomes = {0: 0}
    for it in range(600):
        data = bytearray(seeds[it % len(seeds)])
        kind = it % 4
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                data[int(rng.integers(2, len(data)))] = int(rng.integers(0, 256))
        elif kind == 1:
            data = data[:int(rng.integers(2, len(data)))]
        elif kind == 2:
            i = int(rng.integers(2, len(data)))
            data[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
        else:
            i = int(rng.integers(2, len(data) - 8))
            del data[i:i + int(rng.integers(1, 8))]
        buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
        info = _lib.OvmJpegInfo()
        rc = L.ovm_host_jpeg_info(C.addressof(buf), len(data), C.byref(info))
        outcomes[rc] = outcomes.get(rc, 0) + 1
        if rc != 0:
            assert rc in (-1, -6), rc
            continue
        assert 0 < info.coef_blocks <= (1 << 22) and info.ncomp in (1, 3)
        nb = int(info.coef_blocks)
        coef = np.full((nb + 2) * 64, 0x5a5a, dtype=np.int16)
        rc = L.ovm_host_jpeg_entropy_decode(C.addressof(buf), len(data), coef[64:].ctypes.data, nb * 64, C.byref(info))
        assert rc in (0, -1, -6), rc
        assert (coef[:64] == 0x5a5a).all() and (coef[-64:] == 0x5a5a).all(), "wrote outside the coefficient planes"
    assert outcomes.get(-1, 0) > 20, outcomes                   # the mutations did reach the parser's error paths


@pytest.mark.gpu
@pytest.mark.parametrize("name,hw,kw", CASES, ids=[c[0] for c in CASES])
def test_device_decode_equals_pillow(device, name, hw, kw):
    data = _case_bytes(name, hw, kw)
    out = gpu_jpeg.decode_jpeg(data, device)
    assert out.dtype == torch.uint8 and tuple(out.shape) == (hw[0], hw[1], 3)
    assert np.array_equal(out.cpu().numpy(), _pil(data))


@pytest.mark.gpu
def test_device_decode_coco_example_grey_and_reader(device, tmp_path):
    data = open(COCO, "rb").read()
    assert np.array_equal(gpu_jpeg.decode_jpeg(data, device).cpu().numpy(), _pil(data))
    big = _jpeg(_scene(1080, 1920, 9), quality=85, subsampling=2)
    assert np.array_equal(gpu_jpeg.decode_jpeg(big, device).cpu().numpy(), _pil(big))
    grey = _jpeg(_scene(77, 53, 3)[:, :, 0], quality=80)
    assert np.array_equal(gpu_jpeg.decode_jpeg(grey, device).cpu().numpy(), _pil(grey))
    # the reader: device decode for baseline files (BGR = cv2.imread order), host reader + upload for the rest
    from ovmono3d_amd.data.feeding import read_image
    p1, p2, p3 = str(tmp_path / "a.jpg"), str(tmp_path / "b.jpg"), str(tmp_path / "c.png")
    open(p1, "wb").write(data)
    cm = io.BytesIO(); Image.fromarray(np.dstack([_scene(60, 80, 4), _scene(60, 80, 4)[:, :, :1]]), mode="CMYK").save(cm, format="JPEG")
    open(p2, "wb").write(cm.getvalue())                                    # outside the scope: host reader + upload
    Image.fromarray(_scene(31, 47, 5)).save(p3)
    for p in (p1, p2, p3):
        for fmt in ("RGB", "BGR"):
            assert np.array_equal(gpu_jpeg.read_image_device(p, fmt, device).cpu().numpy(), read_image(p, fmt))
