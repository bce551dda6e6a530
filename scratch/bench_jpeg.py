"""JPEG decode: Pillow on one host core vs the split decoder (host entropy decode + device reconstruction), per image size.
Prints one JSON line per case; `reconstruct_us` is the two device kernels by HIP events, `algorithmic_bytes` their compulsory HBM
traffic (coefficients in, planes out and in, RGB out)."""
import io, json, os, sys, time
import numpy as np, torch
from PIL import Image
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from ovmono3d_amd.data import gpu_jpeg
from ovmono3d_amd import lib as _lib
import ctypes as C
from test_jpeg import _scene, _jpeg

dev = torch.device("cuda:0")
L = _lib.load()
for name, hw, kw in [("coco_480x640_420", (480, 640), dict(quality=90, subsampling=2)), ("1080p_420", (1080, 1920), dict(quality=90, subsampling=2)),
                     ("1080p_444", (1080, 1920), dict(quality=90, subsampling=0))]:
    data = _jpeg(_scene(hw[0], hw[1], 7), **kw)
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        with Image.open(io.BytesIO(data)) as im:
            ref = np.asarray(im.convert("RGB"))
    t_pil = (time.perf_counter() - t0) / n
    for _ in range(3):
        coef, info = gpu_jpeg.entropy_decode(data, pin=True)      # first call: pinned allocation
    t0 = time.perf_counter()
    for _ in range(n):
        coef, info = gpu_jpeg.entropy_decode(data, pin=True)
    t_ent = (time.perf_counter() - t0) / n
    for _ in range(3):
        out = gpu_jpeg.decode_jpeg(data, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = gpu_jpeg.decode_jpeg(data, dev)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    assert np.array_equal(out.cpu().numpy(), ref)
    d_coef = coef.to(dev); planes = torch.empty(int(info.coef_blocks) * 64, dtype=torch.uint8, device=dev)
    rgb = torch.empty((hw[0], hw[1], 3), dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        L.ovm_jpeg_reconstruct(d_coef.data_ptr(), C.byref(info), planes.data_ptr(), rgb.data_ptr(), st)
    e0.record()
    for _ in range(50):
        L.ovm_jpeg_reconstruct(d_coef.data_ptr(), C.byref(info), planes.data_ptr(), rgb.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    nb = int(info.coef_blocks)
    alg = nb * 128 + nb * 64 * 2 + hw[0] * hw[1] * 3
    print(json.dumps({"case": name, "jpeg_bytes": len(data), "pillow_ms_one_core": round(t_pil * 1e3, 3), "host_entropy_decode_ms": round(t_ent * 1e3, 3),
                      "split_decode_end_to_end_ms": round(t_all * 1e3, 3), "reconstruct_us": round(us, 1), "algorithmic_bytes": alg,
                      "reconstruct_GBps": round(alg / us / 1e3, 1), "coef_upload_bytes": nb * 128, "bit_identical_to_pillow": True}))
