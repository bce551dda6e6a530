"""Two images in flight on one GPU: two model replicas (own handles, own detector engines, own streams), one Python thread each (ctypes
releases the GIL inside ovm_infer, which blocks on the step's one host sync). Aggregate images/s against one replica."""
import os, sys, threading, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from common import build_cfg
from ovmono3d_amd.gdino.detector import HashTokenizer, NativeGroundingDino
from ovmono3d_amd.modeling import build_model
from ovmono3d_amd.util.synth_gdino import synth_gdino_model
from ovmono3d_amd.util.synth_weights import synth_state_dict
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
R = int(os.environ.get("REPLICAS", 2)); N = int(os.environ.get("N", 40))
sd = synth_state_dict("vitl14", seed=0); _, gd_sd = synth_gdino_model(0)
CATS = ["chair", "dining table", "sofa", "potted plant", "television", "bookcase"]
models, inputs, streams = [], [], []
for r in range(R):
    cfg = build_cfg("vitl14", 896, "f16x3", max_batch=1, max_rois=1000, roi_heads="ROIHeads3DGDINO")
    m = build_model(cfg, device=dev); m.load_state_dict(sd)
    m.roi_heads.detector = NativeGroundingDino(dev, gd_sd, HashTokenizer(), cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, precision=3)
    models.append(m); streams.append(torch.cuda.Stream(dev))
    g = torch.Generator().manual_seed(1000 + r)
    inputs.append([{"image": torch.randint(0, 256, (3, 532, 532), dtype=torch.uint8, generator=g).to(dev), "height": 512, "width": 512,
                    "K": [[1024.0, 0, 256], [0, 1024.0, 256], [0, 0, 1]], "category_list": CATS}])
def work(r, n, out):
    with torch.cuda.stream(streams[r]):
        nd = 0
        for _ in range(n):
            nd += len(models[r](inputs[r])[0]["instances"])
        streams[r].synchronize()
    out[r] = nd
for r in range(R): work(r, 3, {})
torch.cuda.synchronize()
for use in (1, R, 1, R):
    out = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(r, N // use, out)) for r in range(use)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{use} in flight: {N / dt:.2f} images/s ({dt / N * 1e3:.2f} ms per image), detections {out}", flush=True)
