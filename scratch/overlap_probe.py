"""How the text-prompted detector (side stream, graph replay) and the DINOv2 backbone (main stream) share the chip: durations of
each alone, of both together, with events on their own streams. OVM_PRIO: side stream priority (-1 high, 0 normal)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from common import build_cfg
from ovmono3d_amd.gdino.detector import HashTokenizer
from ovmono3d_amd.gdino.engine import GdinoEngine
from ovmono3d_amd.modeling import build_model
from synth_gdino import synth_gdino_model
from ovmono3d_amd.util.synth_weights import synth_state_dict
dev = torch.device("cuda:0")
from ovmono3d_amd import lib as _lib
for kv in os.environ.get("OVM_TUNE", "").split(","):
    if "=" in kv:
        k, v = kv.split("="); assert _lib.load().ovm_tune_set(k.encode(), int(v)) == 0, kv
cfg = build_cfg("vitl14", 896, "f16x3", max_batch=1, max_rois=1000)
model = build_model(cfg, device=dev); model.load_state_dict(synth_state_dict("vitl14", seed=0))
_, sd = synth_gdino_model(0)
eng = GdinoEngine(dev, sd, pixel_mean=[123.675, 116.28, 103.53], pixel_std=[58.395, 57.12, 57.375], use_graphs=True)
img = torch.randint(0, 256, (3, 532, 532), dtype=torch.uint8).to(dev)
ids = HashTokenizer().encode("chair . dining table . sofa . potted plant . television . bookcase .")
inputs = [{"image": img, "height": 512, "width": 512, "K": [[1024.0, 0, 256], [0, 1024.0, 256], [0, 0, 1]]}]
images = model.preprocess_image(inputs)
side = torch.cuda.Stream(dev, priority=int(os.environ.get("OVM_PRIO", "-1")))
main = torch.cuda.current_stream(dev)
def ev(): return torch.cuda.Event(enable_timing=True)
def vit(): model.backbone(images)
def gd():
    with torch.cuda.stream(side): eng.forward(img, ids)
for _ in range(3): vit(); gd()
torch.cuda.synchronize()
N = 10
def timed(f_main, f_side):
    res = []
    for _ in range(N):
        m0, m1, s0, s1 = ev(), ev(), ev(), ev()
        torch.cuda.synchronize()
        if f_side: s0.record(side); f_side(); s1.record(side)
        if f_main: m0.record(main); f_main(); m1.record(main)
        torch.cuda.synchronize()
        r = {}
        if f_main: r["vit"] = m0.elapsed_time(m1)
        if f_side: r["gd"] = s0.elapsed_time(s1)
        if f_main and f_side: r["span"] = max(s0.elapsed_time(m1), s0.elapsed_time(s1))
        res.append(r)
    return {k: round(sum(r[k] for r in res) / N, 3) for k in res[0]}
print("vit alone", timed(vit, None))
print("gd alone ", timed(None, gd))
print("together ", timed(vit, gd))
print("together again", timed(vit, gd))
