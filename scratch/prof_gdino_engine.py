"""Times the C++ GroundingDINO engine alone (graph replay) at the bench's network resolution; run under rocprofv3 --kernel-trace --stats
for the per-kernel split."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ovmono3d_amd.gdino.detector import HashTokenizer
from ovmono3d_amd.gdino.engine import GdinoEngine
from ovmono3d_amd.util.synth_gdino import synth_gdino_model
dev = torch.device("cuda:0")
from ovmono3d_amd import lib as _lib
for kv in os.environ.get("OVM_TUNE", "").split(","):
    if "=" in kv:
        k, v = kv.split("="); assert _lib.load().ovm_tune_set(k.encode(), int(v)) == 0, kv
_, sd = synth_gdino_model(0)
H = int(os.environ.get("GD_H", 532)); W = int(os.environ.get("GD_W", 532))
eng = GdinoEngine(dev, sd, pixel_mean=[123.675, 116.28, 103.53], pixel_std=[58.395, 57.12, 57.375], use_graphs=os.environ.get("GD_GRAPH", "1") == "1")
img = torch.randint(0, 256, (3, H, W), dtype=torch.uint8).to(dev)
ids = HashTokenizer().encode("chair . dining table . sofa . potted plant . television . bookcase .")
for _ in range(3):
    eng.forward(img, ids)
torch.cuda.synchronize()
n = int(os.environ.get("GD_N", 20))
t0 = time.time()
for _ in range(n):
    eng.forward(img, ids)
torch.cuda.synchronize()
print(f"engine {H}x{W}: {(time.time() - t0) / n * 1e3:.2f} ms per forward, {eng.launches()} launches")
