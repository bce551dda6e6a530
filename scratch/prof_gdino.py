"""Detector-only profile driver: the GroundingDINO engine (Swin-B / BERT-base / 900 queries, synthetic weights) on one 532 x 532 image,
N forwards, for `rocprofv3 --kernel-trace --stats -- python3 scratch/prof_gdino.py` (graphs on by default: kernels inside a replayed
graph are traced too)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ovmono3d_amd import lib  # noqa: E402
from ovmono3d_amd.gdino.engine import GdinoEngine  # noqa: E402
from ovmono3d_amd.gdino.config import GDinoConfig  # noqa: E402
from ovmono3d_amd.util.synth_gdino_weights import synth_gdino_state_dict  # noqa: E402

if os.environ.get("OVM_LIB"):                     # scratch only: a diagnostic build of the library (-DOVM_DIAG)
    lib.LIB_PATH = os.environ["OVM_LIB"]
dev = torch.device("cuda", 0)
for kv in os.environ.get("OVM_TUNE", "").split(","):
    if "=" in kv:
        k, v = kv.split("=")
        assert lib.load().ovm_tune_set(k.encode(), int(v)) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hw = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (532, 532)
eng = GdinoEngine(dev, synth_gdino_state_dict(0), GDinoConfig(), pixel_mean=[103.53, 116.28, 123.675], pixel_std=[57.375, 57.12, 58.395],
                  use_graphs=os.environ.get("OVM_GRAPHS", "1") == "1")
img = torch.randint(0, 256, (3,) + hw, dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).to(dev)
ids = [101] + [2000 + 7 * i for i in range(1, 7) for _ in (0, 1)][:9] + [102]
ids = [101, 2023, 1012, 2024, 2025, 1012, 2026, 1012, 2027, 2028, 1012, 2029, 1012, 2030, 1012, 102]
for _ in range(3):
    eng.forward(img, ids)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    eng.forward(img, ids)
torch.cuda.synchronize()
print(f"detector alone: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per forward, {eng.launches()} op launches")
