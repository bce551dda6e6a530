"""us per call of the generic projection on the shapes the GroundingDINO branch issues (calls queued back to back)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from ovmono3d_amd import lib as _lib
from pyref_gdino.ops import Ops
dev = torch.device("cuda:0")
o = Ops(dev, int(os.environ.get("PREC", "3")))
L = _lib.load()
SH = [(900, 256, 256, 62), (1156, 512, 2048, 18), (1156, 2048, 512, 18), (1296, 1536, 512, 18), (1296, 512, 512, 18), (6015, 256, 256, 21),
      (6015, 1024, 256, 12), (6015, 256, 1024, 6), (6015, 256, 2048, 6), (6015, 2048, 256, 6), (16, 256, 256, 36), (16, 1024, 256, 18),
      (16, 768, 3072, 12), (16, 3072, 768, 12), (16, 2304, 768, 12), (16, 768, 768, 12), (5184, 768, 256, 2), (5184, 1024, 256, 2),
      (4489, 256, 1024, 2), (20736, 384, 128, 2), (20736, 512, 128, 2), (17689, 128, 512, 2), (900, 256, 2048, 6), (900, 2048, 256, 6)]
if os.environ.get("SHAPES"):
    SH = [tuple(int(v) for v in t.split("x")) + (1,) for t in os.environ["SHAPES"].split(",")]
cfgs = [c for c in os.environ.get("CFGS", "").split(";") if c] or [""]
rows = {}
for cfg in cfgs:
    for kv in (cfg.split(",") if cfg else []):
        k, v = kv.split("=")
        assert L.ovm_tune_set(k.encode(), int(v)) == 0, kv
    for (M, N, K, cnt) in SH:
        x = torch.randn(M, K, device=dev); W = o.pack(torch.randn(N, K) / K ** 0.5, torch.randn(N)); y = torch.empty(M, N, device=dev)
        for _ in range(3): o.linear(x, W, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 40
        e0.record()
        for _ in range(n): o.linear(x, W, out=y)
        e1.record(); e1.synchronize()
        rows.setdefault((M, N, K, cnt), []).append(e0.elapsed_time(e1) / n * 1e3)
print("cfgs:", cfgs)
tot = [0.0] * len(cfgs)
for (M, N, K, cnt), v in rows.items():
    for i, t in enumerate(v): tot[i] += t * cnt / 1e3
    print(f"{M:6d} {N:5d} {K:5d} x{cnt:3d}  " + "  ".join(f"{t:7.1f}" for t in v))
print("weighted ms/fwd:", "  ".join(f"{t:7.3f}" for t in tot))
