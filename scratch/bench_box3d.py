"""Throughput of the 3D IoU kernel (pairs/s) next to the scipy float64 oracle on the host (bounded sample)."""
import sys, os, time, json, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from test_eval import _rand_boxes
from oracle import box3d as ob
from ovmono3d_amd.evaluation.omni3d_eval import box3d_overlap
dev = torch.device("cuda:0")
N = M = 2048
dt = torch.tensor(_rand_boxes(N, 1, spread=4.0), dtype=torch.float32, device=dev)
gt = torch.tensor(_rand_boxes(M, 2, spread=4.0), dtype=torch.float32, device=dev)
for _ in range(3): box3d_overlap(dt, gt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): iou = box3d_overlap(dt, gt)
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / 10
t0 = time.time(); n = 0
a, b = dt[:24].cpu().numpy().astype(np.float64), gt[:24].cpu().numpy().astype(np.float64)
ref = ob.iou_matrix(a, b); n = 24 * 24
cpu_s = time.time() - t0
print(json.dumps({"kernel": "box3d_iou_kernel", "pairs": N * M, "ms": round(ms, 3), "gpu_pairs_per_s": round(N * M / ms * 1e3), "overlapping_fraction": float((iou > 0).float().mean()),
                  "cpu_oracle_pairs_per_s": round(n / cpu_s, 1), "cpu_sample": f"{n} pairs, scipy HalfspaceIntersection+ConvexHull float64, 1 core, {cpu_s:.1f} s",
                  "max_abs_diff_vs_oracle_on_sample": float(np.abs(iou[:24, :24].cpu().numpy() - ref).max())}))
