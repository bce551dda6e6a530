"""Per-op, per-shape device time of one GroundingDINO forward (each call synchronised: no overlap, no launch gaps)."""
import sys, os, time, collections, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ovmono3d_amd.gdino.detector import HashTokenizer
from ovmono3d_amd.gdino.model import GDinoConfig, GroundingDinoNative
from ovmono3d_amd.gdino.ops import Ops
from ovmono3d_amd.util.synth_gdino import synth_gdino_model
dev = torch.device("cuda:0")
_, sd = synth_gdino_model(0)
ops = Ops(dev, 3)
net = GroundingDinoNative(ops, sd, GDinoConfig())
H = W = 532
x = torch.randn(H * W, 3, device=dev)
ids = torch.tensor(HashTokenizer().encode("chair . dining table . sofa . potted plant . television . bookcase ."))
for _ in range(2):
    net.forward(x, H, W, ids)
torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(name):
    f = getattr(ops, name)
    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = f(*a, **k); e1.record(); e1.synchronize()
        key = [name]
        for t in a:
            if isinstance(t, torch.Tensor): key.append(tuple(t.shape))
            elif hasattr(t, "N"): key.append(("W", t.N, t.K))
            elif isinstance(t, (int, bool)): key.append(t)
        acc[tuple(key)][0] += e0.elapsed_time(e1); acc[tuple(key)][1] += 1
        return r
    setattr(ops, name, g)
for n in ["linear", "layernorm", "bmm_raw", "softmax", "softmax2", "elt", "add", "gather_rows", "groupnorm", "msdeform", "sine_embed", "rowmax", "topk"]:
    if hasattr(ops, n): wrap(n)
net.forward(x, H, W, ids)
tot = sum(v[0] for v in acc.values())
print("total ms", tot, "calls", sum(v[1] for v in acc.values()))
byop = collections.defaultdict(float)
for k, v in acc.items(): byop[k[0]] += v[0]
print({k: round(v, 2) for k, v in byop.items()})
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{v[0]:7.3f} ms {v[1]:4d}x  {v[0]/v[1]*1e3:7.1f} us  {k}")
