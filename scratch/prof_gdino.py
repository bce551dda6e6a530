import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ovmono3d_amd.gdino.detector import HashTokenizer
from ovmono3d_amd.gdino.model import GDinoConfig, GroundingDinoNative
from ovmono3d_amd.gdino.ops import Ops
from ovmono3d_amd.util.synth_gdino import synth_gdino_model
dev = torch.device("cuda:0")
_, sd = synth_gdino_model(0)
prec = int(os.environ.get("PREC", "3"))
net = GroundingDinoNative(Ops(dev, prec), sd, GDinoConfig())
H = W = 532
x = torch.randn(H * W, 3, device=dev)
ids = torch.tensor(HashTokenizer().encode("chair . dining table . sofa . potted plant . television . bookcase ."))
for _ in range(3):
    net.forward(x, H, W, ids)
torch.cuda.synchronize()
n = int(os.environ.get("N", "10"))
t0 = time.time()
for _ in range(n):
    net.forward(x, H, W, ids)
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
print(f"forward: host issue {(t1 - t0) / n * 1e3:.2f} ms, total {(t2 - t0) / n * 1e3:.2f} ms")

from ovmono3d_amd.gdino.detector import NativeGroundingDino
for use_graphs in (False, True):
    det = NativeGroundingDino(dev, sd, HashTokenizer(), [103.53, 116.28, 123.675], [57.375, 57.12, 58.395], precision=prec, use_graphs=use_graphs)
    im = torch.randint(0, 256, (3, H, W), dtype=torch.uint8, device=dev)
    cap = "chair . dining table . sofa . potted plant . television . bookcase ."
    for _ in range(4):
        det(im, cap)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        det(im, cap)
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    print(f"detector graphs={use_graphs}: host issue {(t1 - t0) / n * 1e3:.2f} ms, total {(t2 - t0) / n * 1e3:.2f} ms")
