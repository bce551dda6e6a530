import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ovmono3d_amd.gdino.detector import HashTokenizer
from ovmono3d_amd.gdino.model import GDinoConfig, GroundingDinoNative
from ovmono3d_amd.gdino.ops import Ops
from ovmono3d_amd.util.synth_gdino import synth_gdino_model
dev = torch.device("cuda:0")
_, sd = synth_gdino_model(0)
net = GroundingDinoNative(Ops(dev, 3), sd, GDinoConfig())
H = W = 532
x = torch.randn(H * W, 3, device=dev)
ids = torch.tensor(HashTokenizer().encode("chair . dining table . sofa . potted plant . television . bookcase ."))
net.forward(x, H, W, ids); net.forward(x, H, W, ids)
cap = list(net._caption_cache.values())[0]
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
print("bert   %.2f ms" % timeit(lambda: net.bert.forward(cap["ids"], cap["mask"], cap["pos"], bias=cap["bias"])))
print("swin   %.2f ms" % timeit(lambda: net.swin.forward(x, H, W)))
print("total  %.2f ms" % timeit(lambda: net.forward(x, H, W, ids)))
