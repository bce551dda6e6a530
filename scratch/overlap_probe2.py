"""Which ViT kernel starves the detector's stream? The GroundingDINO engine (graph replay, side stream) timed while the main stream
loops ONE kernel type: attention (T = 4097, 16 heads), the 256 x 256 GEMM (fc1 shape), the 128 x 128 GEMM (fc2 shape)."""
import os, sys, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ovmono3d_amd import lib
from ovmono3d_amd.gdino.detector import HashTokenizer
from ovmono3d_amd.gdino.engine import GdinoEngine
from ovmono3d_amd.util.synth_gdino import synth_gdino_model
L = lib.load(); dev = torch.device("cuda:0")
_, sd = synth_gdino_model(0)
eng = GdinoEngine(dev, sd, pixel_mean=[123.675, 116.28, 103.53], pixel_std=[58.395, 57.12, 57.375], use_graphs=True)
img = torch.randint(0, 256, (3, 532, 532), dtype=torch.uint8).to(dev)
ids = HashTokenizer().encode("chair . dining table . sofa . potted plant . television . bookcase .")
side = torch.cuda.Stream(dev, priority=-1)
B, T, heads = 1, 4097, 16
qkv = (torch.randn(B * T, 3 * heads * 64) * 1.5).to(dev); aout = torch.empty(B * T, heads * 64, device=dev)
def split(x):
    hi = torch.empty(x.shape, dtype=torch.float16, device=dev); lo = torch.empty_like(hi)
    L.ovm_op_split_f16(x.data_ptr(), x.numel(), hi.data_ptr(), lo.data_ptr(), None); return hi, lo
def il(hi, lo):
    r, K = hi.shape
    out = torch.empty(r, 2 * K, dtype=torch.float16, device=dev)
    L.ovm_op_interleave(hi.data_ptr(), lo.data_ptr(), r, K, out.data_ptr(), None); return out
def gemm_args(N, K):
    A = torch.randn(T, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K)
    ai = il(*split(A)); wi = il(*split(W)); Cc = torch.empty(T, N, device=dev)
    return (ai, wi, Cc), (ai.data_ptr(), ai.data_ptr() + 64, 2 * K, wi.data_ptr(), wi.data_ptr() + 64, T, N, K, None, 0, Cc.data_ptr(), N, 3, None)
k1, a1 = gemm_args(4096, 1024); k2, a2 = gemm_args(1024, 4096)
def attn(): L.ovm_op_attention(qkv.data_ptr(), B, T, heads, aout.data_ptr(), 3, None)
def g256(): L.ovm_tune_set(b"op_gemm256", 1); L.ovm_op_gemm(*a1)
def g128(): L.ovm_tune_set(b"op_gemm256", 0); L.ovm_op_gemm(*a2)
def gd():
    with torch.cuda.stream(side): eng.forward(img, ids)
def ev(): return torch.cuda.Event(enable_timing=True)
for f in (attn, g256, g128, gd): f(); f()
torch.cuda.synchronize()
def run(f, nloop):
    s0, s1, m0, m1 = ev(), ev(), ev(), ev()
    torch.cuda.synchronize()
    m0.record()
    for _ in range(nloop): f()
    m1.record(); torch.cuda.synchronize()
    alone = m0.elapsed_time(m1)
    m0.record()
    for _ in range(4): f()                      # the main stream is already busy when the detector starts
    s0.record(side); gd(); s1.record(side)
    for _ in range(nloop): f()
    m1.record(); torch.cuda.synchronize()
    return alone / nloop * 1e3, s0.elapsed_time(s1), m0.elapsed_time(m1) / (nloop + 4) * 1e3
s0, s1 = ev(), ev(); s0.record(side); gd(); s1.record(side); torch.cuda.synchronize()
print(f"detector alone {s0.elapsed_time(s1):.2f} ms")
for name, f, n in (("attention", attn, 90), ("gemm256 fc1", g256, 250), ("gemm128 fc2", g128, 220)):
    a, g, c = run(f, n)
    print(f"{name:12s}: kernel alone {a:7.1f} us/launch; with the detector beside it {c:7.1f} us/launch, detector {g:6.2f} ms")
