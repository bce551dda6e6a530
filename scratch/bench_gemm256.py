"""ViT-L block contractions (M = 4097): the 256 x 256 two-wave-group kernel against the round-1 128 x 128 kernels, f16x3,
interleaved activations, interleaved rounds in one process (random operands)."""
import sys, os, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ovmono3d_amd import lib
L = lib.load(); dev = torch.device('cuda')
def split(x):
    hi = torch.empty(x.shape, dtype=torch.float16, device=dev); lo = torch.empty_like(hi)
    L.ovm_op_split_f16(x.data_ptr(), x.numel(), hi.data_ptr(), lo.data_ptr(), None); return hi, lo
def il(hi, lo):
    r, K = hi.shape
    out = torch.empty(r, 2 * K, dtype=torch.float16, device=dev)
    L.ovm_op_interleave(hi.data_ptr(), lo.data_ptr(), r, K, out.data_ptr(), None); return out
M = int(os.environ.get("GM", 4097))
for (N, K) in ((3072, 1024), (4096, 1024), (1024, 4096), (1024, 1024)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K)
    ai = il(*split(A)); wi = il(*split(W)); Cc = torch.empty(M, N, device=dev)
    args = (ai.data_ptr(), ai.data_ptr() + 64, 2 * K, wi.data_ptr(), wi.data_ptr() + 64, M, N, K, None, 0, Cc.data_ptr(), N, 3, None)
    variants = [("128x128 auto", 0), ("256x256", 1), ("256x256 split2", 2), ("256x256 split4", 4)]
    res = {n: [] for n, _ in variants}
    ref = (A.double() @ W.double().T).float()
    for rnd in range(5):
        for name, v in variants:
            L.ovm_tune_set(b"op_gemm256", v)
            for _ in range(2): L.ovm_op_gemm(*args)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): L.ovm_op_gemm(*args)
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 10)
            if rnd == 0:
                err = ((Cc - ref).abs().max() / ref.abs().max()).item()
                assert err < 1e-5, (name, err)
    L.ovm_tune_set(b"op_gemm256", 0)
    fl = 2.0 * M * N * K
    for name, _ in variants:
        ms = sorted(res[name])[len(res[name]) // 2]
        print(f"M={M} N={N} K={K} {name:16s}: median {ms*1e3:7.1f} us  min {min(res[name])*1e3:7.1f}  alg {fl/ms/1e9:6.0f} TF/s  executed {3*fl/ms/1e9:6.0f} TF/s")
