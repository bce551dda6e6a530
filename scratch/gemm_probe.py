"""One GEMM configuration per process (for rocprofv3 --kernel-trace --stats): python3 scratch/gemm_probe.py M N K [tune=val,...]  (split operands, interleaved)"""
import os, sys, math, ctypes as C, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ovmono3d_amd import lib
L = lib.load(); dev = torch.device("cuda")
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for kv in (sys.argv[4].split(",") if len(sys.argv) > 4 else []):
    k, v = kv.split("="); assert L.ovm_tune_set(k.encode(), int(v)) == 0
def split(x):
    hi = torch.empty_like(x, dtype=torch.float16); lo = torch.empty_like(x, dtype=torch.float16)
    assert L.ovm_op_split_f16(x.data_ptr(), x.numel(), hi.data_ptr(), lo.data_ptr(), None) == 0
    return hi, lo
def il(hi, lo):
    out = torch.empty(hi.shape[0], 2 * hi.shape[1], dtype=torch.float16, device=dev)
    assert L.ovm_op_interleave(hi.data_ptr(), lo.data_ptr(), hi.shape[0], hi.shape[1], out.data_ptr(), None) == 0
    return out
A = torch.randn(M, K, device=dev); W = torch.randn((N + 255) // 256 * 256, K, device=dev) / math.sqrt(K)
ai, wi = il(*split(A)), il(*split(W))
Cout = torch.empty(M, N, device=dev)
for _ in range(12):
    assert L.ovm_op_gemm(ai.data_ptr(), ai.data_ptr() + 64, 2 * K, wi.data_ptr(), wi.data_ptr() + 64, M, N, K, None, 0, Cout.data_ptr(), N, 3, None) == 0
torch.cuda.synchronize()
