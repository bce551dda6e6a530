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
M, N, K = 4096, 4096, 1024
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K)
ai = il(*split(A)); wi = il(*split(W)); Cc = torch.empty(M, N, device=dev)
args = (ai.data_ptr(), ai.data_ptr() + 64, 2 * K, wi.data_ptr(), wi.data_ptr() + 64, M, N, K, None, 0, Cc.data_ptr(), N, 3, None)
for v in (0, 1):
    L.ovm_tune_set(b"op_gemm256", v)
    for _ in range(30): L.ovm_op_gemm(*args)
    torch.cuda.synchronize()
