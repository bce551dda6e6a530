"""Diagnostic: per-barrier s_memtime stamps of workgroup 0 of the 256 x 256 GEMM (fc1 shape), all CUs busy."""
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
M, N, K = 4096, int(os.environ.get("GN", 4096)), 1024
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K)
ai = il(*split(A)); wi = il(*split(W)); Cc = torch.empty(M, N, device=dev)
args = (ai.data_ptr(), ai.data_ptr() + 64, 2 * K, wi.data_ptr(), wi.data_ptr() + 64, M, N, K, None, 0, Cc.data_ptr(), N, 3, None)
st = torch.zeros(8 * 128, dtype=torch.int64, device=dev)
L.ovm_tune_set(b"op_gemm256", 1)
for _ in range(5): L.ovm_op_gemm(*args)
L.ovm_debug_set_ptr(b"gemm256_stamps", st.data_ptr())
for _ in range(3): L.ovm_op_gemm(*args)
torch.cuda.synchronize()
L.ovm_debug_set_ptr(b"gemm256_stamps", None)
s = st.cpu().view(8, 128)
for w in (0, 4):
    n = int(s[w][124]); t_entry, t_loop_end, t_exit = (int(s[w][i]) for i in (125, 126, 127))
    print(f"wave {w}: entry -> first barrier {int(s[w][0]) - t_entry}, last stamped barrier -> loop end {t_loop_end - int(s[w][n - 1])} "
          f"(stamps cover {n // 16} of 32 k-groups), loop end -> exit (epilogue) {t_exit - t_loop_end}, entry -> exit {t_exit - t_entry}")
    t = s[w][:n]; d = (t[1:] - t[:-1]).tolist()
    # stamps alternate: (before barrier, after barrier); d[2i] = barrier wait, d[2i+1] = segment body
    print(f"wave {w}: {len(t)} stamps; first 40 deltas (wait, body, wait, body ...):")
    print(" ".join(str(x) for x in d[:48]))
    wait = d[0::2]; body = d[1::2]
    # steady state: skip prologue (first 4 entries)
    import statistics
    print(f"  median barrier wait {statistics.median(wait[4:])}, median body {statistics.median(body[4:])}; per 8 bodies mean {sum(body[4:36])/32:.0f}, waits mean {sum(wait[4:36])/32:.0f}")
