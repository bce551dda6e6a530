import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ovmono3d_amd import lib
L = lib.load(); dev = torch.device('cuda')
B, T, heads = 1, 4097, 16
qkv = (torch.randn(B * T, 3 * heads * 64) * 1.5).to(dev)
out = torch.empty(B * T, heads * 64, device=dev)
for pp in (1, 0):
    L.ovm_tune_set(b"attn_pp", pp)
    for _ in range(12): L.ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), 3, None)
    torch.cuda.synchronize()
