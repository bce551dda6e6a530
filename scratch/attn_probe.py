"""One attention configuration per process (for rocprofv3 --kernel-trace --stats): python3 scratch/attn_probe.py B T [tune=val,...]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ovmono3d_amd import lib
L = lib.load(); dev = torch.device("cuda")
B, T, heads = int(sys.argv[1]), int(sys.argv[2]), 16
for kv in (sys.argv[3].split(",") if len(sys.argv) > 3 else []):
    k, v = kv.split("="); assert L.ovm_tune_set(k.encode(), int(v)) == 0
qkv = (torch.randn(B * T, 3 * heads * 64) * 1.5).to(dev)
out = torch.empty(B * T, heads * 64, device=dev)
for _ in range(12):
    L.ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), 3, None)
torch.cuda.synchronize()
