"""ViT-L attention launch (T = 4097, 16 heads): lock-step kernel vs two-wave-group kernel (+ priority experiments), interleaved rounds."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ovmono3d_amd import lib
L = lib.load(); dev = torch.device('cuda')
B, T, heads = 1, 4097, 16
qkv = (torch.randn(B * T, 3 * heads * 64) * 1.5).to(dev)
out = torch.empty(B * T, heads * 64, device=dev)
variants = [("lock-step attn_kernel", 0, 0), ("two-group", 1, 0), ("two-group, S at prio 1", 1, 1), ("two-group, G1 static prio 1", 1, 2)]
res = {v[0]: [] for v in variants}
for rnd in range(5):
    for name, pp, pr in variants:
        L.ovm_tune_set(b"attn_pp", pp); L.ovm_tune_set(b"attn_prio", pr)
        for _ in range(2): L.ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), 3, None)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): L.ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), 3, None)
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 10)
L.ovm_tune_set(b"attn_pp", 0); L.ovm_tune_set(b"attn_prio", 0)
fl = 4.0 * T * T * heads * 64
for name, _, _ in variants:
    ms = sorted(res[name])[2]
    print(f"{name:32s}: median {ms*1e3:7.1f} us  min {min(res[name])*1e3:7.1f}  alg {fl/ms/1e9:5.0f} TF/s")
