"""Diagnostic: per-barrier s_memtime stamps of workgroup 0 of the two-wave-group attention kernel at the ViT-L shape."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ovmono3d_amd import lib
L = lib.load(); dev = torch.device('cuda')
B, T, heads = 1, 4097, 16
qkv = (torch.randn(B * T, 3 * heads * 64) * 1.5).to(dev)
out = torch.empty(B * T, heads * 64, device=dev)
for mode, name in ((1, "full"), (2, "no fragment reads"), (3, "no MFMAs")):
    st = torch.zeros(8 * 128, dtype=torch.int64, device=dev)
    L.ovm_tune_set(b"attn_pp", 1)
    for _ in range(3): L.ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), 3, None)
    L.ovm_tune_set(b"attn_pp", mode)
    L.ovm_debug_set_ptr(b"attn_stamps", st.data_ptr())
    for _ in range(2): L.ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), 3, None)
    torch.cuda.synchronize()
    L.ovm_debug_set_ptr(b"attn_stamps", None)
    L.ovm_tune_set(b"attn_pp", 1)
    s = st.cpu().view(8, 128)
    for w in (0, 4):
        n = int(s[w][127]); t = s[w][:n]; d = (t[1:] - t[:-1]).tolist()
        print(f"[{name}] wave {w} (wait, body, ...): " + " ".join(str(x) for x in d[40:64]))
        if mode == 2:
            f = s[w][109:119].tolist()
            print(f"    inside M(21) of wave {w}: entry -> after double step 0..7: " + " ".join(str(f[i + 1] - f[i]) for i in range(8)) + f"; segment start stamp -> M entry: n/a")
