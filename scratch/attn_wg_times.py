"""Per-workgroup durations of the lock-step attention kernel at the ViT-L shape (B = 1: one round of 256 workgroups on 256 CUs):
is the launch as long as its slowest workgroup, and do the slow ones sit on particular XCDs?"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ovmono3d_amd import lib
L = lib.load(); dev = torch.device('cuda')
for B in (1, 4):
    T, heads = 4097, 16
    qkv = (torch.randn(B * T, 3 * heads * 64) * 1.5).to(dev)
    out = torch.empty(B * T, heads * 64, device=dev)
    n = B * heads * 17
    st = torch.zeros(3 * (n + 64), dtype=torch.int64, device=dev)
    for _ in range(3): L.ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), 3, None)
    L.ovm_debug_set_ptr(b"attn_stamps", st.data_ptr())
    L.ovm_op_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), 3, None)
    torch.cuda.synchronize()
    L.ovm_debug_set_ptr(b"attn_stamps", None)
    s = st.cpu().view(-1, 3)
    s = s[s[:, 1] > 0]
    t0 = s[:, 0].min()
    start, end, xcc = (s[:, 0] - t0).double(), (s[:, 1] - t0).double(), s[:, 2] & 15
    dur = end - start
    print(f"B={B}: {len(s)} workgroups; launch span {end.max():.0f} ticks; workgroup duration mean {dur.mean():.0f} min {dur.min():.0f} max {dur.max():.0f}; "
          f"start spread {start.max():.0f}")
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print(f"   XCC {x}: {int(m.sum())} workgroups, mean {dur[m].mean():.0f}, max {dur[m].max():.0f}, last end {end[m].max():.0f}")
