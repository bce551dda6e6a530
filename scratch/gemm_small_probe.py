"""The detector's GEMM shapes one by one (planar split A, interleaved split W, fp32 C): us per launch, hot operands.
python3 scratch/gemm_small_probe.py [tune=val,...]"""
import os, sys, math, ctypes as C, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ovmono3d_amd import lib
L = lib.load(); dev = torch.device("cuda")
for kv in (sys.argv[1].split(",") if len(sys.argv) > 1 else []):
    k, v = kv.split("="); assert L.ovm_tune_set(k.encode(), int(v)) == 0
def split(x):
    hi = torch.empty_like(x, dtype=torch.float16); lo = torch.empty_like(x, dtype=torch.float16)
    assert L.ovm_op_split_f16(x.data_ptr(), x.numel(), hi.data_ptr(), lo.data_ptr(), None) == 0
    return hi, lo
def il(hi, lo):
    out = torch.empty(hi.shape[0], 2 * hi.shape[1], dtype=torch.float16, device=dev)
    assert L.ovm_op_interleave(hi.data_ptr(), lo.data_ptr(), hi.shape[0], hi.shape[1], out.data_ptr(), None) == 0
    return out
shapes = [("enc fc1 / q|v", 6015, 2048, 256), ("enc fc2", 6015, 256, 2048), ("enc 256x256", 6015, 256, 256), ("s1 qkv", 20736, 384, 128),
          ("s1 fc1", 17689, 512, 128), ("s1 fc2", 17689, 128, 512), ("s2 qkv", 5184, 768, 256), ("s2 fc1", 4489, 1024, 256),
          ("s3 qkv", 1296, 1536, 512), ("s3 proj", 1296, 512, 512), ("s3 fc1", 1156, 2048, 512), ("s3 fc2", 1156, 512, 2048),
          ("s4 qkv", 576, 3072, 1024), ("s4 fc1", 289, 4096, 1024)]
if os.environ.get("IL"):
    shapes = [s for s in shapes if s[2] % 256 == 0]
if os.environ.get("SWEEP"):
    shapes = [(f"1 round K={k}", 2048, 2048, k) for k in (64, 128, 256, 512, 1024, 2048)] + [(f"3 rounds K={k}", 6015, 2048, k) for k in (64, 128, 256, 512, 1024)] + \
             [(f"half round K={k}", 1024, 2048, k) for k in (128, 512, 2048)]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device=dev); W = torch.randn((N + 255) // 256 * 256, K, device=dev) / math.sqrt(K)
    ah, al = split(A); wi = il(*split(W))
    Cout = torch.empty(M, N, device=dev)
    if os.environ.get("IL"):                                  # interleaved activations too (what the 256 x 256 kernel streams)
        ai = il(ah, al)
        def go():
            assert L.ovm_op_gemm(ai.data_ptr(), ai.data_ptr() + 64, 2 * K, wi.data_ptr(), wi.data_ptr() + 64, M, N, K, None, 0, Cout.data_ptr(), N, 3, None) == 0
    else:
        def go():
            assert L.ovm_op_gemm(ah.data_ptr(), al.data_ptr(), K, wi.data_ptr(), wi.data_ptr() + 64, M, N, K, None, 0, Cout.data_ptr(), N, 3, None) == 0
    for _ in range(5): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    ref = (A.double() @ W[:N].double().t()).float()
    err = float((Cout - ref).abs().max() / ref.abs().max())
    print(f"{name:14s} M {M:6d} N {N:5d} K {K:5d}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  out {M * N * 4 / us / 1e6:6.2f} TB/s  tiles {((M + 127) // 128) * ((N + 127) // 128):5d}  err {err:.1e}", flush=True)
