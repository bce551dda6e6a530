"""RoI-head fc1 shape (M boxes x 12544 -> 1024): wave-specialised GEMM with and without split-K."""
import sys, torch, math
sys.path.insert(0,'/root/repo')
from ovmono3d_amd import lib
L=lib.load(); dev=torch.device('cuda')
def split(x):
    hi=torch.empty(x.shape,dtype=torch.float16,device=dev); lo=torch.empty_like(hi)
    L.ovm_op_split_f16(x.data_ptr(), x.numel(), hi.data_ptr(), lo.data_ptr(), None); return hi,lo
N,K=1024,12544
W=torch.randn(N,K,device=dev)/math.sqrt(K); wh,wl=split(W)
wi=torch.empty(N,2*K,dtype=torch.float16,device=dev); L.ovm_op_interleave(wh.data_ptr(),wl.data_ptr(),N,K,wi.data_ptr(),None)
for M in (32, 530, 1000):
    A=torch.randn(M,K,device=dev); ah,al=split(A); Cc=torch.empty(M,N,device=dev)
    for sk in (0,1):
        L.ovm_tune_set(b"gemm_splitk", sk)
        args=(ah.data_ptr(),al.data_ptr(),K,wi.data_ptr(),wi.data_ptr()+64)
        for _ in range(3): L.ovm_op_gemm(*args,M,N,K,None,0,Cc.data_ptr(),N,3,None)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): L.ovm_op_gemm(*args,M,N,K,None,0,Cc.data_ptr(),N,3,None)
        e1.record(); torch.cuda.synchronize()
        print(f"M={M} splitk={sk}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
