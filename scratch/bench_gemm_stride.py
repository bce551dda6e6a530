"""Does a power-of-two row stride (K = 1024, 4096) cost L2 channel conflicts? Compare us per k-step with K slightly off."""
import sys, torch, math
sys.path.insert(0,'/root/repo')
from ovmono3d_amd import lib
L=lib.load(); dev=torch.device('cuda')
def split(x):
    hi=torch.empty(x.shape,dtype=torch.float16,device=dev); lo=torch.empty_like(hi)
    L.ovm_op_split_f16(x.data_ptr(), x.numel(), hi.data_ptr(), lo.data_ptr(), None); return hi,lo
M=4097
for st in (2,6):
  L.ovm_tune_set(b"gemm_stages", st)
  for N in (1024, 3072):
    for K in (1024, 1088, 4096, 4160):
        A=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev)/math.sqrt(K)
        ah,al=split(A); wh,wl=split(W); Cc=torch.empty(M,N,device=dev)
        wi=torch.empty(N,2*K,dtype=torch.float16,device=dev); L.ovm_op_interleave(wh.data_ptr(),wl.data_ptr(),N,K,wi.data_ptr(),None)
        args=(ah.data_ptr(),al.data_ptr(),K,wi.data_ptr(),wi.data_ptr()+64)
        for _ in range(3): L.ovm_op_gemm(*args,M,N,K,None,0,Cc.data_ptr(),N,3,None)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): L.ovm_op_gemm(*args,M,N,K,None,0,Cc.data_ptr(),N,3,None)
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/20
        print(f"st={st} N={N} K={K}: {ms*1e3:.1f} us  {ms*1e3/(K/32):.3f} us per k-step  alg {2.0*M*N*K/ms/1e9:.0f} TF")
