import sys, ctypes as C, torch, math
sys.path.insert(0,'/root/repo')
from ovmono3d_amd import lib
L=lib.load(); dev=torch.device('cuda')
def split(x):
    hi=torch.empty(x.shape,dtype=torch.float16,device=dev); lo=torch.empty_like(hi)
    L.ovm_op_split_f16(x.data_ptr(), x.numel(), hi.data_ptr(), lo.data_ptr(), None); return hi,lo
M=4097
for (N,K) in ((3072,1024),(1024,1024),(4096,1024),(1024,4096)):
    A=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev)/math.sqrt(K)
    ah,al=split(A); wh,wl=split(W); Cc=torch.empty(M,N,device=dev)
    wi=torch.empty(N,2*K,dtype=torch.float16,device=dev); L.ovm_op_interleave(wh.data_ptr(),wl.data_ptr(),N,K,wi.data_ptr(),None)
    ai=torch.empty(M,2*K,dtype=torch.float16,device=dev); L.ovm_op_interleave(ah.data_ptr(),al.data_ptr(),M,K,ai.data_ptr(),None)
    for prec in (1,3):
        for bm,st,ail in ((128,2,0),(128,6,0),(256,2,0),(256,3,0)):
            L.ovm_tune_set(b"gemm_bm", bm); L.ovm_tune_set(b"gemm_stages", st)
            if prec==1 and ail: continue
            args=(ah.data_ptr(),al.data_ptr(),K,wh.data_ptr(),wl.data_ptr()) if prec==1 else ((ai.data_ptr(),ai.data_ptr()+64,2*K) if ail else (ah.data_ptr(),al.data_ptr(),K))+(wi.data_ptr(),wi.data_ptr()+64)
            for _ in range(3): L.ovm_op_gemm(*args,M,N,K,None,0,Cc.data_ptr(),N,prec,None)
            torch.cuda.synchronize()
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): L.ovm_op_gemm(*args,M,N,K,None,0,Cc.data_ptr(),N,prec,None)
            e1.record(); torch.cuda.synchronize()
            ms=e0.elapsed_time(e1)/20
            fl=2.0*M*N*K
            if (N,K)==(3072,1024) or (N,K)==(1024,4096):
                ref=(A.double()@W.double().T).float()
                err=((Cc-ref).abs().max()/ref.abs().max()).item()
            else: err=float('nan')
            print(f"N={N} K={K} prec={prec} st={st} a_il={ail}: {ms*1e3:.1f} us  alg {fl/ms/1e9:.0f} TF  exec {fl*prec/ms/1e9:.0f} TF  err {err:.2e}")
    if (N,K)==(3072,1024):
        ref=(A.double()@W.double().T).float()
        print("check", ((Cc-ref).abs().max()/ref.abs().max()).item())
