import sys, ctypes as C, torch
sys.path.insert(0,'/root/repo')
from ovmono3d_amd import lib
L=lib.load(); dev=torch.device('cuda')
M,N,K=128,128,64
A=torch.full((M,K), 2.0**-16, device=dev).half()      # fp16 subnormal
W=torch.ones(N,K, device=dev).half()
Cc=torch.zeros(M,N,device=dev)
rc=L.ovm_op_gemm(A.data_ptr(), None, K, W.data_ptr(), None, M,N,K, None,0, Cc.data_ptr(), N, 1, None)
torch.cuda.synchronize()
print("rc",rc,"A subnormal:", Cc[0,0].item(), "expected", 64*2.0**-16)
A=torch.ones(M,K,device=dev).half(); W=torch.full((N,K),2.0**-20,device=dev).half()
rc=L.ovm_op_gemm(A.data_ptr(), None, K, W.data_ptr(), None, M,N,K, None,0, Cc.data_ptr(), N, 1, None)
torch.cuda.synchronize()
print("W subnormal:", Cc[0,0].item(), "expected", 64*2.0**-20)
