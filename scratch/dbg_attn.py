import sys, ctypes as C, torch
sys.path.insert(0,'/root/repo')
from ovmono3d_amd import lib
L=lib.load(); dev=torch.device('cuda')
def ref(qkv,B,T,heads):
    D=heads*64
    q,k,v=qkv.double().view(B,T,3,heads,64).permute(2,0,3,1,4)
    a=((q*0.125)@k.transpose(-1,-2)).softmax(-1)
    return (a@v).transpose(1,2).reshape(B*T,D).float()
for (B,T,heads) in ((1,257,2),(1,1370,2),(2,130,2)):
    g=torch.Generator().manual_seed(T)
    qkv=(torch.randn(B*T,3*heads*64,generator=g)*1.5).to(dev)
    out=torch.empty(B*T,heads*64,device=dev)
    for tail in (1,0):
        L.ovm_tune_set(b"attn_tail", tail)
        rc=L.ovm_op_attention(qkv.data_ptr(),B,T,heads,out.data_ptr(),3,None)
        r=ref(qkv,B,T,heads)
        err=(out-r).abs()
        rowerr=err.amax(1)/r.abs().max()
        bad=(rowerr>5e-6).nonzero().flatten()
        print(B,T,heads,"tail",tail,"max",float(rowerr.max()),"nbad rows",len(bad), bad[:10].tolist(), bad[-5:].tolist())
        if len(bad):
            i=int(bad[0]); colerr=err[i]; print("  row",i,"bad cols", (colerr/r.abs().max()>5e-6).nonzero().flatten()[:16].tolist())
