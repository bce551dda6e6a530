import sys, ctypes as C, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from ovmono3d_amd import lib
from oracle.roi_ops import roi_pooler
L=lib.load(); dev=torch.device('cuda')
g = torch.Generator().manual_seed(3)
Cc, B = 64, 2
feats = [torch.randn(B, Cc, s, s, generator=g) for s in (32, 16, 8)]
scales = [1 / 7, 1 / 14, 1 / 28]
n = 40
x1 = torch.rand(n, generator=g) * 150 - 10
y1 = torch.rand(n, generator=g) * 150 - 10
w = torch.rand(n, generator=g) * 200 + 1
h = torch.rand(n, generator=g) * 200 + 1
boxes = torch.stack([x1, y1, x1 + w, y1 + h], 1)
boxes[0] = torch.tensor([5.0, 5.0, 5.0, 9.0]); boxes[1] = torch.tensor([-50.0, -50.0, 400.0, 400.0]); boxes[2] = torch.tensor([10.0, 10.0, 10.5, 10.5])
idx = torch.cat([torch.zeros(25, dtype=torch.int32), torch.ones(15, dtype=torch.int32)])
ref = roi_pooler(feats, [boxes[:25], boxes[25:]], scales, 7, 2, 4).permute(0, 2, 3, 1).reshape(n, -1)
nhwc = [f.permute(0, 2, 3, 1).contiguous().to(dev) for f in feats]
hw = (C.c_int32 * 6)(32, 32, 16, 16, 8, 8); sc = (C.c_float * 3)(*scales)
out = torch.empty(n, 49 * Cc, device=dev)
rc = L.ovm_op_roi_align(nhwc[0].data_ptr(), nhwc[1].data_ptr(), nhwc[2].data_ptr(), hw, sc, Cc, 7, 2, 4, boxes.to(dev).data_ptr(), idx.to(dev).data_ptr(), n, out.data_ptr(), None)
out=out.cpu()
err=(out-ref).abs().amax(1)
for i in range(n):
    if err[i]>1e-5: print(i, boxes[i].tolist(), float(err[i]), (out[i]-ref[i]).abs().view(49,Cc).amax(1).view(7,7))
