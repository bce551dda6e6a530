"""CU-mask experiment: the ViT's stream restricted to a subset of the CUs (hipExtStreamCreateWithCUMask), the detector's stream
unrestricted, so that its short kernels always find free CUs. OVM_MASK_MOD = m: CUs with index % m == m - 1 are withheld from the ViT."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from common import build_cfg
from ovmono3d_amd.gdino.detector import HashTokenizer
from ovmono3d_amd.gdino.engine import GdinoEngine
from ovmono3d_amd.modeling import build_model
from ovmono3d_amd.util.synth_gdino import synth_gdino_model
from ovmono3d_amd.util.synth_weights import synth_state_dict
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
hip = C.CDLL("libamdhip64.so")
def masked_stream(mod, contiguous=0):
    words = (C.c_uint32 * 8)()
    n = 0
    for i in range(256):
        off = (i >= 256 - contiguous) if contiguous else (mod and i % mod == mod - 1)
        if not off:
            words[i // 32] |= (1 << (i % 32)); n += 1
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev), n
cfg = build_cfg("vitl14", 896, "f16x3", max_batch=1, max_rois=1000)
model = build_model(cfg, device=dev); model.load_state_dict(synth_state_dict("vitl14", seed=0))
_, sd = synth_gdino_model(0)
eng = GdinoEngine(dev, sd, pixel_mean=[123.675, 116.28, 103.53], pixel_std=[58.395, 57.12, 57.375], use_graphs=True)
img = torch.randint(0, 256, (3, 532, 532), dtype=torch.uint8).to(dev)
ids = HashTokenizer().encode("chair . dining table . sofa . potted plant . television . bookcase .")
inputs = [{"image": img, "height": 512, "width": 512, "K": [[1024.0, 0, 256], [0, 1024.0, 256], [0, 0, 1]]}]
images = model.preprocess_image(inputs)
side = torch.cuda.Stream(dev, priority=-1)
def ev(): return torch.cuda.Event(enable_timing=True)
def gd():
    with torch.cuda.stream(side): eng.forward(img, ids)
for spec in ((0, 0), (16, 0), (8, 0), (0, 16), (0, 32), (4, 0)):
    vs, n = masked_stream(*spec) if spec != (0, 0) else (torch.cuda.current_stream(dev), 256)
    def vit():
        with torch.cuda.stream(vs): model.backbone(images)
    for _ in range(3): vit(); gd()
    torch.cuda.synchronize()
    res = {"vit_alone": 0.0, "vit": 0.0, "gd": 0.0, "span": 0.0}
    N = 8
    for _ in range(N):
        m0, m1 = ev(), ev()
        torch.cuda.synchronize(); m0.record(vs); vit(); m1.record(vs); torch.cuda.synchronize()
        res["vit_alone"] += m0.elapsed_time(m1) / N
        m0, m1, s0, s1 = ev(), ev(), ev(), ev()
        torch.cuda.synchronize()
        s0.record(side); gd(); s1.record(side)
        m0.record(vs); vit(); m1.record(vs)
        torch.cuda.synchronize()
        res["vit"] += m0.elapsed_time(m1) / N; res["gd"] += s0.elapsed_time(s1) / N
        res["span"] += max(s0.elapsed_time(m1), s0.elapsed_time(s1)) / N
    print(f"mask mod={spec[0]} tail={spec[1]} ViT CUs={n}:", {k: round(v, 2) for k, v in res.items()}, flush=True)
