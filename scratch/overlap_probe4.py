"""Stream-priority experiment: ViT on a LOW priority stream, detector on a HIGH priority one (raw HIP streams, full priority range)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from common import build_cfg
from ovmono3d_amd.gdino.detector import HashTokenizer
from ovmono3d_amd.gdino.engine import GdinoEngine
from ovmono3d_amd.modeling import build_model
from ovmono3d_amd.util.synth_gdino import synth_gdino_model
from ovmono3d_amd.util.synth_weights import synth_state_dict
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
hip = C.CDLL("libamdhip64.so")
lo, hi = C.c_int(), C.c_int()
hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
print("priority range: least", lo.value, "greatest", hi.value)
def prio_stream(p):
    st = C.c_void_p(); assert hip.hipStreamCreateWithPriority(C.byref(st), 1, p) == 0
    return torch.cuda.ExternalStream(st.value, device=dev)
cfg = build_cfg("vitl14", 896, "f16x3", max_batch=1, max_rois=1000)
model = build_model(cfg, device=dev); model.load_state_dict(synth_state_dict("vitl14", seed=0))
_, sd = synth_gdino_model(0)
eng = GdinoEngine(dev, sd, pixel_mean=[123.675, 116.28, 103.53], pixel_std=[58.395, 57.12, 57.375], use_graphs=True)
img = torch.randint(0, 256, (3, 532, 532), dtype=torch.uint8).to(dev)
ids = HashTokenizer().encode("chair . dining table . sofa . potted plant . television . bookcase .")
images = model.preprocess_image([{"image": img, "height": 512, "width": 512, "K": [[1024.0, 0, 256], [0, 1024.0, 256], [0, 0, 1]]}])
def ev(): return torch.cuda.Event(enable_timing=True)
for pv, pg in ((None, hi.value), (lo.value, hi.value), (lo.value, None), (hi.value, lo.value)):
    vs = prio_stream(pv) if pv is not None else torch.cuda.current_stream(dev)
    gs = prio_stream(pg) if pg is not None else torch.cuda.Stream(dev)
    def vit():
        with torch.cuda.stream(vs): model.backbone(images)
    def gd():
        with torch.cuda.stream(gs): eng.forward(img, ids)
    for _ in range(3): vit(); gd()
    torch.cuda.synchronize()
    res = {"vit": 0.0, "gd": 0.0, "span": 0.0}; N = 8
    for _ in range(N):
        m0, m1, s0, s1 = ev(), ev(), ev(), ev()
        torch.cuda.synchronize()
        s0.record(gs); gd(); s1.record(gs)
        m0.record(vs); vit(); m1.record(vs)
        torch.cuda.synchronize()
        res["vit"] += m0.elapsed_time(m1) / N; res["gd"] += s0.elapsed_time(s1) / N
        res["span"] += max(s0.elapsed_time(m1), s0.elapsed_time(s1)) / N
    print(f"ViT priority {pv}, detector priority {pg}:", {k: round(v, 2) for k, v in res.items()}, flush=True)
