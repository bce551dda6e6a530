"""Device ResizeShortestEdge ("next" row 2): Pillow is the live oracle (it is what detectron2 calls for uint8 images)."""
import numpy as np
import pytest
import torch
from PIL import Image

SIZES = [((512, 512), (532, 532)), ((480, 640), (532, 709)), ((375, 1242), (270, 896)), ((720, 1280), (504, 896)), ((97, 61), (200, 126)),
         ((33, 47), (33, 20)), ((50, 40), (17, 40)), ((1080, 1920), (512, 910))]


def _emulate(img, out_h, out_w):
    """The two fixed-point passes in numpy, on the host tables - checks the tables without a GPU."""
    from ovmono3d_amd.data.gpu_resize import pil_bilinear_tables
    a = img.astype(np.int64)
    if out_w != img.shape[1]:
        b, c = pil_bilinear_tables(img.shape[1], out_w)
        o = np.zeros((a.shape[0], out_w, a.shape[2]), np.int64)
        for xx in range(out_w):
            x0, n = b[xx]
            o[:, xx] = (1 << 21) + np.tensordot(a[:, x0:x0 + n], c[xx, :n].astype(np.int64), axes=([1], [0]))
        a = np.clip(o >> 22, 0, 255)
    if out_h != img.shape[0]:
        b, c = pil_bilinear_tables(img.shape[0], out_h)
        o = np.zeros((out_h, a.shape[1], a.shape[2]), np.int64)
        for yy in range(out_h):
            y0, n = b[yy]
            o[yy] = (1 << 21) + np.tensordot(c[yy, :n].astype(np.int64), a[y0:y0 + n], axes=([0], [0]))
        a = np.clip(o >> 22, 0, 255)
    return a.astype(np.uint8)


@pytest.mark.parametrize("src,dst", SIZES[:6])
def test_host_tables_reproduce_pillow(src, dst):
    g = np.random.default_rng(src[0] * 7 + dst[1])
    img = g.integers(0, 256, (src[0], src[1], 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((dst[1], dst[0]), Image.BILINEAR))
    assert np.array_equal(_emulate(img, dst[0], dst[1]), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", SIZES)
def test_device_resize_is_bit_identical_to_pillow(device, src, dst):
    from ovmono3d_amd.data.gpu_resize import resize_bilinear_u8
    g = np.random.default_rng(src[1] * 3 + dst[0])
    img = g.integers(0, 256, (src[0], src[1], 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((dst[1], dst[0]), Image.BILINEAR))
    got = resize_bilinear_u8(torch.from_numpy(img).to(device), dst[0], dst[1]).cpu().numpy()
    assert np.array_equal(got, ref)
    # a CHW-stored image handed over as a permuted (strided) HWC view, as the pipeline stores images
    chw = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).to(device)
    got2 = resize_bilinear_u8(chw.permute(1, 2, 0), dst[0], dst[1]).cpu().numpy()
    assert np.array_equal(got2, ref)


@pytest.mark.gpu
def test_resize_shortest_edge_gpu_matches_host_feeding(device):
    from ovmono3d_amd.data.feeding import ResizeShortestEdge
    from ovmono3d_amd.data.gpu_resize import ResizeShortestEdgeGPU
    g = np.random.default_rng(0)
    for hw in ((512, 512), (480, 640), (375, 1242), (1333, 800)):
        img = g.integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
        ref = ResizeShortestEdge(532, 896)(img)
        got = ResizeShortestEdgeGPU(532, 896)(torch.from_numpy(img).to(device)).cpu().numpy()
        assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.gpu
def test_depth_prompt_resize_on_device_matches_the_reference_mapper(device, tmp_path):
    """ovm_resize_bilinear_f32 == F.interpolate(bilinear, align_corners=False) (up, down, non-integer ratios), and the mapper with
    MODEL.AMD.GPU_RESIZE feeds the same depth prompt as the host mapper: stored map of another size -> image size ->
    ResizeShortestEdge's size (reference dataset_mapper.py:38-72)."""
    import os
    from PIL import Image
    from common import build_cfg
    from ovmono3d_amd.data.feeding import DatasetMapper3D
    from ovmono3d_amd.data.gpu_resize import resize_bilinear_f32
    g = torch.Generator().manual_seed(3)
    for (h, w), (oh, ow) in (((60, 80), (480, 640)), ((480, 640), (37, 37)), ((375, 1242), (270, 896)), ((33, 47), (33, 90)), ((5, 7), (5, 7))):
        x = torch.rand(2, h, w, generator=g) * 7.0
        ref = torch.nn.functional.interpolate(x[:, None], (oh, ow), mode="bilinear", align_corners=False)[:, 0]
        got = resize_bilinear_f32(x.to(device), oh, ow).cpu()
        assert got.shape == ref.shape and float((got - ref).abs().max()) < 2e-6 * 7.0
    with pytest.raises(RuntimeError):
        resize_bilinear_f32(torch.zeros(4, 4), 2, 2)                           # host tensor: no CPU fallback
    # the mapper: image 300 x 400, depth stored at 150 x 200
    (tmp_path / "depth" / "test").mkdir(parents=True)
    rng = np.random.default_rng(1)
    Image.fromarray(rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)).save(tmp_path / "im0.png")
    np.savez(tmp_path / "depth" / "test" / "im0.npz", depth=(rng.random((150, 200)) * 5).astype(np.float32))
    rec = {"file_name": str(tmp_path / "im0.png"), "height": 300, "width": 400, "K": np.eye(3).tolist(), "image_id": 0}
    host = DatasetMapper3D(build_cfg("vittest14", 224, extra=["INPUT.MIN_SIZE_TEST", 168, "INPUT.MAX_SIZE_TEST", 224, "MODEL.AMD.GPU_RESIZE", False]),
                           False, str(tmp_path / "depth"))(rec)
    assert host["depth"].device.type == "cpu"
    dev = DatasetMapper3D(build_cfg("vittest14", 224, extra=["INPUT.MIN_SIZE_TEST", 168, "INPUT.MAX_SIZE_TEST", 224, "MODEL.AMD.GPU_RESIZE", True,
                                                              "MODEL.DEVICE", "cuda"]), False, str(tmp_path / "depth"))(rec)
    assert dev["depth"].device.type == "cuda" and tuple(dev["depth"].shape) == tuple(host["depth"].shape) == (1, 168, 224)
    assert float((dev["depth"].cpu() - host["depth"]).abs().max()) < 1e-5
    assert torch.equal(dev["image"].cpu(), host["image"])
    # a missing depth file gives the reference's all-zero prompt on both routes (dataset_mapper.py:53-55)
    os.remove(tmp_path / "depth" / "test" / "im0.npz")
    z = DatasetMapper3D(build_cfg("vittest14", 224, extra=["INPUT.MIN_SIZE_TEST", 168, "INPUT.MAX_SIZE_TEST", 224, "MODEL.AMD.GPU_RESIZE", True,
                                                            "MODEL.DEVICE", "cuda"]), False, str(tmp_path / "depth"))(rec)
    assert tuple(z["depth"].shape) == (1, 168, 224) and float(z["depth"].abs().max()) == 0.0
