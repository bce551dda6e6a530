"""TEST INFRASTRUCTURE (oracle): numpy restatement of libjpeg-turbo's reconstruction of a baseline JPEG from its quantised DCT
coefficients - the decoder behind the reference's image reads (cv2.imread at /root/reference/demo/demo.py:52; detectron2's
``read_image`` = Pillow at /root/reference/cubercnn/data/dataset_mapper.py:38). Third-party dependency, absent from /root/reference:
libjpeg-turbo (3.1.x inside the Pillow of this image; the algorithms are unchanged since 1.x). Restated from the published
algorithms: ``jidctint.c:jpeg_idct_islow`` (dequantise, column pass descaled by 11 bits, row pass by 18, +128, clamp),
``jdsample.c:h2v1_fancy_upsample / h2v2_fancy_upsample`` (+ the replicated context rows of ``jdmainct.c``, plain replication when
the downsampled width is <= 2) and ``jdcolor.c:ycc_rgb_convert``.

PINNED: ``tests/test_jpeg.py`` checks this restatement bit for bit against Pillow itself (the live reference decoder) on the COCO
example image of the reference and on JPEGs Pillow writes at every subsampling / quality / size class - no tolerance.

Only tests may import this module; the product decodes on the device (ovmono3d_amd/csrc/jpeg.hip)."""
from __future__ import annotations

import numpy as np

F = dict(f0298=2446, f0390=3196, f0541=4433, f0765=6270, f0899=7373, f1175=9633, f1501=12299, f1847=15137, f1961=16069,
         f2053=16819, f2562=20995, f3072=25172)


def _idct_1d(i):
    """i: [8, ...] int64 (the 8 inputs of one 1-D pass along axis 0) -> [8, ...] outputs before the descale (jidctint.c even / odd parts)."""
    z1 = (i[2] + i[6]) * F["f0541"]
    tmp2 = z1 - i[6] * F["f1847"]
    tmp3 = z1 + i[2] * F["f0765"]
    tmp0 = (i[0] + i[4]) << 13
    tmp1 = (i[0] - i[4]) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = i[7], i[5], i[3], i[1]
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * F["f1175"]
    t0, t1, t2, t3 = t0 * F["f0298"], t1 * F["f2053"], t2 * F["f3072"], t3 * F["f1501"]
    z1, z2, z3, z4 = -z1 * F["f0899"], -z2 * F["f2562"], -z3 * F["f1961"] + z5, -z4 * F["f0390"] + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    return np.stack([tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3])


def idct_islow(coef: np.ndarray, qt: np.ndarray) -> np.ndarray:
    """coef [n, 64] int16 (natural order), qt [64] -> samples [n, 8, 8] uint8."""
    x = coef.astype(np.int64).reshape(-1, 8, 8) * qt.astype(np.int64).reshape(1, 8, 8)      # [n, row, col]
    ws = (_idct_1d(np.moveaxis(x, 1, 0)) + (1 << 10)) >> 11                                   # pass 1 along rows index (columns), [8(row), n, col]
    ws = np.moveaxis(ws, 0, 1)                                                                # [n, row, col]
    out = (_idct_1d(np.moveaxis(ws, 2, 0)) + (1 << 17)) >> 18                                 # pass 2 along cols, [8(col), n, row]
    out = np.moveaxis(out, 0, 2) + 128
    return np.clip(out, 0, 255).astype(np.uint8)


def plane_from_blocks(samples: np.ndarray, bw: int, bh: int) -> np.ndarray:
    return samples.reshape(bh, bw, 8, 8).transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)


def upsample_h2v1(p: np.ndarray, out_w: int) -> np.ndarray:
    """p [rows, cw] (true downsampled width) -> [rows, 2*cw][:, :out_w]"""
    cw = p.shape[1]
    a = p.astype(np.int64)
    if cw <= 2:
        return np.repeat(p, 2, axis=1)[:, :out_w]
    left = np.concatenate([a[:, :1], a[:, :-1]], axis=1)
    right = np.concatenate([a[:, 1:], a[:, -1:]], axis=1)
    even = (3 * a + left + 1) >> 2
    odd = (3 * a + right + 2) >> 2
    even[:, 0] = a[:, 0]
    odd[:, -1] = a[:, -1]
    out = np.empty((p.shape[0], 2 * cw), np.int64)
    out[:, 0::2], out[:, 1::2] = even, odd
    return out[:, :out_w].astype(np.uint8)


def upsample_h2v2(p: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """p [ch, cw] (true downsampled size) -> [out_h, out_w]"""
    ch, cw = p.shape
    if cw <= 2:
        return np.repeat(np.repeat(p, 2, axis=0), 2, axis=1)[:out_h, :out_w]
    a = p.astype(np.int64)
    up = np.concatenate([a[:1], a[:-1]], axis=0)                 # the row above (first row replicated)
    dn = np.concatenate([a[1:], a[-1:]], axis=0)                 # the row below (last row replicated)
    rows = np.empty((2 * ch, cw), np.int64)
    rows[0::2], rows[1::2] = 3 * a + up, 3 * a + dn              # column sums of the two output rows of each input row
    left = np.concatenate([rows[:, :1], rows[:, :-1]], axis=1)
    right = np.concatenate([rows[:, 1:], rows[:, -1:]], axis=1)
    even = (3 * rows + left + 8) >> 4
    odd = (3 * rows + right + 7) >> 4
    even[:, 0] = (4 * rows[:, 0] + 8) >> 4
    odd[:, -1] = (4 * rows[:, -1] + 7) >> 4
    out = np.empty((2 * ch, 2 * cw), np.int64)
    out[:, 0::2], out[:, 1::2] = even, odd
    return out[:out_h, :out_w].astype(np.uint8)


def ycc_to_rgb(y: np.ndarray, cb: np.ndarray, cr: np.ndarray) -> np.ndarray:
    y, cb, cr = y.astype(np.int64), cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def reconstruct(coef: np.ndarray, info) -> np.ndarray:
    """coef [coef_blocks, 64] int16 + the header record (fields of include/ovm3d.h OvmJpegInfo) -> RGB [H, W, 3] uint8."""
    H, W, nc = int(info.height), int(info.width), int(info.ncomp)
    planes, off = [], 0
    for c in range(nc):
        bw, bh = int(info.bw[c]), int(info.bh[c])
        qt = np.array(list(info.qt[int(info.qidx[c])]), dtype=np.int64)
        s = idct_islow(coef[off:off + bw * bh], qt)
        planes.append(plane_from_blocks(s, bw, bh)[:int(info.ch[c]), :int(info.cw[c])])
        off += bw * bh
    if nc == 1:
        return np.repeat(planes[0][:, :, None], 3, axis=2)
    hs, vs = int(info.hmax), int(info.vmax)
    if (hs, vs) == (1, 1):
        c1, c2 = planes[1], planes[2]
    elif (hs, vs) == (2, 1):
        c1, c2 = upsample_h2v1(planes[1], W), upsample_h2v1(planes[2], W)
    elif (hs, vs) == (2, 2):
        c1, c2 = upsample_h2v2(planes[1], H, W), upsample_h2v2(planes[2], H, W)
    else:
        raise ValueError("sampling outside the decoder's scope")
    if int(info.colorspace) == 2:
        return np.stack([planes[0], c1, c2], axis=-1)
    return ycc_to_rgb(planes[0], c1, c2)
