"""Torch-tensor front end of the ovm_g_* device ops (device memory handles only; no torch arithmetic)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from ovmono3d_amd import lib as _lib

ADD, MUL, RELU, GELU, SIGMOID, CLAMP, AXPY, INVSIG, COPY, MASKFILL = range(10)     # ovm_g_eltwise op codes
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2                                               # ovm_g_linear activation codes


@dataclass
class PackedW:
    hi: torch.Tensor
    lo: torch.Tensor
    N: int
    K: int
    Kpad: int
    bias: Optional[torch.Tensor] = None


class Ops:
    def __init__(self, device: torch.device, precision: int = 3):
        if device.type != "cuda":
            raise RuntimeError("native ops run on the HIP device only (no CPU fallback)")
        self.dev, self.prec, self.L = device, precision, _lib.load()

    def _s(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def _chk(self, rc, what):
        _lib.check(rc, what=what)

    def f32(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(self.dev, torch.float32).contiguous()

    def empty(self, *shape) -> torch.Tensor:
        return torch.empty(*shape, dtype=torch.float32, device=self.dev)

    # ---- dense projections ----------------------------------------------------------------
    def pack(self, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> PackedW:
        w = self.f32(w.reshape(w.shape[0], -1))
        N, K = w.shape
        Kpad = (K + 63) // 64 * 64
        Npad = (N + 127) // 128 * 128
        if self.prec == 3:
            # split mode: one interleaved image [Npad][Kpad/32][hi 32 | lo 32]; lo is the same buffer 32 halves in
            hi = torch.empty((Npad, 2 * Kpad), dtype=torch.float16, device=self.dev)
            lo = hi.view(-1)[32:]
        else:
            hi, lo = torch.empty((Npad, Kpad), dtype=torch.float16, device=self.dev), None
        self._chk(self.L.ovm_g_pack_weight(w.data_ptr(), N, K, Kpad, hi.data_ptr(), lo.data_ptr() if lo is not None else None, self._s()),
                  "ovm_g_pack_weight")
        return PackedW(hi, lo, N, K, Kpad, self.f32(bias) if bias is not None else None)

    def linear(self, x: torch.Tensor, W: PackedW, act: int = 0, residual: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        assert x2.shape[1] == W.K and x2.stride(1) == 1, (x2.shape, W.K)
        M = x2.shape[0]
        y = out if out is not None else self.empty(M, W.N)
        r = residual.reshape(M, W.N) if residual is not None else None
        self._chk(self.L.ovm_g_linear(x2.data_ptr(), x2.stride(0), M, W.K, W.hi.data_ptr(), W.lo.data_ptr() if W.lo is not None else None, W.N, W.Kpad,
                                      W.bias.data_ptr() if W.bias is not None else None, act,
                                      r.data_ptr() if r is not None else None, r.stride(0) if r is not None else 0,
                                      y.data_ptr(), y.stride(0), self.prec, self._s()), "ovm_g_linear")
        return y.reshape(*shp[:-1], W.N) if out is None else y

    def layernorm(self, x, g, b, eps, residual=None):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        r2 = residual.reshape(-1, shp[-1]).contiguous() if residual is not None else None
        y = torch.empty_like(x2)
        self._chk(self.L.ovm_g_layernorm(x2.data_ptr(), r2.data_ptr() if r2 is not None else None, x2.shape[0], x2.shape[1],
                                         g.data_ptr(), b.data_ptr(), float(eps), y.data_ptr(), self._s()), "ovm_g_layernorm")
        return y.reshape(shp)

    # ---- small matmuls / softmax ------------------------------------------------------------
    def bmm_raw(self, a, a_off, b, b_off, c, c_off, batch, M, N, K, lda, ldb, ldc, sA, sB, sC, transB, alpha=1.0):
        es = 4
        self._chk(self.L.ovm_g_bmm(a.data_ptr() + a_off * es, b.data_ptr() + b_off * es, c.data_ptr() + c_off * es, batch, M, N, K,
                                   lda, ldb, ldc, sA, sB, sC, int(transB), float(alpha), self._s()), "ovm_g_bmm")

    def bmm2_raw(self, a, a_off, b, b_off, c, c_off, nb1, nb2, M, N, K, lda, ldb, ldc, sA1, sB1, sC1, sA2, sB2, sC2, transB, alpha=1.0):
        """two-level batch (z = z1*nb2 + z2), e.g. (window, head)."""
        es = 4
        self._chk(self.L.ovm_g_bmm2(a.data_ptr() + a_off * es, b.data_ptr() + b_off * es, c.data_ptr() + c_off * es, nb1, nb2, M, N, K,
                                    lda, ldb, ldc, sA1, sB1, sC1, sA2, sB2, sC2, int(transB), float(alpha), self._s()), "ovm_g_bmm2")

    def bmm(self, a: torch.Tensor, b: torch.Tensor, transB: bool, alpha: float = 1.0) -> torch.Tensor:
        """a [Bt,M,K]; b [Bt,N,K] if transB else [Bt,K,N]; contiguous."""
        a, b = a.contiguous(), b.contiguous()
        Bt, M, K = a.shape
        N = b.shape[1] if transB else b.shape[2]
        c = self.empty(Bt, M, N)
        self.bmm_raw(a, 0, b, 0, c, 0, Bt, M, N, K, K, b.shape[2], N, M * K, b.shape[1] * b.shape[2], M * N, transB, alpha)
        return c

    def softmax_(self, x: torch.Tensor, bias: Optional[torch.Tensor] = None, bias_rows: int = 1, bias_div: int = 1):
        rows, cols = x.numel() // x.shape[-1], x.shape[-1]
        assert x.is_contiguous()
        self._chk(self.L.ovm_g_softmax(x.data_ptr(), rows, cols, cols, bias.data_ptr() if bias is not None else None, bias_rows, bias_div,
                                       bias.shape[-1] if bias is not None else 0, self._s()), "ovm_g_softmax")
        return x

    def softmax2_(self, x: torch.Tensor, bias, bias_rows, bias_div, bias2, d2, m2):
        """x[r] = softmax(x[r] + bias[(r//bias_div) % bias_rows] + bias2[(r//d2)*m2 + r % m2])"""
        rows, cols = x.numel() // x.shape[-1], x.shape[-1]
        assert x.is_contiguous()
        self._chk(self.L.ovm_g_softmax2(x.data_ptr(), rows, cols, cols, bias.data_ptr() if bias is not None else None, bias_rows, bias_div,
                                        cols, bias2.data_ptr() if bias2 is not None else None, d2, m2, self._s()), "ovm_g_softmax2")
        return x

    # ---- element-wise / gathers ---------------------------------------------------------------
    def elt(self, op, a, b=None, alpha=0.0, beta=0.0, out=None):
        a = a.contiguous()
        o = out if out is not None else torch.empty_like(a)
        bmod = 0
        if b is not None:
            b = b.contiguous()
            bmod = b.numel()
            assert a.numel() % bmod == 0
        self._chk(self.L.ovm_g_eltwise(op, a.data_ptr(), b.data_ptr() if b is not None else None, o.data_ptr(), a.numel(), bmod,
                                       float(alpha), float(beta), self._s()), "ovm_g_eltwise")
        return o

    def add(self, a, b):
        return self.elt(ADD, a, b)

    def gather_rows(self, src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """src [R,Cc] (row stride may exceed Cc); idx int32 [n, nidx] -> [n, nidx*Cc]; negative index -> zeros."""
        assert src.dim() == 2 and src.stride(1) == 1
        idx = idx.to(self.dev, torch.int32).contiguous()
        n, nidx = idx.shape
        cols = src.shape[1]
        dst = self.empty(n, nidx * cols)
        self._chk(self.L.ovm_g_gather_rows(src.data_ptr(), src.stride(0), idx.data_ptr(), n, nidx, cols, dst.data_ptr(), self._s()),
                  "ovm_g_gather_rows")
        return dst

    def groupnorm(self, x: torch.Tensor, groups: int, g, b, eps: float) -> torch.Tensor:
        """x [B,HW,Cc] (NHWC flattened)."""
        x = x.contiguous()
        y = torch.empty_like(x)
        self._chk(self.L.ovm_g_groupnorm(x.data_ptr(), x.shape[0], x.shape[1], x.shape[2], groups, g.data_ptr(), b.data_ptr(), float(eps),
                                         y.data_ptr(), self._s()), "ovm_g_groupnorm")
        return y

    def msdeform(self, value: torch.Tensor, shapes: Sequence, loc: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """value [B,S,H,dh]; loc [B,Q,H,L,P,2]; w [B,Q,H,L,P] -> [B,Q,H*dh]."""
        value, loc, w = value.contiguous(), loc.contiguous(), w.contiguous()
        B, S, H, dh = value.shape
        Q, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
        sh = (C.c_int32 * (2 * L))(*[int(v) for hw in shapes for v in hw])
        out = self.empty(B, Q, H * dh)
        self._chk(self.L.ovm_g_msdeform(value.data_ptr(), sh, L, B, S, Q, H, dh, P, loc.data_ptr(), w.data_ptr(), out.data_ptr(), self._s()),
                  "ovm_g_msdeform")
        return out

    def sine_embed(self, pos: torch.Tensor, F: int, temperature: float) -> torch.Tensor:
        pos = pos.contiguous()
        n, nc = pos.numel() // pos.shape[-1], pos.shape[-1]
        out = self.empty(*pos.shape[:-1], nc * F)
        self._chk(self.L.ovm_g_sine_embed(pos.data_ptr(), n, nc, F, float(temperature), out.data_ptr(), self._s()), "ovm_g_sine_embed")
        return out

    def rowmax(self, x: torch.Tensor) -> torch.Tensor:
        assert x.dim() == 2 and x.stride(1) == 1
        out = self.empty(x.shape[0])
        self._chk(self.L.ovm_g_rowmax(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), out.data_ptr(), self._s()), "ovm_g_rowmax")
        return out

    def normalize_image(self, image_desc, mean, std, flip: bool) -> torch.Tensor:
        """image_desc: lib.OvmImage (uint8, device). Returns fp32 [H*W, 3] NHWC, optionally channel-flipped."""
        H, W = int(image_desc.height), int(image_desc.width)
        out = self.empty(H * W, 3)
        m = (C.c_float * 3)(*[float(v) for v in mean])
        s = (C.c_float * 3)(*[float(v) for v in std])
        self._chk(self.L.ovm_g_normalize_image(C.byref(image_desc), m, s, int(flip), out.data_ptr(), self._s()), "ovm_g_normalize_image")
        return out

    def topk(self, scores: torch.Tensor, k: int) -> torch.Tensor:
        scores = scores.contiguous()
        out = torch.empty(k, dtype=torch.int32, device=self.dev)
        self._chk(self.L.ovm_g_topk(scores.data_ptr(), scores.numel(), k, out.data_ptr(), self._s()), "ovm_g_topk")
        return out
