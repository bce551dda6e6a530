"""Host side of the GroundingDINO glue in ROIHeads3DGDINO (reference roi_heads_gdino.py:174-294).

Text handling is host work (strings): caption building (:176-181), BERT WordPiece tokenisation and the
phrase -> token-span walk (:277-291). The arithmetic on the network outputs - sigmoid, per-phrase sum, max /
argmax, threshold, box conversion, NMS - runs in libovm3d (``ovm_gdino_postprocess``).

``bert-base-uncased``'s vocabulary cannot be fetched in the build environment; ``WordPieceTokenizer`` takes a
user-supplied ``vocab.txt`` (the file GroundingDINO's tokenizer loads) and implements the standard
BasicTokenizer(lower-case, accent strip, punctuation split) + greedy longest-match WordPiece.
"""
from __future__ import annotations

import ctypes as C
import unicodedata
from typing import Dict, List, Sequence, Tuple

import torch

from ... import lib as _lib


def build_caption(category_list: Sequence[str]) -> Tuple[str, List[str]]:
    """reference roi_heads_gdino.py:176-181"""
    cap_list = list(category_list)
    caption = " . ".join(cap_list).lower().strip()
    if not caption.endswith("."):
        caption = caption + " ."
    return caption, cap_list


def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if (33 <= cp <= 47) or (58 <= cp <= 64) or (91 <= cp <= 96) or (123 <= cp <= 126):
        return True
    return unicodedata.category(ch).startswith("P")


class WordPieceTokenizer:
    def __init__(self, vocab_file: str, unk="[UNK]", cls="[CLS]", sep="[SEP]", max_chars=100):
        with open(vocab_file, encoding="utf-8") as f:
            self.vocab: Dict[str, int] = {tok.rstrip("\n"): i for i, tok in enumerate(f)}
        self.unk, self.cls, self.sep, self.max_chars = unk, cls, sep, max_chars

    def _basic(self, text: str) -> List[str]:
        text = unicodedata.normalize("NFD", text.lower())
        text = "".join(ch for ch in text if unicodedata.category(ch) != "Mn")
        out, cur = [], ""
        for ch in text:
            if ch.isspace():
                if cur:
                    out.append(cur); cur = ""
            elif _is_punct(ch):
                if cur:
                    out.append(cur); cur = ""
                out.append(ch)
            else:
                cur += ch
        if cur:
            out.append(cur)
        return out

    def _wordpiece(self, word: str) -> List[str]:
        if len(word) > self.max_chars:
            return [self.unk]
        pieces, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                sub = word[start:end]
                if start > 0:
                    sub = "##" + sub
                if sub in self.vocab:
                    cur = sub
                    break
                end -= 1
            if cur is None:
                return [self.unk]
            pieces.append(cur)
            start = end
        return pieces

    def tokenize(self, text: str) -> List[str]:
        return [p for w in self._basic(text) for p in self._wordpiece(w)]

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        ids = [self.vocab.get(t, self.vocab[self.unk]) for t in self.tokenize(text)]
        if add_special_tokens:
            ids = [self.vocab[self.cls]] + ids + [self.vocab[self.sep]]
        return ids


def phrase_spans(caption_ids: Sequence[int], phrases_ids: Sequence[Sequence[int]]) -> List[Tuple[int, int]]:
    """reference roi_heads_gdino.py:277-291 (same assertion message)."""
    begin, spans = 1, []
    for toks in phrases_ids:
        end = begin + len(toks)
        assert list(toks) == list(caption_ids[begin:end]), "assert error!!!"
        spans.append((begin, end))
        begin = end + 1
    return spans


def gdino_postprocess(pred_logits: torch.Tensor, pred_boxes: torch.Tensor, spans: Sequence[Tuple[int, int]], image_hw,
                      box_threshold: float = 0.001, nms_threshold: float = 0.5):
    """Native glue on device tensors: returns (boxes [n,4] xyxy pixels, scores [n], phrase index [n] int64)."""
    L = _lib.load()
    dev = pred_logits.device
    if dev.type != "cuda":
        raise RuntimeError("gdino_postprocess runs on the HIP device only (no CPU fallback)")
    logits = pred_logits.to(torch.float32).contiguous()
    boxes = pred_boxes.to(torch.float32).contiguous()
    nq, ld = int(logits.shape[0]), int(logits.shape[1])
    sp = (C.c_int32 * (2 * max(len(spans), 1)))()
    for i, (b, e) in enumerate(spans):
        sp[2 * i], sp[2 * i + 1] = int(b), int(e)
    ob = torch.empty((max(nq, 1), 4), dtype=torch.float32, device=dev)
    os_ = torch.empty(max(nq, 1), dtype=torch.float32, device=dev)
    oc = torch.empty(max(nq, 1), dtype=torch.int32, device=dev)
    n = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = L.ovm_gdino_postprocess(logits.data_ptr(), nq, ld, boxes.data_ptr(), sp, len(spans), int(image_hw[0]), int(image_hw[1]),
                                 float(box_threshold), float(nms_threshold), ob.data_ptr(), os_.data_ptr(), oc.data_ptr(),
                                 n.data_ptr(), stream)
    _lib.check(rc, what="ovm_gdino_postprocess")
    k = int(n.item())
    return ob[:k], os_[:k], oc[:k].to(torch.int64)
