"""GroundingDINO output glue of ROIHeads3DGDINO (fp32, CPU) - the reference-owned part of scope row a10.

Restates reference cubercnn/modeling/roi_heads/roi_heads_gdino.py:
  caption building          :176-181 (get_grounding_output)
  phrase spans / logits     :273-294 (get_phrase_logits_from_token_logits)
  threshold / argmax        :187-202
  cxcywh -> xyxy, NMS       :236-263, :266-270 (grounding_dino_inference_detector, box_cxcywh_to_xyxy)
  class index               :162 (filtered_texts.index([class_name]))
The GroundingDINO network that produces ``pred_logits`` / ``pred_boxes`` (third-party, IDEA-Research/GroundingDINO
@856dde2, not in the reference tree) is outside this file.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from .roi_ops import nms


def build_caption(category_list: Sequence[str]) -> Tuple[str, List[str]]:
    cap_list = [c for c in category_list]
    caption = " . ".join(cap_list)
    caption = caption.lower()
    caption = caption.strip()
    if not caption.endswith("."):
        caption = caption + " ."
    return caption, cap_list


def phrase_spans(caption_ids: Sequence[int], phrases_ids: Sequence[Sequence[int]]) -> List[Tuple[int, int]]:
    """:277-291. caption_ids = tokenizer(caption)['input_ids'] ([CLS] ... [SEP]); phrases_ids = tokenizer(cap_list,
    add_special_tokens=False)['input_ids']."""
    begin = 1
    spans = []
    for toks in phrases_ids:
        end = begin + len(toks)
        assert list(toks) == list(caption_ids[begin:end]), "assert error!!!"
        spans.append((begin, end))
        begin = end + 1
    return spans


def gdino_postprocess(pred_logits: torch.Tensor, pred_boxes: torch.Tensor, spans: Sequence[Tuple[int, int]], cap_list: Sequence[str],
                      filtered_texts: Sequence[Sequence[str]], image_hw: Tuple[int, int], box_threshold: float = 0.001,
                      nms_threshold: float = 0.5):
    """pred_logits [nq,256] (pre-sigmoid), pred_boxes [nq,4] cxcywh in [0,1].
    Returns boxes [n,4] xyxy pixels, scores [n], pred_classes [n] (int64), in NMS keep order."""
    logits = pred_logits.sigmoid()                                                     # :187
    phrase_logits = torch.stack([logits[:, b:e].sum(dim=-1) for b, e in spans], dim=1)  # :287-291
    filt = phrase_logits.max(dim=1)[0] > box_threshold                                 # :197
    im_logits = phrase_logits[filt]
    boxes = pred_boxes[filt]
    scores, cls = im_logits.max(dim=-1)                                                # :201
    phrases = [cap_list[int(i)] for i in cls]
    h, w = image_hw
    b = boxes * torch.tensor([w, h, w, h], dtype=boxes.dtype)                          # :253
    xc, yc, bw, bh = b.unbind(-1)
    xyxy = torch.stack([xc - 0.5 * bw, yc - 0.5 * bh, xc + 0.5 * bw, yc + 0.5 * bh], dim=-1)
    keep = nms(xyxy, scores, nms_threshold)                                            # :254
    classes = torch.tensor([list(filtered_texts).index([phrases[int(i)]]) for i in keep], dtype=torch.int64)   # :162
    return xyxy[keep], scores[keep], classes
