"""Detector object ``ROIHeads3DGDINO`` plugs in: native GroundingDINO network + tokenisation.

Replaces ``load_model("./configs/GroundingDINO_SwinB_cfg.py", "./checkpoints/groundingdino_swinb_cogcoor.pth")`` and the
``model(image[None], captions=[caption])`` call of reference cubercnn/modeling/roi_heads/roi_heads_gdino.py:16-23,186.
"""
from __future__ import annotations

import ctypes as C
import re
import zlib
from typing import Dict, List, Optional, Sequence

import torch

from .. import lib as _lib
from .config import GDinoConfig


class HashTokenizer:
    """Stand-in tokenizer for synthetic-weight runs (bench / smoke): one deterministic id per lower-cased word, the
    real ids of BERT's special tokens ([CLS] 101, [SEP] 102, '.' 1012, '?' 1029). NOT WordPiece - for real checkpoints
    use ``gdino_glue.WordPieceTokenizer`` with bert-base-uncased's vocab.txt."""

    def __init__(self, vocab_size: int = 30522):
        self.vocab_size = vocab_size

    def _id(self, w: str) -> int:
        if w == ".":
            return 1012
        if w == "?":
            return 1029
        return 2000 + zlib.crc32(w.encode()) % (self.vocab_size - 2000)

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        ids = [self._id(w) for w in re.findall(r"[\w']+|[.?]", text.lower())]
        return [101] + ids + [102] if add_special_tokens else ids


def convert_upstream_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """IDEA-Research/GroundingDINO checkpoint names -> the Hugging Face port's names the engine consumes.
    Written from the published module structures; it could not be checked against a real checkpoint offline."""
    out: Dict[str, torch.Tensor] = {}
    bb = "model.backbone.conv_encoder.model."

    def split3(prefix_new: str, names: Sequence[str], w: torch.Tensor, suffix: str):
        for n, part in zip(names, w.chunk(3, 0)):
            out[f"{prefix_new}{n}.{suffix}"] = part

    for k, v in sd.items():
        k = k[len("module."):] if k.startswith("module.") else k
        if k.startswith("backbone.0."):
            r = k[len("backbone.0."):]
            r = r.replace("patch_embed.proj", "embeddings.patch_embeddings.projection").replace("patch_embed.norm", "embeddings.norm")
            m = re.match(r"norm(\d)\.(weight|bias)", r)
            if m:
                out[f"{bb}hidden_states_norms.stage{int(m.group(1)) + 1}.{m.group(2)}"] = v
                continue
            if "relative_position_index" in r or "attn_mask" in r:
                continue
            m = re.match(r"layers\.(\d+)\.blocks\.(\d+)\.attn\.qkv\.(weight|bias)", r)
            if m:
                split3(f"{bb}swin.encoder.layers.{m.group(1)}.blocks.{m.group(2)}.attention.", ("q_proj", "k_proj", "v_proj"), v, m.group(3))
                continue
            r = (r.replace(".attn.proj.", ".attention.o_proj.").replace(".attn.relative_position_bias_table", ".attention.relative_position_bias.relative_position_bias_table")
                  .replace(".norm1.", ".layernorm_before.").replace(".norm2.", ".layernorm_after."))
            r = "swin.encoder." + r if r.startswith("layers.") else "swin." + r
            out[bb + r] = v
        elif k.startswith("input_proj."):
            out["model.input_proj_vision." + k[len("input_proj."):]] = v
        elif k.startswith("bert."):
            out["model.text_backbone." + k[len("bert."):]] = v
        elif k.startswith("feat_map."):
            out["model.text_projection." + k[len("feat_map."):]] = v
        elif k == "transformer.level_embed":
            out["model.level_embed"] = v
        elif k.startswith("transformer.encoder.layers."):
            r = k[len("transformer.encoder.layers."):]
            i, rest = r.split(".", 1)
            rest = (rest.replace("norm1.", "self_attn_layer_norm.").replace("norm2.", "final_layer_norm.")
                    .replace("linear1.", "fc1.").replace("linear2.", "fc2."))
            out[f"model.encoder.layers.{i}.deformable_layer.{rest}"] = v
        elif k.startswith("transformer.encoder.text_layers."):
            r = k[len("transformer.encoder.text_layers."):]
            i, rest = r.split(".", 1)
            p = f"model.encoder.layers.{i}.text_enhancer_layer."
            if rest.startswith("self_attn.in_proj_"):
                split3(p + "self_attn.", ("query", "key", "value"), v, rest.split("in_proj_")[1])
            else:
                rest = (rest.replace("norm1.", "layer_norm_before.").replace("norm2.", "layer_norm_after.")
                        .replace("linear1.", "fc1.").replace("linear2.", "fc2."))
                out[p + rest] = v
        elif k.startswith("transformer.encoder.fusion_layers."):
            r = k[len("transformer.encoder.fusion_layers."):]
            i, rest = r.split(".", 1)
            rest = (rest.replace("layer_norm_v.", "layer_norm_vision.").replace("layer_norm_l.", "layer_norm_text.")
                    .replace("attn.values_v_proj.", "attn.values_vision_proj.").replace("attn.values_l_proj.", "attn.values_text_proj.")
                    .replace("attn.out_v_proj.", "attn.out_vision_proj.").replace("attn.out_l_proj.", "attn.out_text_proj.")
                    .replace("attn.v_proj.", "attn.vision_proj.").replace("attn.l_proj.", "attn.text_proj."))
            rest = {"gamma_v": "vision_param", "gamma_l": "text_param"}.get(rest, rest)
            out[f"model.encoder.layers.{i}.fusion_layer.{rest}"] = v
        elif k.startswith("transformer.decoder.layers."):
            r = k[len("transformer.decoder.layers."):]
            i, rest = r.split(".", 1)
            p = f"model.decoder.layers.{i}."
            if rest.startswith("self_attn.in_proj_"):
                split3(p + "self_attn.", ("query", "key", "value"), v, rest.split("in_proj_")[1])
            elif rest.startswith("ca_text.in_proj_"):
                split3(p + "encoder_attn_text.", ("query", "key", "value"), v, rest.split("in_proj_")[1])
            else:
                rest = (rest.replace("cross_attn.", "encoder_attn.").replace("ca_text.", "encoder_attn_text.")
                        .replace("catext_norm.", "encoder_attn_text_layer_norm.").replace("norm1.", "encoder_attn_layer_norm.")
                        .replace("norm2.", "self_attn_layer_norm.").replace("norm3.", "final_layer_norm.")
                        .replace("linear1.", "fc1.").replace("linear2.", "fc2."))
                out[p + rest] = v
        elif k.startswith("transformer.decoder.norm."):
            out["model.decoder.layer_norm." + k.rsplit(".", 1)[1]] = v
        elif k.startswith("transformer.decoder.ref_point_head."):
            out["model.decoder.reference_points_head." + k[len("transformer.decoder.ref_point_head."):]] = v
        elif k.startswith("transformer.decoder.bbox_embed.") or k.startswith("bbox_embed."):
            out["bbox_embed." + k.split("bbox_embed.", 1)[1]] = v
        elif k.startswith("transformer.enc_output_norm."):
            out["model.enc_output_norm." + k.rsplit(".", 1)[1]] = v
        elif k.startswith("transformer.enc_output."):
            out["model.enc_output." + k.rsplit(".", 1)[1]] = v
        elif k.startswith("transformer.enc_out_bbox_embed."):
            out["model.encoder_output_bbox_embed." + k[len("transformer.enc_out_bbox_embed."):]] = v
        elif k == "transformer.tgt_embed.weight":
            out["model.query_position_embeddings.weight"] = v
    return out


class NativeGroundingDino:
    """callable(image_u8_chw, caption) -> raw network outputs + token ids, as ROIHeads3DGDINO.forward expects.

    The network runs inside libovm3d (``ovm_gdino_forward``, ``gdino/engine.py``): C++ sequencing, fused kernels, one plan + HIP
    graph per (image size, caption), scratch owned by the plan. There is no other route: without the HIP library the constructor
    raises. (The round-1 form - generic ``ovm_g_*`` ops sequenced from Python - lives on under tests/pyref_gdino/ as an independent
    cross-check of the engine; it is test infrastructure and not importable from the package.)"""

    def __init__(self, device: torch.device, state_dict: Dict[str, torch.Tensor], tokenizer, pixel_mean, pixel_std,
                 cfg: GDinoConfig = GDinoConfig(), precision: int = 3, use_graphs: bool = True, max_plans: int = 0,
                 plan_budget_mb: int = 0):
        if "model.text_projection.weight" not in state_dict:
            state_dict = convert_upstream_state_dict(state_dict)
        self.tok, self.mean, self.std = tokenizer, list(pixel_mean), list(pixel_std)
        self.use_graphs = use_graphs
        self._tok_cache: Dict[str, tuple] = {}
        from .engine import GdinoEngine
        # the reference hands GroundingDINO images[0][[2,1,0]]: the normalised, unpadded image with channels flipped (:146)
        self.engine = GdinoEngine(device, state_dict, cfg, pixel_mean=self.mean, pixel_std=self.std, flip_channels=True,
                                  precision=precision, use_graphs=use_graphs, max_plans=max_plans, plan_budget_mb=plan_budget_mb)
        self.dev = device

    def _tokens(self, caption: str):
        t = self._tok_cache.get(caption)
        if t is None:
            ids = self.tok.encode(caption)
            phrases = [p.strip() for p in caption.rstrip(" .").split(" . ")]
            phrase_ids = [self.tok.encode(p, add_special_tokens=False) for p in phrases]
            if len(self._tok_cache) > 256:
                self._tok_cache.clear()
            t = self._tok_cache[caption] = (ids, phrase_ids, torch.tensor(ids, dtype=torch.int64))
        return t

    def __call__(self, image_u8_chw: torch.Tensor, caption: str) -> Dict:
        im = image_u8_chw.to(self.dev)
        ids, phrase_ids, _ = self._tokens(caption)
        logits, boxes = self.engine.forward(im, ids)
        return {"pred_logits": logits, "pred_boxes": boxes, "input_ids": ids, "phrase_ids": phrase_ids}
