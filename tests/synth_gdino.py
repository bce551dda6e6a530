"""Random-init GroundingDINO (Swin-B + BERT-base) state dict in the Hugging Face port's naming, for benchmarking and
smoke tests when no checkpoint can be fetched. Shapes come from instantiating the HF model class from its config (no
download); weights are re-drawn from a seeded generator so the fusion gates / heads are not inert."""
from __future__ import annotations

from typing import Dict

import torch


def synth_gdino_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    return synth_gdino_model(seed)[1]


def synth_gdino_model(seed: int = 0):
    """-> (HF GroundingDinoForObjectDetection carrying the perturbed weights, its state dict)."""
    from transformers import BertConfig, GroundingDinoConfig, GroundingDinoForObjectDetection, SwinConfig
    bb = SwinConfig(image_size=384, patch_size=4, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=12,
                    out_indices=[2, 3, 4], layer_norm_eps=1e-5)
    tc = BertConfig(attn_implementation="eager")                                         # bert-base-uncased dimensions
    cfg = GroundingDinoConfig(backbone_config=bb, text_config=tc, d_model=256, encoder_layers=6, decoder_layers=6,
                              encoder_attention_heads=8, decoder_attention_heads=8, encoder_ffn_dim=2048, decoder_ffn_dim=2048,
                              num_queries=900, num_feature_levels=4, max_text_len=256, positional_embedding_temperature=20,
                              two_stage=True, embedding_init_target=True, decoder_bbox_embed_share=True, disable_custom_kernels=True,
                              attn_implementation="eager")
    torch.manual_seed(seed)
    hf = GroundingDinoForObjectDetection(cfg)
    g = torch.Generator().manual_seed(seed + 1)
    sd = {}
    for k, v in hf.state_dict().items():
        v = v.detach().clone()
        if v.dtype.is_floating_point:
            if v.dim() > 1:
                v.add_(torch.randn(v.shape, generator=g) * 0.02)
            elif k.endswith("vision_param") or k.endswith("text_param"):
                v.copy_(0.3 + 0.4 * torch.rand(v.shape, generator=g))
        sd[k] = v
    hf.load_state_dict(sd)
    # tied modules (decoder_bbox_embed_share) list one tensor under several names: re-read so every alias agrees
    sd = {k: v.detach().clone() for k, v in hf.state_dict().items()}
    return hf.eval(), sd
