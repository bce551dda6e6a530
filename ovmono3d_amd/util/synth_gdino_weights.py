"""Random-init GroundingDINO weights (Swin + BERT + fusion encoder + decoder) in the parameter naming the engine consumes (the
Hugging Face port's), built from the architecture record alone - no model class is instantiated and nothing is downloaded. For
``MODEL.AMD.GDINO_WEIGHTS synthetic://gdino?seed=N`` (smoke runs and entry-point tests when no checkpoint can be fetched); the
key set and shapes are held equal to the Hugging Face port's by tests/test_gdino_convert.py."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

from ..gdino.config import GDinoConfig


def synth_gdino_state_dict(seed: int = 0, cfg: GDinoConfig = GDinoConfig(), bert_hidden: int = 768, bert_layers: int = 12,
                           bert_ffn: int = 3072, vocab: int = 30522, bert_positions: int = 512) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def mat(name: str, *shape: int, std: float = 0.0):
        fan_out, fan_in = shape[0], int(torch.tensor(shape[1:]).prod()) if len(shape) > 1 else shape[0]
        s = std if std > 0 else math.sqrt(2.0 / (fan_in + fan_out))
        sd[name] = torch.randn(shape, generator=g) * s

    def lin(prefix: str, n_out: int, n_in: int):
        mat(prefix + ".weight", n_out, n_in)
        sd[prefix + ".bias"] = torch.randn(n_out, generator=g) * 0.02

    def norm(prefix: str, n: int):
        sd[prefix + ".weight"] = 1.0 + torch.randn(n, generator=g) * 0.02
        sd[prefix + ".bias"] = torch.randn(n, generator=g) * 0.02

    D, H = cfg.d_model, cfg.heads
    # ---- Swin backbone
    sw = "model.backbone.conv_encoder.model.swin."
    E = cfg.swin_embed
    mat(sw + "embeddings.patch_embeddings.projection.weight", E, 3, 4, 4)
    sd[sw + "embeddings.patch_embeddings.projection.bias"] = torch.randn(E, generator=g) * 0.02
    norm(sw + "embeddings.norm", E)
    ws = cfg.swin_window
    n_stages = len(cfg.swin_depths)
    for i, (depth, heads) in enumerate(zip(cfg.swin_depths, cfg.swin_heads)):
        C = E << i
        for j in range(depth):
            b = f"{sw}encoder.layers.{i}.blocks.{j}."
            for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
                lin(b + "attention." + n, C, C)
            sd[b + "attention.relative_position_bias.relative_position_bias_table"] = torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.02
            norm(b + "layernorm_before", C)
            norm(b + "layernorm_after", C)
            lin(b + "mlp.fc1", 4 * C, C)
            lin(b + "mlp.fc2", C, 4 * C)
        if i < n_stages - 1:
            mat(f"{sw}encoder.layers.{i}.downsample.reduction.weight", 2 * C, 4 * C)
            norm(f"{sw}encoder.layers.{i}.downsample.norm", 4 * C)
    norm(sw + "layernorm", E << (n_stages - 1))
    # the detector taps the last n_levels - 1 stages; the extra level is a 3x3 / stride-2 conv on the last one
    taps = list(range(n_stages - (cfg.n_levels - 1), n_stages))
    for i in taps:
        norm(f"model.backbone.conv_encoder.model.hidden_states_norms.stage{i + 1}", E << i)
    for lvl, i in enumerate(taps):
        mat(f"model.input_proj_vision.{lvl}.0.weight", D, E << i, 1, 1)
        sd[f"model.input_proj_vision.{lvl}.0.bias"] = torch.randn(D, generator=g) * 0.02
        norm(f"model.input_proj_vision.{lvl}.1", D)
    mat(f"model.input_proj_vision.{len(taps)}.0.weight", D, E << taps[-1], 3, 3)
    sd[f"model.input_proj_vision.{len(taps)}.0.bias"] = torch.randn(D, generator=g) * 0.02
    norm(f"model.input_proj_vision.{len(taps)}.1", D)
    sd["model.level_embed"] = torch.randn(cfg.n_levels, D, generator=g)
    # ---- BERT text encoder
    tb = "model.text_backbone."
    mat(tb + "embeddings.word_embeddings.weight", vocab, bert_hidden, std=0.02)
    mat(tb + "embeddings.position_embeddings.weight", bert_positions, bert_hidden, std=0.02)
    mat(tb + "embeddings.token_type_embeddings.weight", 2, bert_hidden, std=0.02)
    norm(tb + "embeddings.LayerNorm", bert_hidden)
    for l in range(bert_layers):
        b = f"{tb}encoder.layer.{l}."
        for n in ("query", "key", "value"):
            lin(b + "attention.self." + n, bert_hidden, bert_hidden)
        lin(b + "attention.output.dense", bert_hidden, bert_hidden)
        norm(b + "attention.output.LayerNorm", bert_hidden)
        lin(b + "intermediate.dense", bert_ffn, bert_hidden)
        lin(b + "output.dense", bert_hidden, bert_ffn)
        norm(b + "output.LayerNorm", bert_hidden)
    lin("model.text_projection", D, bert_hidden)
    mat("model.query_position_embeddings.weight", cfg.num_queries, D, std=1.0)
    # ---- fusion encoder
    F2 = cfg.ffn_dim // 2                          # text-enhancer FFN width and the fusion attention's embedding (1024)
    n_samp = H * cfg.n_levels * cfg.n_points

    def deform(prefix: str):
        lin(prefix + "sampling_offsets", 2 * n_samp, D)
        lin(prefix + "attention_weights", n_samp, D)
        lin(prefix + "value_proj", D, D)
        lin(prefix + "output_proj", D, D)

    def mha(prefix: str):
        for n in ("query", "key", "value", "out_proj"):
            lin(prefix + n, D, D)

    for l in range(cfg.enc_layers):
        e = f"model.encoder.layers.{l}."
        mha(e + "text_enhancer_layer.self_attn.")
        lin(e + "text_enhancer_layer.fc1", F2, D)
        lin(e + "text_enhancer_layer.fc2", D, F2)
        norm(e + "text_enhancer_layer.layer_norm_before", D)
        norm(e + "text_enhancer_layer.layer_norm_after", D)
        sd[e + "fusion_layer.vision_param"] = 0.3 + 0.4 * torch.rand(D, generator=g)      # layer-scale of the fusion branch: not inert
        sd[e + "fusion_layer.text_param"] = 0.3 + 0.4 * torch.rand(D, generator=g)
        norm(e + "fusion_layer.layer_norm_vision", D)
        norm(e + "fusion_layer.layer_norm_text", D)
        for n in ("vision_proj", "text_proj", "values_vision_proj", "values_text_proj"):
            lin(e + "fusion_layer.attn." + n, F2, D)
        lin(e + "fusion_layer.attn.out_vision_proj", D, F2)
        lin(e + "fusion_layer.attn.out_text_proj", D, F2)
        deform(e + "deformable_layer.self_attn.")
        norm(e + "deformable_layer.self_attn_layer_norm", D)
        lin(e + "deformable_layer.fc1", cfg.ffn_dim, D)
        lin(e + "deformable_layer.fc2", D, cfg.ffn_dim)
        norm(e + "deformable_layer.final_layer_norm", D)
    # ---- decoder
    norm("model.decoder.layer_norm", D)
    for l in range(cfg.dec_layers):
        d = f"model.decoder.layers.{l}."
        mha(d + "self_attn.")
        norm(d + "self_attn_layer_norm", D)
        mha(d + "encoder_attn_text.")
        norm(d + "encoder_attn_text_layer_norm", D)
        deform(d + "encoder_attn.")
        norm(d + "encoder_attn_layer_norm", D)
        lin(d + "fc1", cfg.ffn_dim, D)
        lin(d + "fc2", D, cfg.ffn_dim)
        norm(d + "final_layer_norm", D)
    lin("model.decoder.reference_points_head.layers.0", D, 2 * D)
    lin("model.decoder.reference_points_head.layers.1", D, D)

    def box_mlp(prefix: str):
        lin(prefix + "layers.0", D, D)
        lin(prefix + "layers.1", D, D)
        lin(prefix + "layers.2", 4, D)

    box_mlp("bbox_embed.0.")                       # decoder_bbox_embed_share: one MLP under every alias
    for l in range(cfg.dec_layers):
        for alias in (f"bbox_embed.{l}.", f"model.decoder.bbox_embed.{l}."):
            for k in ("layers.0", "layers.1", "layers.2"):
                for wb in ("weight", "bias"):
                    sd[f"{alias}{k}.{wb}"] = sd[f"bbox_embed.0.{k}.{wb}"]
    lin("model.enc_output", D, D)
    norm("model.enc_output_norm", D)
    box_mlp("model.encoder_output_bbox_embed.")
    return sd
