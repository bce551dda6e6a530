"""Upstream (IDEA-Research/GroundingDINO) -> Hugging Face parameter-name conversion used when loading
groundingdino_swinb_cogcoor.pth. No upstream checkpoint exists offline, so this checks self-consistency: an upstream-style state
dict is synthesised from the HF-named one by the inverse rules below (written independently, from the two module trees), and
the converter must give back every tensor the native network reads, bit for bit (including the fused qkv / in_proj splits)."""
import re

import pytest
import torch


def _to_upstream(sd):
    up, fuse = {}, {}
    bb = "model.backbone.conv_encoder.model."

    def put_fused(key, part, t):
        fuse.setdefault(key, {})[part] = t
    for k, v in sd.items():
        if k.startswith(bb + "hidden_states_norms.stage"):
            m = re.match(re.escape(bb) + r"hidden_states_norms\.stage(\d)\.(\w+)", k)
            up[f"backbone.0.norm{int(m.group(1)) - 1}.{m.group(2)}"] = v
        elif k.startswith(bb + "swin."):
            r = k[len(bb + "swin."):]
            m = re.match(r"encoder\.layers\.(\d+)\.blocks\.(\d+)\.attention\.([qkv])_proj\.(weight|bias)", r)
            if m:
                put_fused(f"backbone.0.layers.{m.group(1)}.blocks.{m.group(2)}.attn.qkv.{m.group(4)}", "qkv".index(m.group(3)), v)
                continue
            if "relative_position_index" in r:
                continue
            r = r.replace("embeddings.patch_embeddings.projection", "patch_embed.proj").replace("embeddings.norm", "patch_embed.norm")
            r = r.replace("encoder.layers.", "layers.").replace(".attention.o_proj.", ".attn.proj.")
            r = r.replace(".attention.relative_position_bias.relative_position_bias_table", ".attn.relative_position_bias_table")
            r = r.replace(".layernorm_before.", ".norm1.").replace(".layernorm_after.", ".norm2.")
            up["backbone.0." + r] = v
        elif k.startswith("model.input_proj_vision."):
            up["input_proj." + k[len("model.input_proj_vision."):]] = v
        elif k.startswith("model.text_backbone."):
            up["bert." + k[len("model.text_backbone."):]] = v
        elif k.startswith("model.text_projection."):
            up["feat_map." + k[len("model.text_projection."):]] = v
        elif k == "model.level_embed":
            up["transformer.level_embed"] = v
        elif k.startswith("model.encoder.layers."):
            i, rest = k[len("model.encoder.layers."):].split(".", 1)
            if rest.startswith("deformable_layer."):
                r = rest[len("deformable_layer."):]
                r = r.replace("self_attn_layer_norm.", "norm1.").replace("final_layer_norm.", "norm2.").replace("fc1.", "linear1.").replace("fc2.", "linear2.")
                up[f"transformer.encoder.layers.{i}.{r}"] = v
            elif rest.startswith("text_enhancer_layer."):
                r = rest[len("text_enhancer_layer."):]
                m = re.match(r"self_attn\.(query|key|value)\.(weight|bias)", r)
                if m:
                    put_fused(f"transformer.encoder.text_layers.{i}.self_attn.in_proj_{m.group(2)}", ("query", "key", "value").index(m.group(1)), v)
                    continue
                r = r.replace("layer_norm_before.", "norm1.").replace("layer_norm_after.", "norm2.").replace("fc1.", "linear1.").replace("fc2.", "linear2.")
                up[f"transformer.encoder.text_layers.{i}.{r}"] = v
            elif rest.startswith("fusion_layer."):
                r = rest[len("fusion_layer."):]
                table = {"layer_norm_vision.": "layer_norm_v.", "layer_norm_text.": "layer_norm_l.", "attn.values_vision_proj.": "attn.values_v_proj.",
                         "attn.values_text_proj.": "attn.values_l_proj.", "attn.out_vision_proj.": "attn.out_v_proj.", "attn.out_text_proj.": "attn.out_l_proj.",
                         "attn.vision_proj.": "attn.v_proj.", "attn.text_proj.": "attn.l_proj."}
                for a, b in table.items():
                    if r.startswith(a):
                        r = b + r[len(a):]
                        break
                r = {"vision_param": "gamma_v", "text_param": "gamma_l"}.get(r, r)
                up[f"transformer.encoder.fusion_layers.{i}.{r}"] = v
        elif k.startswith("model.enc_output_norm."):
            up["transformer.enc_output_norm." + k.rsplit(".", 1)[1]] = v
        elif k.startswith("model.enc_output."):
            up["transformer.enc_output." + k.rsplit(".", 1)[1]] = v
        elif k.startswith("model.encoder_output_bbox_embed."):
            up["transformer.enc_out_bbox_embed." + k[len("model.encoder_output_bbox_embed."):]] = v
        elif k == "model.query_position_embeddings.weight":
            up["transformer.tgt_embed.weight"] = v
        elif k.startswith("model.decoder.layers."):
            i, rest = k[len("model.decoder.layers."):].split(".", 1)
            m = re.match(r"(self_attn|encoder_attn_text)\.(query|key|value)\.(weight|bias)", rest)
            if m:
                name = "self_attn" if m.group(1) == "self_attn" else "ca_text"
                put_fused(f"transformer.decoder.layers.{i}.{name}.in_proj_{m.group(3)}", ("query", "key", "value").index(m.group(2)), v)
                continue
            r = rest
            for a, b in (("encoder_attn_text_layer_norm.", "catext_norm."), ("encoder_attn_text.", "ca_text."), ("encoder_attn_layer_norm.", "norm1."),
                         ("encoder_attn.", "cross_attn."), ("self_attn_layer_norm.", "norm2."), ("final_layer_norm.", "norm3."), ("fc1.", "linear1."),
                         ("fc2.", "linear2.")):
                if r.startswith(a):
                    r = b + r[len(a):]
                    break
            up[f"transformer.decoder.layers.{i}.{r}"] = v
        elif k.startswith("model.decoder.layer_norm."):
            up["transformer.decoder.norm." + k.rsplit(".", 1)[1]] = v
        elif k.startswith("model.decoder.reference_points_head."):
            up["transformer.decoder.ref_point_head." + k[len("model.decoder.reference_points_head."):]] = v
        elif k.startswith("bbox_embed."):
            up[k] = v
            up["transformer.decoder." + k] = v                  # upstream lists the shared heads under both names
    for key, parts in fuse.items():
        up[key] = torch.cat([parts[0], parts[1], parts[2]], 0)
    return {"module." + k: v for k, v in up.items()}           # checkpoints saved from DDP carry the prefix


def test_upstream_names_convert_back_to_every_native_key():
    pytest.importorskip("transformers")
    from ovmono3d_amd.gdino.detector import convert_upstream_state_dict
    from test_gpu_gdino import _small_hf_gdino
    hf, _ = _small_hf_gdino()
    sd = {k: v.detach().clone() for k, v in hf.state_dict().items()}
    back = convert_upstream_state_dict(_to_upstream(sd))
    missing = [k for k in sd if k not in back and "relative_position_index" not in k and "position_ids" not in k and not k.startswith("class_embed")
               and not k.startswith("model.encoder_output_class_embed") and not k.startswith("model.decoder.class_embed")
               and not k.startswith("model.decoder.bbox_embed")]
    assert not missing, missing[:10]
    for k, v in back.items():
        assert k in sd, k
        assert torch.equal(v, sd[k]), k


def test_synthetic_gdino_weights_cover_the_hf_ports_parameter_tree():
    """`synthetic://gdino?seed=N` (ovmono3d_amd/util/synth_gdino_weights.py) builds random-init GroundingDINO weights from the
    architecture record alone - the product package imports no model library. Its key set and shapes must be exactly the float
    parameters of the Hugging Face port (the naming ovm_gdino_create consumes), at the full Swin-B / BERT-base / 900-query size."""
    from synth_gdino import synth_gdino_model
    from ovmono3d_amd.util.synth_gdino_weights import synth_gdino_state_dict
    _, ref = synth_gdino_model(0)
    mine = synth_gdino_state_dict(3)
    want = {k: tuple(v.shape) for k, v in ref.items() if v.dtype.is_floating_point}
    got = {k: tuple(v.shape) for k, v in mine.items()}
    assert got == want
    a, b = synth_gdino_state_dict(3), synth_gdino_state_dict(4)
    assert all(torch.equal(mine[k], a[k]) for k in mine) and not torch.equal(a["model.level_embed"], b["model.level_embed"])
    assert mine["bbox_embed.5.layers.2.weight"] is mine["model.decoder.bbox_embed.0.layers.2.weight"]      # shared box MLP under every alias


def test_product_package_does_not_import_checker_libraries():
    """The product package must not import the checker libraries (transformers / the oracle / tests): static scan of its sources."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ovmono3d_amd")
    bad = []
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                for m in re.finditer(r"^\s*(?:from|import)\s+(transformers|oracle|pyref_gdino|synth_gdino|hf_gdino_patches|parity|common)\b", src, re.M):
                    bad.append((os.path.relpath(os.path.join(dp, f), root), m.group(0).strip()))
    assert not bad, bad
