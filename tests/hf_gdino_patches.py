"""Test-side patches that make the Hugging Face GroundingDINO port (transformers 5.x) compute what upstream
IDEA-Research/GroundingDINO @856dde2 computes, in the three places where the port departs from it. Used by the
GPU parity tests and by bench.py's cpu_baseline / parity leg; never imported by the product package."""
from __future__ import annotations

import contextlib

import torch


def patch_hf_to_upstream(hf):
    """(1) transformers 5.x's BertModel adds a 4-D bool mask as +1.0 (nothing is masked); upstream builds the additive mask
    (get_extended_attention_mask): feed HF what upstream computes. (3) encode_sinusoidal_position_embedding casts its result
    back to the input dtype, which truncates the text position embedding of int64 position ids to integers; upstream
    (get_sine_pos_embed) keeps floats."""
    tb = hf.model.text_backbone
    if not getattr(tb, "_ovm_patched", False):
        orig = tb.forward

        def patched(input_ids, attention_mask=None, token_type_ids=None, position_ids=None, **kw):
            if attention_mask is not None and attention_mask.dtype == torch.bool:
                attention_mask = torch.where(attention_mask, 0.0, torch.finfo(torch.float32).min)
            return orig(input_ids, attention_mask, token_type_ids, position_ids, **kw)
        tb.forward = patched
        tb._ovm_patched = True
    import transformers.models.grounding_dino.modeling_grounding_dino as mgd
    if not getattr(mgd, "_ovm_patched", False):
        orig_enc = mgd.encode_sinusoidal_position_embedding
        mgd.encode_sinusoidal_position_embedding = lambda pos, **kw: orig_enc(pos.float(), **kw)
        mgd._ovm_patched = True
    return hf


@contextlib.contextmanager
def upstream_position_ids():
    """(2) HF numbers the text positions its own way (the '.' delimiters get position 0); upstream - which the native path
    follows - numbers them 0..len inside each phrase, delimiter included. Inside this context HF uses upstream's ids."""
    import transformers.models.grounding_dino.modeling_grounding_dino as mgd
    from pyref_gdino.bert import masks_and_position_ids
    orig = mgd.generate_masks_with_special_tokens_and_transfer_map
    mgd.generate_masks_with_special_tokens_and_transfer_map = \
        lambda ids: (orig(ids)[0], masks_and_position_ids(ids[0].cpu())[1][None].to(ids.device))
    try:
        yield
    finally:
        mgd.generate_masks_with_special_tokens_and_transfer_map = orig
