import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from common import rel_err
import test_gpu_gdino as T
from transformers.models.grounding_dino.modeling_grounding_dino import generate_masks_with_special_tokens_and_transfer_map
from ovmono3d_amd.gdino.detector import HashTokenizer
from ovmono3d_amd.gdino.model import GDinoConfig, GroundingDinoNative
from ovmono3d_amd.util.synth_gdino import synth_gdino_model
dev = torch.device("cuda:0")
hf, sd = synth_gdino_model(5); T._patch_hf_to_upstream(hf); hf = hf.to(dev)
H, W = 532, 708
img = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(2)).to(dev)
ids = torch.tensor(HashTokenizer().encode("chair . dining table . sofa . potted plant . television . bookcase ."))
with torch.no_grad():
    out = hf(pixel_values=img, input_ids=ids[None].to(dev), return_dict=True, output_hidden_states=True)
_, p_hf = generate_masks_with_special_tokens_and_transfer_map(ids[None])
net = GroundingDinoNative(T._ops(dev), sd, GDinoConfig())
x = img[0].permute(1, 2, 0).reshape(H * W, 3).contiguous()
logits, boxes, aux = net.forward(x, H, W, ids, position_ids=p_hf[0], return_aux=True)
theirs = torch.topk(out.enc_outputs_class[0].max(-1)[0], 900)[1]
logits, boxes, aux = net.forward(x, H, W, ids, position_ids=p_hf[0], return_aux=True, force_topk=theirs)
mine = aux["topk"].cpu(); theirs = torch.topk(out.enc_outputs_class[0].max(-1)[0], 900)[1].cpu()
print("order mismatches", int((mine != theirs).sum()), "set equal", sorted(mine.tolist()) == sorted(theirs.tolist()))
print("enc vis", rel_err(aux["enc_vision"], out.encoder_last_hidden_state_vision[0]), "enc text", rel_err(aux["enc_text"], out.encoder_last_hidden_state_text[0]))
print("init_ref", rel_err(aux["init_ref"], out.init_reference_points[0]))
ihs, iref = out.intermediate_hidden_states[0], out.intermediate_reference_points[0]
print("shapes", ihs.shape, iref.shape)
ln = hf.model.decoder.layer_norm
for i in range(6):
    mh = torch.nn.functional.layer_norm(aux["dec_hs"][i], (256,), ln.weight, ln.bias, 1e-5)
    e = (mh - ihs[i]).abs().amax(-1)
    print(i, "hs", rel_err(mh, ihs[i]), "rows>1e-3:", int((e > 1e-3 * ihs[i].abs().max()).sum()), "ref_in", rel_err(aux["dec_ref"][i], iref[i]))
print("boxes", rel_err(boxes, out.pred_boxes[0]), "logits", rel_err(logits[:, :len(ids)], out.logits[0][:, :len(ids)]))
