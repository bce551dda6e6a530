import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from test_gpu_gdino import _small_hf_gdino
from transformers.models.grounding_dino.modeling_grounding_dino import generate_masks_with_special_tokens_and_transfer_map
from ovmono3d_amd.gdino.model import GDinoConfig, GroundingDinoNative
from ovmono3d_amd.gdino.ops import Ops
hf,cfg=_small_hf_gdino()
H,W=96,132
g=torch.Generator().manual_seed(2); img=torch.randn(1,3,H,W,generator=g)
ids=torch.tensor([101, 500, 1012, 600, 601, 1012, 700, 701, 702, 1012, 102])
cap={}
def hook(name):
    def f(m,i,o): cap[name]=(i,o)
    return f
for l in range(4): hf.model.input_proj_vision[l].register_forward_hook(hook(f"ip{l}"))
hf.model.text_projection.register_forward_hook(hook("tp"))
e0=hf.model.encoder.layers[0]
e0.fusion_layer.register_forward_hook(hook("fus")); e0.text_enhancer_layer.register_forward_hook(hook("te")); e0.deformable_layer.register_forward_hook(hook("de"))
e0.deformable_layer.self_attn.register_forward_hook(hook("msda"))
with torch.no_grad(): out=hf(pixel_values=img,input_ids=ids[None],return_dict=True)
_,p_hf=generate_masks_with_special_tokens_and_transfer_map(ids[None])
ncfg=GDinoConfig(d_model=64,enc_layers=2,dec_layers=2,heads=4,ffn_dim=128,num_queries=30,bert_heads=2,swin_embed=32,swin_depths=(2,2,2,2),swin_heads=(1,2,4,8),swin_window=12)
dev=torch.device('cuda'); o=Ops(dev)
net=GroundingDinoNative(o,hf.state_dict(),ncfg)
# replay pieces with native ops and compare
def rel(a,b): return float((a.cpu()-b).abs().max()/b.abs().max())
x=img[0].permute(1,2,0).reshape(H*W,3).contiguous().to(dev)
lg,bx,aux=net.forward(x,H,W,ids,position_ids=p_hf[0],return_aux=True)
print("text_features", rel(aux["text_features"], cap["tp"][1][0]))
srcs=torch.cat([cap[f"ip{l}"][1][0].flatten(1).T for l in range(4)],0)
print("source_flatten", rel(aux["source_flatten"], srcs), [tuple(cap[f"ip{l}"][1].shape) for l in range(4)])
# encoder layer 0 pieces: run native layer 0 manually
import ovmono3d_amd.gdino.ops as O
c=ncfg; ly=net.enc[0]
vis=aux["source_flatten"]; text=aux["text_features"]
v=o.layernorm(vis,ly["lnv"][0],ly["lnv"][1],c.eps); t=o.layernorm(text,ly["lnt"][0],ly["lnt"][1],c.eps)
(fv,_),(ft,_)=cap["fus"][1]
S=vis.shape[0]; T=len(ids); HF_,dhf,E=2,32,64
q,k=o.linear(v,ly["vq"]),o.linear(t,ly["tk"]); vv,tv=o.linear(v,ly["vv"]),o.linear(t,ly["tv"])
a_v=o.empty(HF_,S,T); o.bmm_raw(q,0,k,0,a_v,0,HF_,S,T,dhf,E,E,T,dhf,dhf,S*T,True,dhf**-0.5)
a_t=o.empty(HF_,T,S); o.bmm_raw(k,0,q,0,a_t,0,HF_,T,S,dhf,E,E,S,dhf,dhf,T*S,True,dhf**-0.5)
o.softmax_(a_v); o.softmax_(a_t)
cv,ct=o.empty(S,E),o.empty(T,E)
o.bmm_raw(a_v,0,tv,0,cv,0,HF_,S,dhf,T,T,E,E,S*T,dhf,dhf,False,1.0); o.bmm_raw(a_t,0,vv,0,ct,0,HF_,T,dhf,S,S,E,E,T*S,dhf,dhf,False,1.0)
vis1=o.linear(cv,ly["ov"],residual=v); text1=o.linear(ct,ly["ot"],residual=t)
print("fusion vision", rel(vis1,fv[0]), "fusion text", rel(text1,ft[0]))
te_in=cap["te"][0]; print("te kwargs?", len(te_in))
print("te out", "hf", tuple(cap["te"][1][0].shape))
print("de out", tuple(cap["de"][1][0].shape))
# text enhancer
from ovmono3d_amd.gdino.bert import masks_and_position_ids
mask,_=masks_and_position_ids(ids)
text_bias=torch.where(mask,0.0,torch.finfo(torch.float32).min).to(dev).contiguous()
text_pos=o.sine_embed(p_hf[0].float().view(T,1).to(dev),64,10000.0)
from transformers.models.grounding_dino.modeling_grounding_dino import encode_sinusoidal_position_embedding
print("text_pos", rel(text_pos, encode_sinusoidal_position_embedding(p_hf[0][...,None].float(), num_pos_feats=64)))
qk=o.add(text1,text_pos)
t2=o.layernorm(ly["te_attn"](qk,qk,text1,bias=text_bias,residual=text1),ly["te_ln1"][0],ly["te_ln1"][1],c.eps)
ff=o.linear(o.linear(t2,ly["te_fc1"],act=1),ly["te_fc2"],residual=t2)
t3=o.layernorm(ff,ly["te_ln2"][0],ly["te_ln2"][1],c.eps)
print("text enhancer", rel(t3,cap["te"][1][0][0]))
shapes=[(12,17),(6,9),(3,5),(2,3)]
tb=net._shape_tables(shapes)
qd=o.add(vis1,tb["pos"])
loc_fn=lambda off: o.add(o.elt(O.MUL,off,tb["inv_norm"]),tb["ref_exp"])
m=ly["msda"](qd,vis1,shapes,loc_fn,residual=None)
print("msda out", rel(m, cap["msda"][1][0][0]))
vis2=o.layernorm(o.add(m,vis1),ly["de_ln1"][0],ly["de_ln1"][1],c.eps)
ff=o.linear(o.linear(vis2,ly["de_fc1"],act=1),ly["de_fc2"],residual=vis2)
vis3=o.layernorm(ff,ly["de_ln2"][0],ly["de_ln2"][1],c.eps)
print("deformable layer", rel(vis3,cap["de"][1][0][0]))
print("---- text enhancer internals")
import math, torch.nn.functional as F
te=hf.model.encoder.layers[0].text_enhancer_layer; sd2=te.state_dict()
x_c=text1.cpu(); pe_c=text_pos.cpu(); qk_c=x_c+pe_c
att=ly["te_attn"]
qn=o.linear(qk,att.q); kn=o.linear(qk,att.k); vn=o.linear(text1,att.v)
qr=F.linear(qk_c,sd2["self_attn.query.weight"],sd2["self_attn.query.bias"]); kr=F.linear(qk_c,sd2["self_attn.key.weight"],sd2["self_attn.key.bias"]); vr=F.linear(x_c,sd2["self_attn.value.weight"],sd2["self_attn.value.bias"])
print("q",rel(qn,qr),"k",rel(kn,kr),"v",rel(vn,vr))
Hh,dh,Dm=2,32,64
s=o.empty(Hh,T,T); o.bmm_raw(qn,0,kn,0,s,0,Hh,T,T,dh,Dm,Dm,T,dh,dh,T*T,True,1/math.sqrt(dh))
sr=(qr.view(T,Hh,dh).transpose(0,1)@kr.view(T,Hh,dh).transpose(0,1).transpose(1,2))/math.sqrt(dh)
print("scores",rel(s,sr))
o.softmax_(s,text_bias,bias_rows=T,bias_div=1); pr=(sr+text_bias.cpu()).softmax(-1); print("probs",rel(s,pr))
ctx=o.empty(T,Dm); o.bmm_raw(s,0,vn,0,ctx,0,Hh,T,dh,T,T,Dm,Dm,T*T,dh,dh,False,1.0)
cr=(pr@vr.view(T,Hh,dh).transpose(0,1)).transpose(0,1).reshape(T,Dm); print("ctx",rel(ctx,cr))
ao=o.linear(ctx,att.out,residual=text1); ar=F.linear(cr,sd2["self_attn.out_proj.weight"],sd2["self_attn.out_proj.bias"])+x_c; print("out+res",rel(ao,ar))
h1=o.layernorm(ao,ly["te_ln1"][0],ly["te_ln1"][1],c.eps); h1r=F.layer_norm(ar,(64,),sd2["layer_norm_before.weight"],sd2["layer_norm_before.bias"],1e-5); print("ln1",rel(h1,h1r))
f1=o.linear(h1,ly["te_fc1"],act=1); f1r=F.relu(F.linear(h1r,sd2["fc1.weight"],sd2["fc1.bias"])); print("fc1",rel(f1,f1r), tuple(f1.shape))
f2=o.linear(f1,ly["te_fc2"],residual=h1); f2r=F.linear(f1r,sd2["fc2.weight"],sd2["fc2.bias"])+h1r; print("fc2+res",rel(f2,f2r))
h2=o.layernorm(f2,ly["te_ln2"][0],ly["te_ln2"][1],c.eps); h2r=F.layer_norm(f2r,(64,),sd2["layer_norm_after.weight"],sd2["layer_norm_after.bias"],1e-5); print("ln2",rel(h2,h2r), "vs hook", rel(h2, cap["te"][1][0][0]), rel(h2r, cap["te"][1][0][0]))
print("t3 vs h2", rel(t3,h2r), "t2 vs h1", rel(t2,h1r))
