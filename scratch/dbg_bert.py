import sys, torch, torch.nn.functional as F
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from transformers import BertConfig, BertModel
from ovmono3d_amd.gdino.ops import Ops
from ovmono3d_amd.gdino.bert import BertEncoder, masks_and_position_ids
torch.manual_seed(0)
cfg = BertConfig(vocab_size=2000, hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072, max_position_embeddings=512, attn_implementation="eager")
hf = BertModel(cfg, add_pooling_layer=False).eval()
with torch.no_grad():
    for p_ in hf.parameters(): p_.mul_(3.0)
ids = torch.tensor([101, 500, 1012, 600, 601, 1012, 700, 701, 702, 1012, 102]); T=len(ids)
mask,pos=masks_and_position_ids(ids)
sd=hf.state_dict(); dev=torch.device('cuda'); o=Ops(dev)
enc=BertEncoder(o, {"model.text_backbone."+k:v for k,v in sd.items()})
def rel(a,b): return float((a.cpu()-b).abs().max()/b.abs().max())
x = sd["embeddings.word_embeddings.weight"][ids] + sd["embeddings.position_embeddings.weight"][pos] + sd["embeddings.token_type_embeddings.weight"][0]
xr = F.layer_norm(x,(768,),sd["embeddings.LayerNorm.weight"],sd["embeddings.LayerNorm.bias"],1e-12)
i32=lambda t: t.to(torch.int32).view(T,1)
xe=o.gather_rows(enc.word,i32(ids)); print("word", rel(xe, sd["embeddings.word_embeddings.weight"][ids]))
xe=o.add(xe,o.gather_rows(enc.pos,i32(pos))); xe=o.add(xe,o.gather_rows(enc.typ,i32(torch.zeros_like(ids))))
print("emb sum", rel(xe,x))
xn=o.layernorm(xe,enc.eg,enc.eb,1e-12); print("emb ln", rel(xn,xr))
ly=enc.layers[0]; pfx="encoder.layer.0."
qkv=o.linear(xn,ly["qkv"])
qr=F.linear(xr,sd[pfx+"attention.self.query.weight"],sd[pfx+"attention.self.query.bias"]); kr=F.linear(xr,sd[pfx+"attention.self.key.weight"],sd[pfx+"attention.self.key.bias"]); vr=F.linear(xr,sd[pfx+"attention.self.value.weight"],sd[pfx+"attention.self.value.bias"])
print("qkv", rel(qkv, torch.cat([qr,kr,vr],1)))
H,D,dh=12,768,64
s=o.empty(H,T,T); o.bmm_raw(qkv,0,qkv,D,s,0,H,T,T,dh,3*D,3*D,T,dh,dh,T*T,True,dh**-0.5)
sr=(qr.view(T,H,dh).transpose(0,1)@kr.view(T,H,dh).transpose(0,1).transpose(1,2))/8
print("scores", rel(s,sr))
bias=torch.where(mask,0.0,torch.finfo(torch.float32).min).to(dev).contiguous()
o.softmax_(s,bias,bias_rows=T,bias_div=1); pr=(sr+bias.cpu()).softmax(-1); print("probs", rel(s,pr))
ctx=o.empty(T,D); o.bmm_raw(s,0,qkv,2*D,ctx,0,H,T,dh,T,T,3*D,D,T*T,dh,dh,False,1.0)
cr=(pr@vr.view(T,H,dh).transpose(0,1)).transpose(0,1).reshape(T,D); print("ctx", rel(ctx,cr))
