import math, sys, os
sys.path.insert(0, os.getcwd())
import torch
from any4_amd import decode_ops as G
from any4_amd.decode import DecodeConfig, _rope_tables, _rope
DEV = "cuda:0"
hl, kvl, d, S, bs = 4, 2, 128, 64, 1
cfg = DecodeConfig(head_dim=d, max_seq=S)
cos, sin = _rope_tables(cfg, DEV)
scale = 1.0 / math.sqrt(d)
gen = torch.Generator(device=DEV).manual_seed(1)
p = 5
kc = torch.randn(bs, kvl, S, d, device=DEV, generator=gen).bfloat16()
vc = torch.randn(bs, kvl, S, d, device=DEV, generator=gen).bfloat16()
qkv = torch.randn(bs, (hl + 2 * kvl) * d, device=DEV, generator=gen).bfloat16()
pos = torch.tensor([p], device=DEV)
k2, v2 = kc.clone(), vc.clone()
got = G.rope_attn_online(qkv, cos, sin, pos, k2, v2, hl, kvl, d, scale)
torch.cuda.synchronize()
xs = v2[0, kvl - 1, S - 1].view(torch.float32)[: p + 1]
c, s_ = cos[p].view(1, 1, -1), sin[p].view(1, 1, -1)
q = _rope(qkv[:, : hl * d].reshape(bs, hl, d), c, s_)
sc = (q[0, 0].float() @ k2[0, 0, : p + 1].float().t())
print("kernel scores", xs.tolist())
print("expected     ", (sc.bfloat16().float() * scale).tolist())
import itertools
qraw = qkv[:, : hl * d].reshape(bs, hl, d)
for name, qq in (("rot", q), ("raw", qraw)):
    for hh in range(hl):
        for kvh in range(kvl):
            for kn_, kt in (("after", k2), ("before", kc)):
                sc = (qq[0, hh].float() @ kt[0, kvh, : p + 1].float().t()).bfloat16().float() * scale
                print(name, "q head", hh, "kv", kvh, kn_, [round(v, 4) for v in sc.tolist()])
