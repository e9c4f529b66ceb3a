import math, sys, os
sys.path.insert(0, os.getcwd())
import torch
from any4_amd import decode_ops as G
from any4_amd.decode import DecodeConfig, _rope_tables
DEV = "cuda:0"
hl, kvl, d, S, bs = 4, 2, 128, 64, 1
cfg = DecodeConfig(head_dim=d, max_seq=S)
cos, sin = _rope_tables(cfg, DEV)
scale = 1.0 / math.sqrt(d)
gen = torch.Generator().manual_seed(1)
p = 5
kc = torch.randn(bs, kvl, S, d, generator=gen).bfloat16()
vc = torch.randn(bs, kvl, S, d, generator=gen).bfloat16()
qkv = torch.randn(bs, (hl + 2 * kvl) * d, generator=gen).bfloat16()
pos = torch.tensor([p], device=DEV)
k2, v2 = kc.clone().to(DEV), vc.clone().to(DEV)
got = G.rope_attn_online(qkv.to(DEV), cos, sin, pos, k2, v2, hl, kvl, d, scale)
torch.cuda.synchronize()
xs = v2[0, kvl - 1, S - 1].view(torch.float32)[: p + 1].cpu()
os.makedirs("gpurun_out", exist_ok=True)
torch.save({"xs": xs, "kc": kc, "vc": vc, "qkv": qkv, "cos": cos.cpu(), "sin": sin.cpu(), "k2": k2.cpu(), "got": got.cpu()}, "gpurun_out/attn_dbg.pt")
print(xs.tolist())
