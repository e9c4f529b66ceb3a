import math, sys, os
sys.path.insert(0, os.getcwd())
import torch
from any4_amd import decode_ops as G
from any4_amd.decode import DecodeConfig, _rope_tables
DEV = "cuda:0"
hl, kvl, d, S, bs = 4, 2, 128, 64, 1
cfg = DecodeConfig(head_dim=d, max_seq=S)
cos, sin = _rope_tables(cfg, DEV)
scale = 1.0
gen = torch.Generator().manual_seed(1)
p = 40
qkv = torch.randn(bs, (hl + 2 * kvl) * d, generator=gen).bfloat16()
res = {}
for run, off in (("A", 0), ("B", 32), ("C", 64), ("D", 96)):
    kc = torch.zeros(bs, kvl, S, d).bfloat16()
    for r in range(32):
        kc[:, :, r, off + r] = 1.0
    vc = torch.zeros(bs, kvl, S, d).bfloat16()
    pos = torch.tensor([p], device=DEV)
    k2, v2 = kc.clone().to(DEV), vc.clone().to(DEV)
    got = G.rope_attn_online(qkv.to(DEV), cos, sin, pos, k2, v2, hl, kvl, d, scale)
    torch.cuda.synchronize()
    res[run] = v2[0, kvl - 1, S - 1].view(torch.float32)[:32].cpu()
torch.save({"A": res["A"], "B": res["B"], "C": res["C"], "D": res["D"], "qkv": qkv, "cos": cos.cpu(), "sin": sin.cpu()}, "gpurun_out/attn_dbg5.pt")
