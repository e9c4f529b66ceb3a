import math, sys, os
sys.path.insert(0, os.getcwd())
import torch
from any4_amd import decode_ops as G
from any4_amd.decode import DecodeConfig, _rope_tables
DEV = "cuda:0"
gen = torch.Generator(device=DEV).manual_seed(1)
for (hl, kvl, d) in ((4, 2, 128), (4, 2, 64)):
    bs, S = 1, 64
    cfg = DecodeConfig(head_dim=d, max_seq=S)
    cos, sin = _rope_tables(cfg, DEV)
    scale = 1.0 / math.sqrt(d)
    for p in (0, 1, 5, 40):
        kc = torch.randn(bs, kvl, S, d, device=DEV, generator=gen).bfloat16()
        vc = torch.randn(bs, kvl, S, d, device=DEV, generator=gen).bfloat16()
        pos = torch.tensor([p], device=DEV)
        qkv = torch.randn(bs, (hl + 2 * kvl) * d, device=DEV, generator=gen).bfloat16()
        k1, v1, k2, v2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
        want = G.rope_attn(qkv, cos, sin, pos, k1, v1, hl, kvl, d, scale)
        got = G.rope_attn_online(qkv, cos, sin, pos, k2, v2, hl, kvl, d, scale)
        print(d, p, "k equal", torch.equal(k1, k2), "v equal", torch.equal(v1, v2), "max diff", (got.float() - want.float()).abs().max().item(), "max", want.float().abs().max().item())
        if not torch.equal(k1, k2):
            dd = (k1.float() - k2.float()).abs()
            print("   k diff at", dd.nonzero()[:5].tolist(), k1[0, 0, p, :4].tolist(), k2[0, 0, p, :4].tolist())
