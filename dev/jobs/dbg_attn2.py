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
for p in (1, 2, 5, 33):
    kc = torch.zeros(bs, kvl, S, d, device=DEV).bfloat16()
    vc = torch.zeros(bs, kvl, S, d, device=DEV).bfloat16()
    for s in range(S):
        vc[:, :, s, :] = s + 1
    qkv = torch.zeros(bs, (hl + 2 * kvl) * d, device=DEV).bfloat16()
    qkv[:, (hl + kvl) * d:] = p + 1   # v of the new token
    pos = torch.tensor([p], device=DEV)
    k2, v2 = kc.clone(), vc.clone()
    got = G.rope_attn_online(qkv, cos, sin, pos, k2, v2, hl, kvl, d, scale)
    print("uniform scores: p", p, "expect", (p + 2) / 2, "got", got[0, :4].tolist(), got[0, 128:132].tolist())
    # one-hot: q = e0 * 8, K[s] = e0 * s  -> scores s * 8 * scale
    kc2 = kc.clone(); kc2[:, :, :, 0] = torch.arange(S, device=DEV).bfloat16() * 0.25
    qkv2 = qkv.clone(); qkv2[:, 0::d][:, :hl] = 4.0
    k1, v1, k2, v2 = kc2.clone(), vc.clone(), kc2.clone(), vc.clone()
    want = G.rope_attn(qkv2, cos, sin, pos, k1, v1, hl, kvl, d, scale)
    got = G.rope_attn_online(qkv2, cos, sin, pos, k2, v2, hl, kvl, d, scale)
    print("   ramp scores: want", want[0, :3].tolist(), "got", got[0, :3].tolist())
