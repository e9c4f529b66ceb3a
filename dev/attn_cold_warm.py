"""Developer probe: what does the decode step's attention node cost with its K / V rows cold (HBM) and warm (the XCD's L2)?
A hipGraph of 32 rope_attn_online launches over 32 distinct caches (Llama-3-8B: 32 / 8 heads, d = 128, position 160), replayed
as is (rows stay in L2 / Infinity Cache between replays) and with a 600 MB read-modify-write between the launches (everything evicted)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from any4_amd import decode_ops as D

dev = "cuda:0"
L, hl, kvl, d, max_seq, pos0 = 32, 32, 8, 128, 1024, 160
g = torch.Generator(device=dev).manual_seed(0)
qkv = [torch.randn(1, (hl + 2 * kvl) * d, device=dev, generator=g).bfloat16() for _ in range(L)]
kc = [torch.randn(1, kvl, max_seq, d, device=dev, generator=g).bfloat16() for _ in range(L)]
vc = [torch.randn(1, kvl, max_seq, d, device=dev, generator=g).bfloat16() for _ in range(L)]
inv = 1.0 / (500000.0 ** (torch.arange(0, d, 2, device=dev).float() / d))
ang = torch.arange(max_seq, device=dev).float()[:, None] * inv[None, :]
cos, sin = torch.cat([ang.cos(), ang.cos()], 1).contiguous(), torch.cat([ang.sin(), ang.sin()], 1).contiguous()
pos = torch.full((1,), pos0, dtype=torch.int64, device=dev)
big = torch.zeros(150 * 1024 * 1024, dtype=torch.float32, device=dev)


def graph(attn, flush):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(L):
            if flush:
                big.add_(1.0)
            if attn:
                D.rope_attn_online(qkv[i], cos, sin, pos, kc[i], vc[i], hl, kvl, d, 1.0 / d ** 0.5)
    return gr


def time(gr, reps=10):
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / L


D.rope_attn_online(qkv[0], cos, sin, pos, kc[0], vc[0], hl, kvl, d, 1.0 / d ** 0.5)
torch.cuda.synchronize()
warm = time(graph(True, False))
both = time(graph(True, True), 3)
flush = time(graph(False, True), 3)
print(f"attention node, rows warm (L2 / Infinity Cache): {warm:.2f} us;  rows cold: {both - flush:.2f} us  (flush + attention {both:.2f}, flush alone {flush:.2f})")
