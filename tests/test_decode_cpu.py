"""CPU suite, part 3: the decode harness's plumbing (any4_amd/decode.py) -- fused-linear row bookkeeping, KV cache,
and tensor-parallel row-sharding with all-gathers over gloo (world_size 2) against the unsharded stack.  The
linears here are a tests-only dense stand-in (the product's linears need the GPU); no HIP compute."""
import os

import torch

from any4_amd.decode import DecodeConfig, DecodeStack, shard_rows

CFG = dict(hidden=64, inter=96, layers=2, heads=4, kv_heads=2, head_dim=16, vocab=50, max_seq=16, group_size=32)


class SeededDense:
    """Deterministic full weight per (name, layer); returns the rows `shard_rows` assigns to this rank."""

    def __init__(self, cfg, rank, world):
        self.cfg, self.rank, self.world = cfg, rank, world

    def __call__(self, name, layer, in_features, rows):
        full_rows = {n: o for n, o, _ in self.cfg.linear_shapes()}[name]
        gen = torch.Generator().manual_seed(1000 * layer + sum(map(ord, name)))
        w = torch.randn(full_rows, in_features, generator=gen) / in_features ** 0.5
        idx = shard_rows(self.cfg, name, self.rank, self.world)
        assert idx.numel() == rows
        lin = torch.nn.Linear(in_features, rows, bias=False)
        lin.weight.data = w[idx].contiguous()
        return lin


def _run(cfg, rank, world, tokens_seq):
    stack = DecodeStack(cfg, SeededDense(cfg, rank, world), "cpu", torch.float32, bs=tokens_seq.shape[1], rank=rank,
                        world=world, seed=7)
    return torch.stack([stack.decode(t, i).clone() for i, t in enumerate(tokens_seq)])


def test_shard_rows_partition_the_fused_weights():
    cfg = DecodeConfig(**CFG)
    for name, n, _ in cfg.linear_shapes():
        for world in (1, 2):
            got = torch.cat([shard_rows(cfg, name, r, world) for r in range(world)])
            assert sorted(got.tolist()) == list(range(n)), (name, world)
    # the local qkv of rank 1 of 2 starts with the second half of the q heads
    assert shard_rows(cfg, "qkv", 1, 2)[0].item() == cfg.heads * cfg.head_dim // 2


def _full_sequence_logits(stack, toks):
    """Independent restatement: the whole prefix at once, causal mask, no cache (batch of 1 sequence)."""
    import math

    cfg, T = stack.cfg, toks.shape[0]
    d, H, KV = cfg.head_dim, cfg.heads, cfg.kv_heads
    h = stack.embed(toks)                                                        # [T, hidden]
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    ang = torch.arange(T, dtype=torch.float32)[:, None] * inv[None, :]
    cos, sin = torch.cat([ang.cos()] * 2, -1)[:, None, :], torch.cat([ang.sin()] * 2, -1)[:, None, :]

    def rope(x):
        return x * cos + torch.cat([-x[..., d // 2:], x[..., : d // 2]], -1) * sin

    def rms(x, w):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + cfg.rms_eps) * w

    causal = torch.triu(torch.ones(T, T, dtype=torch.bool), 1)
    for layer in stack.layers:
        qkv = layer.qkv(rms(h, layer.norm1.weight))
        q = rope(qkv[:, : H * d].view(T, H, d))
        k = rope(qkv[:, H * d: (H + KV) * d].view(T, KV, d)).repeat_interleave(H // KV, dim=1)
        v = qkv[:, (H + KV) * d:].view(T, KV, d).repeat_interleave(H // KV, dim=1)
        att = torch.einsum("thd,shd->hts", q, k) / math.sqrt(d)
        att = att.masked_fill(causal, float("-inf")).softmax(-1)
        h = h + layer.o(torch.einsum("hts,shd->thd", att, v).reshape(T, H * d))
        gu = layer.gate_up(rms(h, layer.norm2.weight))
        h = h + layer.down(torch.nn.functional.silu(gu[:, : cfg.inter]) * gu[:, cfg.inter:])
    return stack.lm_head(rms(h, stack.norm.weight))


def test_decode_matches_a_full_sequence_reference():
    """Token-by-token decode over the static KV cache == causal attention over the whole prefix."""
    cfg = DecodeConfig(**CFG)
    torch.manual_seed(0)
    toks = torch.randint(0, cfg.vocab, (6, 1))
    stack = DecodeStack(cfg, SeededDense(cfg, 0, 1), "cpu", torch.float32, bs=1, seed=7)
    with torch.no_grad():
        step = torch.stack([stack.decode(t, i).clone() for i, t in enumerate(toks)])[:, 0]
        full = _full_sequence_logits(stack, toks[:, 0])
    assert torch.allclose(step, full, atol=1e-4), (step - full).abs().max()
    # replaying the same tokens on a fresh stack gives the same logits (no stale cache state)
    assert torch.equal(_run(cfg, 0, 1, toks), _run(cfg, 0, 1, toks))
    # a position outside the KV cache is refused on the host (the fused kernels read it from device memory)
    import pytest

    for bad in (-1, cfg.max_seq, cfg.max_seq + 5):
        with pytest.raises(ValueError, match="outside the KV cache"):
            stack.decode(toks[0], bad)


def _tp_worker(rank, world, port, results):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = DecodeConfig(**CFG)
        toks = torch.randint(0, cfg.vocab, (4, 3), generator=torch.Generator().manual_seed(3))
        full = _run(cfg, 0, 1, toks)        # unsharded, no collectives
        tp = _run(cfg, rank, world, toks)   # heads / rows split over the two ranks
        results[rank] = float((tp - full).abs().max())
    finally:
        dist.destroy_process_group()


def test_tensor_parallel_decode_gloo():
    import torch.multiprocessing as mp

    world = 2
    port = 31500 + (os.getpid() % 2000)
    results = mp.Manager().dict()
    mp.spawn(_tp_worker, args=(world, port, results), nprocs=world, join=True)
    assert set(results.keys()) == {0, 1}
    assert max(results.values()) < 1e-4, dict(results)
