"""Batch-1 decode harness around the quantized linears (BASELINE config 5).

The reference measures model-level speed with `benchmark.py:113-215`: a HuggingFace causal LM, every
`nn.Linear` except the LM head swapped for the quantized module by `quantize_model` (quantize.py:32-85),
forward on random `input_ids[bs, seqlen]`.  Neither checkpoints nor the hub are reachable here, so this
module builds the same *shape* of work from scratch -- a Llama-architecture decoder (RMSNorm, rotary
embeddings, grouped-query attention over a static KV cache, SwiGLU MLP) with random-initialised weights of
the named configuration -- and runs ONE decode step (one new token per sequence) through it.  Everything
except the linears is plain torch (plumbing); the linears are whatever the `linear_factory` returns:
`Any4Factory` (the product: `Any4Linear` on the HIP kernels) or `DenseFactory` (bf16 `nn.Linear`, the
baseline the reference's README quotes speedups against).

Launch structure, chosen for the hardware rather than copied from HF:
  * q/k/v are ONE linear (rows concatenated: weight rows are independent units of the tinygemm path) and
    gate/up likewise -> 4 GEMM launches per layer instead of 7;
  * the whole step is captured in a hipGraph (`DecodeStack.capture`): at batch 1 a layer's GEMMs take a few
    microseconds each, so launch gaps would otherwise dominate;
  * tensor parallelism = row-sharding of every linear (any4_amd/shard.py): heads are split across ranks so
    attention and the KV cache stay local; per layer 4 all-gathers of [bs, n/G] partial outputs (attention
    output, o_proj, SwiGLU activation, down_proj) over RCCL.  One process per GPU.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Callable, Optional

import torch
import torch.distributed as dist


@dataclass
class DecodeConfig:
    hidden: int = 4096
    inter: int = 14336
    layers: int = 32
    heads: int = 32
    kv_heads: int = 8
    head_dim: int = 128
    vocab: int = 128256
    max_seq: int = 2048
    rope_theta: float = 500000.0
    rms_eps: float = 1e-5
    group_size: int = 128
    # row order of the fused gate_up weight: 0 = [gate rows; up rows]; 8 = blocks of 8 gate rows followed by the 8 matching up
    # rows, the order the GEMM's fused SwiGLU epilogue wants (tg_w4_gemm.epilogue): a gate row and its up row then meet in one
    # 16-row tile of one workgroup.  `shard_rows` returns the row indices in this order, so a loader needs nothing else.
    gate_up_interleave: int = 0

    @classmethod
    def llama3_8b(cls, **kw) -> "DecodeConfig":
        return cls(**kw)

    @classmethod
    def llama2_7b(cls, **kw) -> "DecodeConfig":
        # the reference's default benchmark model (benchmark.py: --model-name meta-llama/Llama-2-7b-hf)
        return cls(hidden=4096, inter=11008, layers=32, heads=32, kv_heads=32, head_dim=128, vocab=32000,
                   rope_theta=10000.0, **kw)

    def linear_shapes(self):
        """(name, out_features, in_features) of the four fused linears of one layer."""
        qkv = (self.heads + 2 * self.kv_heads) * self.head_dim
        return [("qkv", qkv, self.hidden), ("o", self.hidden, self.heads * self.head_dim),
                ("gate_up", 2 * self.inter, self.hidden), ("down", self.hidden, self.inter)]

    def weight_bytes_4bit(self) -> int:
        """Algorithmic bytes one decode step streams through the quantized linears (bench.py formula, m=1)."""
        total = 0
        for _, n, k in self.linear_shapes():
            total += n * k // 2 + (k // self.group_size) * n * 4 + 32 * n + k * 2 + n * 2
        return total * self.layers


def shard_rows(cfg: DecodeConfig, name: str, rank: int, world: int) -> torch.Tensor:
    """Row indices of the FULL fused weight `name` ("qkv" = [q; k; v] rows, "gate_up" = [gate; up] rows, "o",
    "down") that rank `rank` of `world` owns, in the order of its local weight.  Heads are split across ranks,
    so the local qkv is [q heads of the rank; k heads of the rank; v heads of the rank] and the local gate_up is
    [gate rows of the rank; up rows of the rank].  A checkpoint loader slices codes / LUT rows /
    scales_and_zeros[:, rows, :] with exactly these indices."""
    d = cfg.head_dim

    def span(base, total):
        per = total // world
        return torch.arange(base + rank * per, base + (rank + 1) * per)

    if name == "qkv":
        return torch.cat([span(0, cfg.heads * d), span(cfg.heads * d, cfg.kv_heads * d),
                          span((cfg.heads + cfg.kv_heads) * d, cfg.kv_heads * d)])
    if name == "gate_up":
        gate, up = span(0, cfg.inter), span(cfg.inter, cfg.inter)
        if cfg.gate_up_interleave:
            b = cfg.gate_up_interleave
            return torch.stack([gate.view(-1, b), up.view(-1, b)], dim=1).reshape(-1)
        return torch.cat([gate, up])
    if name in ("o", "down"):
        return span(0, cfg.hidden)
    raise ValueError(name)


# ---------------------------------------------------------------------------------------------------
# linear factories: (name, layer index, in_features, row ranges of the FULL weight owned by this rank) -> module
# ---------------------------------------------------------------------------------------------------

class Any4Factory:
    """Random any4 weights straight in the packed layout (no k-means, no packing pass): uniformly random
    nibbles are what packing uniformly random codes gives.  Per-row LUT, per-group scale/zero."""

    def __init__(self, cfg: DecodeConfig, device, dtype=torch.bfloat16, seed: int = 0, w_inner_k: int = 4,
                 kernel: str = "linear_y_f16RM_x_f16RM_W_any4TC"):
        from .modules import Any4Linear  # imports the HIP library; fails loudly without it

        self._cls = Any4Linear
        self.cfg, self.device, self.dtype, self.inner, self.kernel = cfg, device, dtype, w_inner_k, kernel
        self.gen = torch.Generator(device=device).manual_seed(seed)

    def __call__(self, name: str, layer: int, in_features: int, rows: int) -> torch.nn.Module:
        g, dev = self.cfg.group_size, self.device
        mod = self._cls(in_features, rows, bias=False, device=dev, dtype=self.dtype, group_size=g,
                        kernel=self.kernel, w_inner_k=self.inner, per_row=True)
        on_right = "x_f16RM_W" in self.kernel or "x_f16TC_W" in self.kernel
        if on_right:
            shape = (rows // 8, in_features // (16 * self.inner), 32, self.inner // 2)
        else:
            shape = (rows // 16, in_features // (16 * self.inner), 32, self.inner)
        w = torch.randint(-2 ** 31, 2 ** 31 - 1, shape, dtype=torch.int64, device=dev, generator=self.gen)
        mod.weight.data = w.to(torch.int32)
        mod.weight_reshaped = True
        # keep activations O(1) through the stack: w ~ N(0, 1/k) after dequant
        std = 1.0 / math.sqrt(in_features)
        scales = torch.rand(in_features // g, rows, device=dev, generator=self.gen) * 0.4 + 0.8
        zeros = torch.randn(in_features // g, rows, device=dev, generator=self.gen) * 0.05
        mod.scales_and_zeros.data = (torch.stack([scales, zeros], dim=2) * std).to(self.dtype).contiguous()
        mod.lut.data = torch.randn(rows, 16, device=dev, generator=self.gen).to(self.dtype)
        return mod


class DenseFactory:
    """16-bit `nn.Linear` baseline (what benchmark.py times before quantize_model)."""

    def __init__(self, cfg: DecodeConfig, device, dtype=torch.bfloat16, seed: int = 0):
        self.cfg, self.device, self.dtype = cfg, device, dtype
        self.gen = torch.Generator(device=device).manual_seed(seed)

    def __call__(self, name: str, layer: int, in_features: int, rows: int) -> torch.nn.Module:
        lin = torch.nn.Linear(in_features, rows, bias=False, device=self.device, dtype=self.dtype)
        w = torch.randn(rows, in_features, device=self.device, generator=self.gen) / math.sqrt(in_features)
        lin.weight.data = w.to(self.dtype)
        lin.weight.requires_grad_(False)
        return lin


# ---------------------------------------------------------------------------------------------------
# the decoder
# ---------------------------------------------------------------------------------------------------

class RMSNorm(torch.nn.Module):
    def __init__(self, dim, eps, device, dtype):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(dim, device=device, dtype=dtype), requires_grad=False)
        self.eps = eps

    def forward(self, x):
        xf = x.float()
        xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)
        return xf.to(x.dtype) * self.weight


def _rope_tables(cfg: DecodeConfig, device):
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, device=device, dtype=torch.float32) / cfg.head_dim))
    ang = torch.arange(cfg.max_seq, device=device, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([ang.cos(), ang.cos()], dim=-1), torch.cat([ang.sin(), ang.sin()], dim=-1)  # [S, d]


def _rope(x, cos, sin):
    # x [bs, h, d]; HF Llama convention: rotate_half over the two halves of the head dimension
    d2 = x.shape[-1] // 2
    rot = torch.cat([-x[..., d2:], x[..., :d2]], dim=-1)
    return (x.float() * cos + rot.float() * sin).to(x.dtype)


class DecodeLayer(torch.nn.Module):
    def __init__(self, cfg: DecodeConfig, idx: int, factory: Callable, rank: int, world: int, device, dtype, bs: int):
        super().__init__()
        if cfg.heads % world or cfg.kv_heads % world or cfg.inter % (16 * world) or cfg.hidden % (16 * world):
            raise ValueError(f"heads={cfg.heads}, kv_heads={cfg.kv_heads}, inter={cfg.inter}, hidden={cfg.hidden} "
                             f"must split over world_size={world} in whole heads / 16-row tiles")
        self.cfg, self.world = cfg, world
        self.hl, self.kvl = cfg.heads // world, cfg.kv_heads // world
        d = cfg.head_dim
        self.qkv = factory("qkv", idx, cfg.hidden, (self.hl + 2 * self.kvl) * d)
        self.o = factory("o", idx, cfg.heads * d, cfg.hidden // world)
        self.gate_up = factory("gate_up", idx, cfg.hidden, 2 * cfg.inter // world)
        self.down = factory("down", idx, cfg.inter, cfg.hidden // world)
        self.norm1 = RMSNorm(cfg.hidden, cfg.rms_eps, device, dtype)
        self.norm2 = RMSNorm(cfg.hidden, cfg.rms_eps, device, dtype)
        self.register_buffer("k_cache", torch.zeros(bs, self.kvl, cfg.max_seq, d, device=device, dtype=dtype), persistent=False)
        self.register_buffer("v_cache", torch.zeros(bs, self.kvl, cfg.max_seq, d, device=device, dtype=dtype), persistent=False)
        # forward_fused5: which stages the library fused (None: not tried yet; set by the first step)
        self._fuse = {"norm1": None, "norm2": None, "mlp": None}

    def forward(self, h, pos, cos, sin, mask, gather):
        cfg, d, bs = self.cfg, self.cfg.head_dim, h.shape[0]
        qkv = self.qkv(self.norm1(h))
        q = _rope(qkv[:, : self.hl * d].reshape(bs, self.hl, d), cos, sin)
        k = _rope(qkv[:, self.hl * d: (self.hl + self.kvl) * d].reshape(bs, self.kvl, d), cos, sin)
        v = qkv[:, (self.hl + self.kvl) * d:].reshape(bs, self.kvl, d)
        self.k_cache.index_copy_(2, pos, k.unsqueeze(2))
        self.v_cache.index_copy_(2, pos, v.unsqueeze(2))
        rep = self.hl // self.kvl
        qg = q.reshape(bs, self.kvl, rep, d)
        att = torch.matmul(qg, self.k_cache.transpose(2, 3)).float() * (1.0 / math.sqrt(d))  # [bs, kvl, rep, S]
        att = att.masked_fill(mask, float("-inf")).softmax(-1).to(h.dtype)
        ctx = torch.matmul(att, self.v_cache).reshape(bs, self.hl * d)
        h = h + gather(self.o(gather(ctx)))
        gu = self._split_gate_up(self.gate_up(self.norm2(h)))
        il = cfg.inter // self.world
        act = torch.nn.functional.silu(gu[:, :il]) * gu[:, il:]
        return h + gather(self.down(gather(act)))

    def forward_fused(self, h, delta, pos, cos_tab, sin_tab, gather, attn_scratch=None, attn_split=1):
        """Same layer on the HIP glue kernels (include/decode_glue_hip.h): 4 launches + 4 GEMMs.  `h` is the
        residual stream (updated in place), `delta` the previous layer's not-yet-added MLP output."""
        from . import decode_ops as G

        cfg, d = self.cfg, self.cfg.head_dim
        h, y = G.add_rmsnorm(h, delta, self.norm1.weight, self.norm1.eps)
        if attn_scratch is not None:
            ctx = G.rope_attn_split(self.qkv(y), cos_tab, sin_tab, pos, self.k_cache, self.v_cache, self.hl, self.kvl, d,
                                    1.0 / math.sqrt(d), attn_scratch, attn_split)
        else:
            attn = G.rope_attn_online if d in (64, 128) else G.rope_attn
            ctx = attn(self.qkv(y), cos_tab, sin_tab, pos, self.k_cache, self.v_cache, self.hl, self.kvl, d, 1.0 / math.sqrt(d))
        h, y = G.add_rmsnorm(h, gather(self.o(gather(ctx))), self.norm2.weight, self.norm2.eps)
        return h, gather(self.down(gather(G.swiglu(self._split_gate_up(self.gate_up(y)).contiguous()))))


    # ---- five launches per layer: every element-wise stage rides in a GEMM launch (tg_w4_gemm ABI 5) ----
    def _w4(self, lin, x, **kw):
        """`lin` (an Any4Linear / Int4Linear with Bint4 weights) through ops.w4_linear_fused; None if the library has no kernel
        with the requested stages for this problem."""
        from . import ops

        lut = getattr(lin, "lut", None)
        return ops.w4_linear_fused(x, lin.weight, lin.group_size, lin.scales_and_zeros, lut, **kw)

    def launches(self) -> int:
        """Kernel launches of one decode step of this layer on the fused path (after the first step has settled `_fuse`)."""
        f = self._fuse
        n = 5  # qkv GEMM, attention, o GEMM (+ residual), gate_up GEMM, down GEMM (+ residual)
        n += 0 if f.get("norm1") else 1                      # RMSNorm in front of qkv as its own launch
        if not f.get("mlp"):
            n += 0 if f.get("norm2") else 1                  # RMSNorm in front of gate_up
            n += 0 if f.get("swiglu") else 1                 # SwiGLU behind it
        return n

    def fusable(self) -> bool:
        """The four linears hold Bint4 weights behind the row-major weights-on-the-right kernel (what w4_linear_fused drives)."""
        ok = ("linear_y_f16RM_x_f16RM_W_any4TC", "linear_y_f16RM_x_f16RM_W_int4TC")
        return all(getattr(m, "kernel", None) in ok and getattr(m, "weight_reshaped", False) and getattr(m, "bias", None) is None
                   for m in (self.qkv, self.o, self.gate_up, self.down))

    def _split_gate_up(self, gu):
        """[bs, 2 il] in the weight's row order -> contiguous [gate | up] halves (what dg_swiglu reads)."""
        b = self.cfg.gate_up_interleave
        if not b:
            return gu
        return gu.view(gu.shape[0], -1, 2, b).transpose(1, 2).reshape(gu.shape[0], -1)

    def forward_fused5(self, h, pos, cos_tab, sin_tab, attn_scratch=None, attn_split=1):
        """TP = 1.  qkv GEMM (RMSNorm in its activation staging) -> RoPE + KV write + attention -> o GEMM (residual add in its
        store) -> gate_up GEMM (RMSNorm in its staging, SwiGLU in its store) -> down GEMM (residual add in its store): 5 launches
        instead of 8.  `h` [bs, hidden] is the residual stream, updated in place.  A stage the library cannot fuse for this
        problem (w4_linear_fused returns None) runs as its own launch, as in forward_fused."""
        from . import decode_ops as G

        cfg, d = self.cfg, self.cfg.head_dim
        f = self._fuse
        # ---- attention block
        qkv = self._w4(self.qkv, h, norm_weight=self.norm1.weight, norm_eps=self.norm1.eps) if f["norm1"] is not False else None
        if f["norm1"] is None:
            f["norm1"] = qkv is not None
        if qkv is None:
            qkv = self.qkv(G.add_rmsnorm(h, None, self.norm1.weight, self.norm1.eps)[1])
        if attn_scratch is not None:
            ctx = G.rope_attn_split(qkv, cos_tab, sin_tab, pos, self.k_cache, self.v_cache, self.hl, self.kvl, d,
                                    1.0 / math.sqrt(d), attn_scratch, attn_split)
        else:
            attn = G.rope_attn_online if d in (64, 128) else G.rope_attn
            ctx = attn(qkv, cos_tab, sin_tab, pos, self.k_cache, self.v_cache, self.hl, self.kvl, d, 1.0 / math.sqrt(d))
        if self._w4(self.o, ctx, residual=h, out=h) is None:       # (a residual add is available in every 4-bit kernel)
            G.add_rmsnorm(h, self.o(ctx), self.norm2.weight, self.norm2.eps, want_norm=False)
        # ---- MLP block: norm2 + SwiGLU inside the gate_up launch; else whichever of the two the library can fuse (a block too
        # large to stage on chip, e.g. 8 sequences at k = 4096, has no fused norm but still the SwiGLU store)
        il8 = cfg.gate_up_interleave == 8
        act = None
        if f["mlp"] is not False and il8:
            act = self._w4(self.gate_up, h, norm_weight=self.norm2.weight, norm_eps=self.norm2.eps, swiglu=True)
        if f["mlp"] is None:
            f["mlp"] = act is not None
        if act is None:
            gu = None
            if f["norm2"] is not False and not il8:
                gu = self._w4(self.gate_up, h, norm_weight=self.norm2.weight, norm_eps=self.norm2.eps)
                if f["norm2"] is None:
                    f["norm2"] = gu is not None
            if gu is None:
                y = G.add_rmsnorm(h, None, self.norm2.weight, self.norm2.eps)[1]
                if il8 and f.get("swiglu") is not False:
                    act = self._w4(self.gate_up, y, swiglu=True)
                    if f.get("swiglu") is None:
                        f["swiglu"] = act is not None
                if act is None:
                    gu = self.gate_up(y)
            if act is None:
                act = G.swiglu(self._split_gate_up(gu).contiguous())
        if self._w4(self.down, act, residual=h, out=h) is None:
            G.add_rmsnorm(h, self.down(act), self.norm2.weight, self.norm2.eps, want_norm=False)
        return h


class DecodeStack(torch.nn.Module):
    """Embedding -> `cfg.layers` decoder layers -> final norm -> LM head (16-bit, as in the reference:
    quantize_model skips the LM head by default, quantize.py:34-36)."""

    def __init__(self, cfg: DecodeConfig, linear_factory: Callable, device, dtype=torch.bfloat16, bs: int = 1,
                 rank: int = 0, world: int = 1, group=None, seed: int = 0, lm_head: bool = True,
                 fused: Optional[bool] = None, emulate_gather: bool = False, gather: str = "rccl", fuse_gemm_stages: bool = True):
        """fused: run the non-GEMM parts on the HIP glue kernels (default on a GPU) or as plain torch ops
        (the formulation the glue kernels are tested against; also what runs in the CPU plumbing tests).
        emulate_gather: TIMING ONLY -- build rank `rank` of `world` in a single process and replace every all-gather
        by a local copy of the rank's shard into all `world` slots (the values are meaningless): the per-GPU compute
        of a TP=world decode step, without the interconnect.
        fuse_gemm_stages: (fused, TP = 1, 4-bit linears with Bint4 weights) run a layer as FIVE launches -- RMSNorm inside the
        qkv / gate_up GEMMs' activation staging, the residual adds inside the o / down GEMMs' stores, SwiGLU inside gate_up's store
        when cfg.gate_up_interleave == 8 (DecodeLayer.forward_fused5) -- instead of 4 GEMMs + 4 glue kernels.
        gather: "rccl" = all_gather_into_tensor per exchange; "peer" = the one-shot peer-write gather of
        include/peer_gather_hip.h (any4_amd.shard.PeerWriteGather: one kernel per exchange, stores into the peers' buffers)."""
        super().__init__()
        if gather not in ("rccl", "peer"):
            raise ValueError("gather must be 'rccl' or 'peer'")
        self.gather_mode = gather
        self._peer = {}  # output width per rank -> PeerWriteGather
        self.cfg, self.bs, self.rank, self.world, self.group = cfg, bs, rank, world, group
        self.emulate_gather = emulate_gather
        self.fused = torch.device(device).type == "cuda" if fused is None else fused
        self.fuse_gemm_stages = fuse_gemm_stages
        gen = torch.Generator(device=device).manual_seed(seed)
        self.embed = torch.nn.Embedding(cfg.vocab, cfg.hidden, device=device, dtype=dtype)
        self.embed.weight.data = torch.randn(cfg.vocab, cfg.hidden, device=device, generator=gen).to(dtype)
        self.embed.weight.requires_grad_(False)
        self.layers = torch.nn.ModuleList(
            [DecodeLayer(cfg, i, linear_factory, rank, world, device, dtype, bs) for i in range(cfg.layers)])
        self.norm = RMSNorm(cfg.hidden, cfg.rms_eps, device, dtype)
        self.lm_head = None
        if lm_head:
            self.lm_head = torch.nn.Linear(cfg.hidden, cfg.vocab, bias=False, device=device, dtype=dtype)
            self.lm_head.weight.data = (torch.randn(cfg.vocab, cfg.hidden, device=device, generator=gen)
                                        / math.sqrt(cfg.hidden)).to(dtype)
            self.lm_head.weight.requires_grad_(False)
        cos, sin = _rope_tables(cfg, device)
        self.register_buffer("cos", cos, persistent=False)
        self.register_buffer("sin", sin, persistent=False)
        self.register_buffer("arange", torch.arange(cfg.max_seq, device=device), persistent=False)
        # static inputs (so a captured graph can be replayed with new values)
        self.register_buffer("tokens", torch.zeros(bs, dtype=torch.long, device=device), persistent=False)
        self.register_buffer("pos", torch.zeros(1, dtype=torch.long, device=device), persistent=False)
        self._graph = None
        self._out = None
        self._decodes = 0
        self.peer_poll_every = 16  # decode() calls between two non-blocking reads of the peer-gather status words
        # split-sequence attention: enough blocks per head to fill the 256 CUs (one scratch buffer, launches are stream-ordered)
        self._attn_scratch, self._attn_split = None, 1
        if self.fused:
            from . import decode_ops as G

            hl = cfg.heads // world
            # One block per head walks the cache at one CU's pace (Llama-3-8B, tokens/s at positions 136 / 500 / 900 / 1900: 624 / 583 /
            # 553 / 483); split 4 ways: 579 / 579 / 577 / 554, 8 ways: 537 / 535 / 537 / 539 (the cross-block combine costs ~1 us per
            # block of a head; profiles/r04_ab_attention_split.txt).  The grid is fixed when the step is captured, so the cache's
            # capacity decides: up to 1024 positions one block, up to 4096 four, beyond that eight.
            want = 1 if cfg.max_seq <= 1024 else 4 if cfg.max_seq <= 4096 else 8
            self._attn_split = max(1, min(want, 256 // max(1, bs * hl)))
            if os.environ.get("ANY4_ATTN_SPLIT"):  # developer override (A/B of the threshold above)
                self._attn_split = max(1, int(os.environ["ANY4_ATTN_SPLIT"]))
            if self._attn_split > 1:
                self._attn_scratch = G.rope_attn_split_scratch(bs, hl, cfg.head_dim, self._attn_split, device)

    # [bs, n/G] on every rank -> [bs, n], rank-major feature order (== row order of the unsharded weight)
    def _gather(self, y):
        if self.world == 1:
            return y
        y = y.contiguous()
        if self.emulate_gather:
            return y.repeat(1, self.world)
        if self.gather_mode == "peer":
            from .shard import PeerWriteGather

            pg = self._peer.get(y.shape[1])
            if pg is None:
                pg = self._peer[y.shape[1]] = PeerWriteGather(self.bs, y.shape[1], group=self.group, device=y.device, dtype=y.dtype)
            return pg.gather(y)
        parts = torch.empty((self.world,) + tuple(y.shape), dtype=y.dtype, device=y.device)
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(parts, y, group=self.group)
        else:
            dist.all_gather(list(parts.unbind(0)), y, group=self.group)
        if y.shape[0] == 1:
            return parts.view(1, -1)
        return parts.movedim(0, 1).reshape(y.shape[0], -1)

    @torch.no_grad()
    def step(self) -> torch.Tensor:
        """One decode step on the static inputs `self.tokens` [bs], `self.pos` [1]; returns logits (or the
        final hidden state when built without LM head)."""
        if self._peer or (self.gather_mode == "peer" and self.world > 1 and not self.emulate_gather):
            before = {w: pg._calls for w, pg in self._peer.items()}
            out = self._step()
            # a gather object alternates between two buffers per call: an odd number of calls per step would make the last
            # call of one step and the first of the next share a buffer (and a captured graph replays the SAME buffers)
            for w, pg in self._peer.items():
                if (pg._calls - before.get(w, 0)) & 1:
                    pg.gather(torch.zeros(1, w, dtype=pg.dtype, device=pg.device))
            return out
        return self._step()

    def _step(self) -> torch.Tensor:
        pos = self.pos
        if self.fused:
            from . import decode_ops as G

            h, delta = self.embed(self.tokens), None
            if self.fuse_gemm_stages and self.world == 1 and all(layer.fusable() for layer in self.layers):
                for layer in self.layers:
                    h = layer.forward_fused5(h, pos, self.cos, self.sin, self._attn_scratch, self._attn_split)
            else:
                for layer in self.layers:
                    h, delta = layer.forward_fused(h, delta, pos, self.cos, self.sin, self._gather, self._attn_scratch, self._attn_split)
            _, y = G.add_rmsnorm(h, delta, self.norm.weight, self.norm.eps)
            if self.lm_head is None:
                return y
            # the un-quantised LM head (quantize.py:34-36 skips it): a streaming GEMV for up to four rows, else torch's GEMM
            logits = G.linear16(y, self.lm_head.weight) if self.lm_head.bias is None else None
            return logits if logits is not None else self.lm_head(y)
        cos = self.cos.index_select(0, pos).view(1, 1, -1)
        sin = self.sin.index_select(0, pos).view(1, 1, -1)
        mask = (self.arange > pos).view(1, 1, 1, -1)
        h = self.embed(self.tokens)
        for layer in self.layers:
            h = layer(h, pos, cos, sin, mask, self._gather)
        h = self.norm(h)
        return self.lm_head(h) if self.lm_head is not None else h

    @torch.no_grad()
    def capture(self, warmup: int = 3) -> None:
        """Capture `step()` in a hipGraph (torch.cuda.graph).  All ranks must call it together when world > 1."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._out = self.step()
        self._graph = g
        if self.fused and self.fuse_gemm_stages and self.world == 1 and all(layer.fusable() for layer in self.layers):
            self.kernels_per_layer = self.layers[0].launches()
            # + embedding gather, final norm, LM head
            self.graph_nodes = sum(layer.launches() for layer in self.layers) + 2 + (1 if self.lm_head is not None else 0)

    @torch.no_grad()
    def decode(self, tokens: torch.Tensor, position: int) -> torch.Tensor:
        """Feed `tokens` [bs] at sequence position `position`; graph replay if captured, else eager."""
        if not 0 <= int(position) < self.cfg.max_seq:
            raise ValueError(f"position {position} outside the KV cache [0, {self.cfg.max_seq})")
        self.tokens.copy_(tokens)
        self.pos.fill_(position)
        out = self._out if self._graph is not None else None
        if self._graph is not None:
            self._graph.replay()
        else:
            out = self.step()
        # peer-write gathers: a slice that did not arrive is NaN in the buffers; here the status words are read back on a
        # cadence without synchronising the device (PeerWriteGather.poll) so that a slow / dead rank raises instead of decoding on
        self._decodes += 1
        if self._peer and self._decodes % self.peer_poll_every == 0:
            for pg in self._peer.values():
                pg.poll()
        return out


def memory_allocated_mb(device=None) -> float:
    """ROCm replacement of the reference's nvidia-smi based MemoryTracker (utils.py:241): peak bytes the
    caching allocator handed out on this device, in MiB."""
    return torch.cuda.max_memory_allocated(device) / 2 ** 20
