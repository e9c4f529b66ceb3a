"""ctypes front-end of include/decode_glue_hip.h: the non-GEMM kernels of a batch-1 decode step.

Every function takes torch tensors on a ROCm device, allocates the output with `torch.empty` (caching
allocator, current stream) and launches on `torch.cuda.current_stream()`; they are legal inside
`torch.cuda.graph` capture.  No CPU fallback: a CPU tensor raises.
"""
from __future__ import annotations

import torch

from . import _lib


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return _lib.TG_BF16
    if t.dtype == torch.float16:
        return _lib.TG_F16
    raise RuntimeError(f"decode glue kernels need bf16 or fp16 tensors, got {t.dtype}")


def _gpu(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda or not t.is_contiguous()):
            raise RuntimeError("decode glue kernels need contiguous tensors on a ROCm device (there is no CPU fallback)")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    if _raw_stream is not None:
        return _raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


def add_rmsnorm(h: torch.Tensor, delta, weight: torch.Tensor, eps: float, want_norm: bool = True):
    """(h + delta, rmsnorm(h + delta) * weight); delta may be None.  h is updated IN PLACE when delta is given
    (the residual stream is a running sum).  h [rows, dim]."""
    _gpu(h, delta, weight)
    rows, dim = h.shape
    y = torch.empty_like(h) if want_norm else None
    _lib.check(_lib.load().dg_add_rmsnorm(h.data_ptr(), None if delta is None else delta.data_ptr(), weight.data_ptr(),
                                          h.data_ptr(), None if y is None else y.data_ptr(), rows, dim, float(eps),
                                          _dt(h), h.device.index, _stream(h)), "dg_add_rmsnorm")
    return h, y


def rope_kv(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, pos: torch.Tensor, k_cache: torch.Tensor,
            v_cache: torch.Tensor, hl: int, kvl: int, d: int) -> torch.Tensor:
    """qkv [bs, (hl + 2 kvl) d] -> rotated q [bs, hl, d]; rotated k and v are written into the caches at `pos`."""
    _gpu(qkv, cos, sin, pos, k_cache, v_cache)
    if cos.dtype != torch.float32 or pos.dtype != torch.int64:
        raise RuntimeError("rope tables must be float32 and pos int64")
    bs, max_seq = qkv.shape[0], k_cache.shape[2]
    q = torch.empty((bs, hl, d), dtype=qkv.dtype, device=qkv.device)
    _lib.check(_lib.load().dg_rope_kv(qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos.data_ptr(), q.data_ptr(),
                                      k_cache.data_ptr(), v_cache.data_ptr(), bs, hl, kvl, d, max_seq, _dt(qkv),
                                      qkv.device.index, _stream(qkv)), "dg_rope_kv")
    return q


def decode_attn(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, pos: torch.Tensor, scale: float) -> torch.Tensor:
    """q [bs, hl, d], caches [bs, kvl, max_seq, d] -> context [bs, hl * d] over positions 0..pos."""
    _gpu(q, k_cache, v_cache, pos)
    bs, hl, d = q.shape
    kvl, max_seq = k_cache.shape[1], k_cache.shape[2]
    out = torch.empty((bs, hl * d), dtype=q.dtype, device=q.device)
    _lib.check(_lib.load().dg_decode_attn(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), pos.data_ptr(),
                                          out.data_ptr(), bs, hl, kvl, d, max_seq, float(scale), _dt(q), q.device.index,
                                          _stream(q)), "dg_decode_attn")
    return out


def rope_attn(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, pos: torch.Tensor, k_cache: torch.Tensor,
              v_cache: torch.Tensor, hl: int, kvl: int, d: int, scale: float) -> torch.Tensor:
    """rope_kv + decode_attn in one launch: qkv [bs, (hl + 2 kvl) d] -> context [bs, hl * d]; caches updated at `pos`."""
    _gpu(qkv, cos, sin, pos, k_cache, v_cache)
    if cos.dtype != torch.float32 or pos.dtype != torch.int64:
        raise RuntimeError("rope tables must be float32 and pos int64")
    bs, max_seq = qkv.shape[0], k_cache.shape[2]
    out = torch.empty((bs, hl * d), dtype=qkv.dtype, device=qkv.device)
    _lib.check(_lib.load().dg_rope_attn(qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos.data_ptr(), k_cache.data_ptr(),
                                        v_cache.data_ptr(), out.data_ptr(), bs, hl, kvl, d, max_seq, float(scale), _dt(qkv),
                                        qkv.device.index, _stream(qkv)), "dg_rope_attn")
    return out


def rope_attn_online(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, pos: torch.Tensor, k_cache: torch.Tensor,
                     v_cache: torch.Tensor, hl: int, kvl: int, d: int, scale: float) -> torch.Tensor:
    """rope_attn built for latency (head_dim 64 / 128): one barrier, every load issued up front, softmax statistics combined
    flash-decoding style -- the same caches bit for bit, the output within 16-bit rounding of rope_attn's."""
    _gpu(qkv, cos, sin, pos, k_cache, v_cache)
    if cos.dtype != torch.float32 or pos.dtype != torch.int64:
        raise RuntimeError("rope tables must be float32 and pos int64")
    bs, max_seq = qkv.shape[0], k_cache.shape[2]
    out = torch.empty((bs, hl * d), dtype=qkv.dtype, device=qkv.device)
    _lib.check(_lib.load().dg_rope_attn_online(qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos.data_ptr(), k_cache.data_ptr(),
                                               v_cache.data_ptr(), out.data_ptr(), bs, hl, kvl, d, max_seq, float(scale), _dt(qkv),
                                               qkv.device.index, _stream(qkv)), "dg_rope_attn_online")
    return out


def rope_attn_split_scratch(bs: int, hl: int, d: int, nsplit: int, device) -> torch.Tensor:
    """Zeroed scratch buffer for rope_attn_split (counters + per-chunk partials); reusable by stream-ordered launches."""
    n = _lib.load().dg_rope_attn_split_scratch_bytes(bs, hl, d, nsplit)
    return torch.zeros((n + 3) // 4, dtype=torch.int32, device=device)


def rope_attn_split(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, pos: torch.Tensor, k_cache: torch.Tensor,
                    v_cache: torch.Tensor, hl: int, kvl: int, d: int, scale: float, scratch: torch.Tensor, nsplit: int) -> torch.Tensor:
    """rope_attn with the sequence split over `nsplit` blocks per head (fills the GPU at batch 1 / long contexts)."""
    _gpu(qkv, cos, sin, pos, k_cache, v_cache, scratch)
    if cos.dtype != torch.float32 or pos.dtype != torch.int64:
        raise RuntimeError("rope tables must be float32 and pos int64")
    bs, max_seq = qkv.shape[0], k_cache.shape[2]
    out = torch.empty((bs, hl * d), dtype=qkv.dtype, device=qkv.device)
    _lib.check(_lib.load().dg_rope_attn_split(qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos.data_ptr(), k_cache.data_ptr(),
                                              v_cache.data_ptr(), out.data_ptr(), scratch.data_ptr(), scratch.numel() * 4, bs, hl,
                                              kvl, d, max_seq, float(scale), nsplit, _dt(qkv), qkv.device.index, _stream(qkv)),
               "dg_rope_attn_split")
    return out


def swiglu(gu: torch.Tensor) -> torch.Tensor:
    """gu [bs, 2 il] = [gate | up] -> silu(gate) * up [bs, il]."""
    _gpu(gu)
    bs, il = gu.shape[0], gu.shape[1] // 2
    out = torch.empty((bs, il), dtype=gu.dtype, device=gu.device)
    _lib.check(_lib.load().dg_swiglu(gu.data_ptr(), out.data_ptr(), bs, il, _dt(gu), gu.device.index, _stream(gu)), "dg_swiglu")
    return out


def linear16(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """y = x @ weight.T for a ROW-MAJOR 16-bit weight [n][k] (an nn.Linear's) and 1 ... 4 rows of x: the decode step's LM head.
    Returns None when the library has no instantiation for the shape (the caller keeps its GEMM)."""
    _gpu(x)
    m, k = x.shape
    n = weight.shape[0]
    if not (1 <= m <= (2 if k == 8192 else 4) and k in (2048, 4096, 8192) and weight.dtype == x.dtype and weight.is_contiguous() and x.is_contiguous()
            and weight.shape[1] == k and x.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0):
        return None
    y = torch.empty((m, n), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().dg_linear16(x.data_ptr(), weight.data_ptr(), y.data_ptr(), m, n, k, _dt(x), x.device.index, _stream(x)), "dg_linear16")
    return y

