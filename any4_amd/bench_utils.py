"""Timing helpers with the reference's names and protocols (utils.py:32-112 of facebookresearch/any4), on HIP.

benchmark_in_ms           wall clock around `iters` calls, device synchronised before and after (includes host
                          dispatch -- at batch 1 that is most of the time of an eager module call)
benchmark_cuda_only_in_ms device time only: one HIP-event pair per call, with a cache flush between calls so every
                          call streams its weights from HBM.  The reference flushes 256 MB (utils.py:68,98), which is
                          smaller than MI355X's L2 + Infinity Cache (32 + 256 MiB); this one writes 1 GiB.
memory_allocated_mb       ROCm replacement of the nvidia-smi based MemoryTracker (utils.py:241)
"""
from __future__ import annotations

import time

import torch


def benchmark_in_ms(f, warmup: int, iters: int, *args, **kwargs) -> float:
    for _ in range(warmup):
        f(*args, **kwargs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        f(*args, **kwargs)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / iters


_flush = None


def benchmark_cuda_only_in_ms(f, warmup: int, iters: int, *args, **kwargs) -> float:
    global _flush
    if _flush is None or _flush.device != torch.device("cuda", torch.cuda.current_device()):
        _flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda")  # 1 GiB
    for _ in range(warmup):
        f(*args, **kwargs)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for e0, e1 in ev:
        _flush.zero_()
        e0.record()
        f(*args, **kwargs)
        e1.record()
    torch.cuda.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1 in ev) / iters


def benchmark_kernels_only_in_ms(f, warmup: int, iters: int, *args, **kwargs) -> float:
    """Sum of the device kernels' own durations per call (torch.profiler / roctracer), cache flushed between calls.
    An empty HIP-event pair already measures ~4.3 us on MI355X (tools/ubench/cold_launch.hip), which is comparable to
    a whole 4-bit 4096x4096 GEMV; this number leaves that bracket out."""
    global _flush
    if _flush is None:
        _flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    for _ in range(warmup):
        f(*args, **kwargs)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        for _ in range(iters):
            _flush.zero_()
            f(*args, **kwargs)
        torch.cuda.synchronize()
    total_us = 0.0
    for ev in prof.key_averages():
        if "fill" in ev.key.lower() or "memset" in ev.key.lower():  # the flush
            continue
        total_us += getattr(ev, "self_device_time_total", 0.0) or getattr(ev, "self_cuda_time_total", 0.0)
    return total_us / 1e3 / iters


def memory_allocated_mb(device=None) -> float:
    return torch.cuda.max_memory_allocated(device) / 2 ** 20
