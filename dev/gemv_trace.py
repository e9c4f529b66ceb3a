#!/usr/bin/env python3
"""Phase timeline of w4_gemv_kernel inside a decode step (developer builds with -DGEMV_TRACE=1 only).

    dev/build_variant.sh trace -DTG_DEV_MIN=99 -DGEMV_TRACE=1 && cp variants/trace.so any4_amd/lib/libtinygemm_hip.so   (on the box)
    python dev/gemv_trace.py [--layers 3]

Every gemv launch of the captured step writes, per workgroup, s_memrealtime stamps (10 ns) of: entry, loads issued, table built,
barrier passed, first step consumed, end.  Printed per launch of the LAST layer, relative to the previous launch's last end.
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


STAMPS = ["entry", "arguments loaded", "loads issued", "table built", "barrier passed", "first step done", "end"]
ORDER = [0, 6, 1, 2, 3, 4, 5]


def show(t, nslots, names):
    prev_end = None
    for s in range(nslots):
        ts = t[s]
        on = ts[:, 0] > 0
        if not on.any():
            continue
        ts = ts[on]
        base = ts[:, 0].min()
        line = f"[{s:2d}] {names[s % len(names)]:24s} wgs {on.sum():3d}"
        if prev_end is not None:
            line += f"  first entry {base - prev_end:+6.2f} us after the previous gemv's last end"
        print(line)
        for nm, j in zip(STAMPS, ORDER):
            v = ts[:, j] - base
            print(f"       {nm:18s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f}" +
                  (f"   (workgroups 0-255: median {np.median(v[:256]):6.2f}; 256-511: median {np.median(v[256:]):6.2f})" if len(v) > 256 else ""))
        prev_end = ts[:, 5].max()


P16_STAMPS = ["entry", "loads issued", "table built", "barrier passed", "fragments arranged", "main loop done", "partial sums visible"]


def show_p16(t, nslots):
    prev_end = None
    for s in range(nslots):
        ts = t[s]
        on = ts[:, 0] > 0
        if not on.any():
            continue
        ts = ts[on]
        base = ts[:, 0].min()
        line = f"[{s:2d}] pair16 wgs {on.sum():3d}"
        if prev_end is not None:
            line += f"  first entry {base - prev_end:+6.2f} us after the previous launch's last stamp"
        print(line)
        for j, nm in enumerate(P16_STAMPS):
            v = ts[:, j] - base
            print(f"       {nm:22s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f}")
        prev_end = ts[:, 6].max()


def repeat_p16(n, m):
    """n launches of one 4096 x 4096 layer with m (5 ... 16) activation rows from a graph: w4_gemm_pair16_kernel's stamps."""
    from any4_amd import _lib, ops
    import tinygemm  # noqa: F401

    L = _lib.load()
    dev = torch.device("cuda:0")
    N = K = 4096
    ws = [torch.randint(-2 ** 31, 2 ** 31 - 1, (N // 8, K // 64, 32, 2), dtype=torch.int64, device=dev).to(torch.int32) for _ in range(n)]
    sz = torch.rand(K // 128, N, 2, device=dev).bfloat16()
    lut = torch.randn(N, 16, device=dev).bfloat16()
    x = torch.randn(m, K, device=dev).bfloat16()
    buf = torch.zeros(n * 512 * 8, dtype=torch.int64, device=dev)
    for w in ws[:2]:
        ops.w4_linear_fused(x, w, 128, sz, lut)
    torch.cuda.synchronize()
    L.tg_dev_p16_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.tg_dev_p16_trace.restype = None
    L.tg_dev_p16_trace(buf.data_ptr(), n)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            y = x
            for w in ws:
                y = ops.w4_linear_fused(y, w, 128, sz, lut)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(n, 512, 8).astype(np.float64) * 0.01
    show_p16(t, n)


def repeat(n, m=1):
    from any4_amd import _lib, ops
    import tinygemm  # noqa: F401

    L = _lib.load()
    dev = torch.device("cuda:0")
    N = K = 4096
    ws = [torch.randint(-2 ** 31, 2 ** 31 - 1, (N // 8, K // 64, 32, 2), dtype=torch.int64, device=dev).to(torch.int32) for _ in range(n)]
    sz = torch.rand(K // 128, N, 2, device=dev).bfloat16()
    lut = torch.randn(N, 16, device=dev).bfloat16()
    x = torch.randn(m, K, device=dev).bfloat16()
    buf = torch.zeros(n * 512 * 8, dtype=torch.int64, device=dev)
    for w in ws[:2]:
        ops.w4_linear_fused(x, w, 128, sz, lut)
    torch.cuda.synchronize()
    L.tg_dev_gemv_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.tg_dev_gemv_trace.restype = None
    L.tg_dev_gemv_trace(buf.data_ptr(), n)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            y = x
            for w in ws:
                y = ops.w4_linear_fused(y, w, 128, sz, lut)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(n, 512, 8).astype(np.float64) * 0.01
    show(t, n, ["4096x4096"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--repeat", type=int, default=0, help="instead of a decode step: this many launches of ONE 4096 x 4096 layer shape back to back (distinct weights), from a graph")
    ap.add_argument("--m", type=int, default=1, help="with --repeat: activation rows (9 ... 16: the stamps of w4_gemm_pair16_kernel)")
    a = ap.parse_args()
    if a.repeat and a.m > 8:
        return repeat_p16(a.repeat, a.m)
    if a.repeat:
        return repeat(a.repeat, a.m)
    from any4_amd import _lib
    from any4_amd.decode import Any4Factory, DecodeConfig, DecodeStack

    L = _lib.load()
    dev = torch.device("cuda:0")
    cfg = DecodeConfig.llama3_8b(max_seq=1024)
    cfg.layers = a.layers
    cfg.gate_up_interleave = 8
    slots = 4 * a.layers + 8
    buf = torch.zeros(slots * 512 * 8, dtype=torch.int64, device=dev)
    stack = DecodeStack(cfg, Any4Factory(cfg, dev, torch.bfloat16, seed=1), dev, torch.bfloat16, bs=1, lm_head=False)
    tok = torch.randint(0, cfg.vocab, (1,), device=dev)
    for i in range(3):
        stack.decode(tok, 130 + i)
    L.tg_dev_gemv_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.tg_dev_gemv_trace.restype = None
    L.tg_dev_gemv_trace(buf.data_ptr(), slots)
    abuf = torch.zeros(64 * 8, dtype=torch.int64, device=dev)
    L.tg_dev_attn_trace.argtypes = [ctypes.c_void_p]
    L.tg_dev_attn_trace.restype = None
    L.tg_dev_attn_trace(abuf.data_ptr())
    stack.capture(warmup=0)
    for i in range(5):
        stack.decode(tok, 140 + i)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(slots, 512, 8).astype(np.float64) * 0.01  # us
    show(t, 4 * a.layers, ["qkv(+norm)", "o(+res)", "gate_up(+norm,swiglu)", "down(+res)"])
    at = abuf.cpu().numpy().reshape(64, 8).astype(np.float64)[:32] * 0.01
    base = at[:, 0].min()
    qkv_end = t[4 * (a.layers - 1)][:, 5].max()
    print(f"attention (last layer): first entry {base - qkv_end:+.2f} us after the qkv gemv's last end")
    for j, nm in enumerate(["entry", "pos known", "loads issued", "rotated", "scores", "values", "barrier", "end"]):
        v = at[:, j] - base
        print(f"       {nm:14s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f}")


if __name__ == "__main__":
    main()
