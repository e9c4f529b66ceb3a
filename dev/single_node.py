#!/usr/bin/env python3
"""One n x k layer per hipGraph node, N distinct layers chained (developer tool; works with the TG_DEV_MIN variant builds that
tools/quick_bench.py does not): microseconds per node for each --m.
    python dev/single_node.py --m 1,4,8,16 [--n 4096 --k 4096 --layers 64 --g 128]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", default="1,4,8,16")
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--g", type=int, default=128)
    ap.add_argument("--layers", type=int, default=64)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--numerics", default="fast", choices=["fast", "reference"])
    a = ap.parse_args()
    import any4_amd
    from any4_amd import ops
    import tinygemm  # noqa: F401

    any4_amd.set_numerics(a.numerics)

    dev = torch.device("cuda:0")
    N, K = a.n, a.k
    ws = [torch.randint(-2 ** 31, 2 ** 31 - 1, (N // 8, K // 64, 32, 2), dtype=torch.int64, device=dev).to(torch.int32) for _ in range(a.layers)]
    sz = torch.rand(K // a.g, N, 2, device=dev).bfloat16()
    lut = torch.randn(N, 16, device=dev).bfloat16()
    for m in [int(v) for v in a.m.split(",")]:
        x = torch.randn(m, K, device=dev).bfloat16()
        for w in ws[:2]:
            ops.w4_linear_fused(x, w, a.g, sz, lut)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g):
                ys = [ops.w4_linear_fused(x, w, a.g, sz, lut) for w in ws]
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps / a.layers
        print(f"m={m:2d} {N}x{K} g={a.g}: {us:6.2f} us per node   plan={ops.gemm_w4_plan(m, N, K, a.g, 2, True, 4, batch=1, detail=True)}")
        del g, ys


if __name__ == "__main__":
    main()
