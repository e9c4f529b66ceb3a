#!/usr/bin/env python3
"""Time torch.ops.tinygemm.tinygemm_y_f16RM_x_f16RM_w_any4TC for one n x k layer per graph node at several m (developer tool):
    ANY4_LARGE_M=0 python dev/large_m.py     (always the 4-bit kernels)      ANY4_LARGE_M=1 python dev/large_m.py   (always dequantise + GEMM)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", default="32,48,64,96,128,256,512,1024")
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=16)
    a = ap.parse_args()
    import tinygemm  # noqa: F401
    T = torch.ops.tinygemm
    dev = torch.device("cuda:0")
    N, K, g = a.n, a.k, 128
    ws = [torch.randint(-2 ** 31, 2 ** 31 - 1, (N // 8, K // 64, 32, 2), dtype=torch.int64, device=dev).to(torch.int32) for _ in range(a.layers)]
    sz = torch.rand(K // g, N, 2, device=dev).bfloat16()
    lut = torch.randn(N, 16, device=dev).bfloat16()
    for m in [int(v) for v in a.m.split(",")]:
        x = torch.randn(m, K, device=dev).bfloat16()
        f = lambda w: T.tinygemm_y_f16RM_x_f16RM_w_any4TC(x, w, g, sz, lut, True)  # noqa: E731
        for w in ws[:2]:
            f(w)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(gr):
                ys = [f(w) for w in ws]
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"m={m:5d} {N}x{K}: {e0.elapsed_time(e1) * 1e3 / 20 / a.layers:8.2f} us per layer   (ANY4_LARGE_M={os.environ.get('ANY4_LARGE_M', 'default')})")
        del gr, ys


if __name__ == "__main__":
    main()
