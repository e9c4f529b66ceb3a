#!/usr/bin/env python3
"""One quantized layer per launch at MANY activation rows, through the modules (what a prefill / a wide decode batch issues): device time
per forward inside a hipGraph over several distinct layers, against nn.Linear (bf16) of the same shape and -- with --library -- the opt-in
dequantise + vendor GEMM route.   python dev/many_rows_bench.py [--shapes "128,4096,4096;..."] [--layers 6] [--library]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def graph_time(fns, x, reps=20):
    for f in fns:
        f(x)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for f in fns:
            f(x)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for f in fns:
                f(x)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(fns))


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="32,4096,4096;33,4096,4096;48,4096,4096;64,4096,4096;65,4096,4096;128,4096,4096;256,4096,4096;512,4096,4096;"
                                        "1024,4096,4096;2048,4096,4096;128,14336,4096;128,4096,14336;512,1024,4096;512,14336,4096;512,4096,14336")
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--library", action="store_true")
    a = ap.parse_args()
    from any4_amd import ops
    from any4_amd import quantize as Q

    for shape in a.shapes.split(";"):
        m, n, k = (int(v) for v in shape.split(","))
        x = torch.randn(m, k, dtype=torch.bfloat16, device="cuda") * 0.05
        lins = [torch.nn.Linear(k, n, dtype=torch.bfloat16, device="cuda", bias=False) for _ in range(a.layers)]
        qs = [Q.anyq_layer(torch.nn.Linear(k, n, dtype=torch.bfloat16, device="cuda", bias=False), pseudo=False) for _ in range(2)]
        import copy   # (distinct memory per layer: copies of two quantized prototypes are as good for timing)
        mods = []
        for i in range(a.layers):
            q = copy.deepcopy(qs[i % 2])
            mods.append(q)
        t_lin = graph_time(lins, x)
        t_q = graph_time(mods, x)
        line = f"m={m:5d} n={n:6d} k={k:6d}  nn.Linear {t_lin:8.2f} us   quantized {t_q:8.2f} us   speed-up {t_lin / t_q:5.2f} x   {2.0 * m * n * k / t_q * 1e-6:7.1f} TFLOP/s"
        if a.library:
            os.environ["ANY4_LARGE_M_GEMM"] = "library"
            ops._LARGE_M = None
            try:
                t_lib = graph_time(mods, x)
                line += f"   library route {t_lib:8.2f} us"
            finally:
                os.environ.pop("ANY4_LARGE_M_GEMM", None)
                ops._LARGE_M = None
        print(line, flush=True)


if __name__ == "__main__":
    main()
