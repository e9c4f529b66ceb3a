#!/usr/bin/env python3
"""One linear layer: 16-bit nn.Linear vs the quantized module (counterpart of the reference's microbenchmark.py:20-59,
same flags and protocol: warm-up 50, 100 iterations, wall-clock and device-only times).

    python tools/microbenchmark.py --input-dim 4096 --output-dim 4096 --quantize anyq
    python tools/microbenchmark.py --quantize intq --quantize-args group_size=64
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def parse_kv(s):
    out = {}
    for item in (s or "").split(","):
        if item:
            k, v = item.split("=")
            out[k] = int(v) if v.lstrip("-").isdigit() else (v == "True" if v in ("True", "False") else v)
    return out


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser(description="Benchmark quantization on a linear layer.")
    ap.add_argument("--batch-size", type=int, default=1)
    ap.add_argument("--seqlen", type=int, default=1)
    ap.add_argument("--input-dim", type=int, default=4096)
    ap.add_argument("--output-dim", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--quantize", default="anyq", choices=["anyq", "intq", "int8", "none"])
    ap.add_argument("--quantize-args", type=str, default="")
    ap.add_argument("--dtype", default="bfloat16")
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU (no CPU fallback)")
    from any4_amd import quantize as Q
    from any4_amd.bench_utils import benchmark_cuda_only_in_ms, benchmark_in_ms, benchmark_kernels_only_in_ms

    dtype = getattr(torch, a.dtype)
    x = torch.randn(a.batch_size * a.seqlen, a.input_dim, dtype=dtype, device="cuda")
    linear = torch.nn.Linear(a.input_dim, a.output_dim, dtype=dtype, device="cuda", bias=False)
    t, tc = benchmark_in_ms(linear, a.warmup, a.iters, x), benchmark_cuda_only_in_ms(linear, a.warmup, a.iters, x)
    tk = benchmark_kernels_only_in_ms(linear, a.warmup, a.iters, x)
    print("Baseline:")
    print(f"\tTotal: {t:.4f} ms\tCUDA: {tc:.4f} ms\tkernels only: {tk:.4f} ms")
    if a.quantize != "none":
        kw = parse_kv(a.quantize_args)
        if a.quantize == "anyq":
            q = Q.anyq_layer(linear, pseudo=False, **kw)
        elif a.quantize == "intq":
            q = Q.intq_layer(linear, pseudo=False, **kw)
        else:
            import modules
            from tinygemm_lib.utils import group_quantize_tensor

            g = kw.get("group_size", 128)
            q = modules.Int8Linear(a.input_dim, a.output_dim, bias=False, device="cuda", dtype=dtype, group_size=g)
            q.weight.data, q.scales_and_zeros.data = group_quantize_tensor(linear.weight, 8, g)
            q.reshape_weight()
        qt, qtc = benchmark_in_ms(q, a.warmup, a.iters, x), benchmark_cuda_only_in_ms(q, a.warmup, a.iters, x)
        qtk = benchmark_kernels_only_in_ms(q, a.warmup, a.iters, x)
        print("Quantized:")
        print(f"\tTotal: {qt:.4f} ms\tCUDA: {qtc:.4f} ms\tkernels only: {qtk:.4f} ms")
        print(f"Speedup:\tTotal {t / qt:.2f}x\tCUDA {tc / qtc:.2f}x\tkernels only {tk / qtk:.2f}x")


if __name__ == "__main__":
    main()
