#!/usr/bin/env python3
"""Seeded random shapes through the torch ops (both operand sides, every quantisation type) against the CPU oracle, with the loosest
tolerance any path may use (reference-faithful result + 0.5 ulp + (4e-6 + 2^-9) sum|x w|): a developer smoke for routing changes.
    python tests/stress_random.py [--cases 60] [--seed 1]"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import any4_amd
    from any4_amd import ops
    from tests.conftest import from_bits16
    from tests.test_gpu_parity import oracle_weights, rand_problem, run_rm, ulp16
    import tinygemm  # noqa: F401
    from oracle import oracle  # (test infrastructure: this script lives under tests/ for that reason)
    oracle.build()
    T = torch.ops.tinygemm
    rng = random.Random(a.seed)
    bad = 0
    for i in range(a.cases):
        qtype = rng.choice(["any4_rowwise", "any4_rowwise", "int4", "any4_global", "mx4"])
        g = 32 if qtype == "mx4" else rng.choice([32, 64, 128, 256])
        k = rng.choice([512, 1024, 2048, 4096, 4096, 4096, 5120, 8192, 11008 // 64 * 64, 14336])
        k = k // g * g
        m = rng.choice([1, 2, 3, 5, 8, 9, 13, 16, 17, 33, 64, 65, 100, 130, 300])
        n = rng.choice([16, 48, 64, 136 // 16 * 16, 256, 1024, 4096, 5120, 6144, 8208 // 16 * 16])
        on_right = rng.random() < 0.7
        inner = rng.choice([2, 4] if not on_right else [2, 4, 8])
        if k % (16 * inner) or (k * n > 40_000_000):
            continue
        dtype = torch.bfloat16 if qtype == "mx4" else rng.choice([torch.bfloat16, torch.float16])
        codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, dtype=dtype, seed=1000 + i)
        with any4_amd.weight_format(rng.choice(["native", "reference"])):
            y = run_rm(T, codes, x, qinfo, lut, g, qtype, on_right, inner)
        w = from_bits16(oracle_weights(oracle, codes, g, qtype, qinfo, lut, dtype), dtype).double()
        x64 = x.double()
        y64 = (x64 @ w.t()).numpy()
        S = (x64.abs() @ w.abs().t()).numpy()
        eps16 = 2.0 ** -9 if dtype == torch.bfloat16 else 2.0 ** -12
        tol = 0.5 * ulp16(y64, dtype) * (1 + 2.0 ** -7) + (4e-6 + eps16) * S + 1e-37
        got = y.detach().double().cpu().numpy()[:, :n]
        nb = int((np.abs(got - y64) > tol).sum())
        plan = ops.gemm_w4_plan(m, n, k, g, {"int4": 0, "any4_global": 1, "any4_rowwise": 2, "mx4": 3}[qtype], on_right, inner, dtype, 1, detail=True) if True else "?"
        print(f"[{i:3d}] m={m:4d} n={n:5d} k={k:5d} g={g:3d} {qtype:13s} {'B' if on_right else 'A'} I={inner} {str(dtype)[6:]:9s} plan={plan:8s} bad={nb}", flush=True)
        bad += nb
    print("TOTAL BAD", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
