import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from any4_amd import decode_ops as G
DEV = "cuda:0"
gen = torch.Generator(device=DEV).manual_seed(5)
for dtype in (torch.bfloat16, torch.float16):
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for (m, n, k) in [(1, 128256, 4096), (3, 1000, 2048), (4, 32000, 4096), (2, 5003, 8192), (1, 7, 4096)]:
        x = torch.randn(m, k, device=DEV, generator=gen).to(dtype)
        w = (torch.randn(n, k, device=DEV, generator=gen) * 0.02).to(dtype)
        y = G.linear16(x, w)
        want = x.double() @ w.double().t()
        S = x.double().abs() @ w.double().abs().t()
        err = (y.double() - want).abs()
        tol = 0.5 * ulp * want.abs() * 1.02 + 8e-6 * S
        bad = err > tol
        print(dtype, m, n, k, "bad", int(bad.sum()), "max err/S", float((err / S).max()), "max (err - halfulp)/S", float(((err - 0.5 * ulp * want.abs()) / S).max()),
              "first bad", bad.nonzero()[:3].tolist())
