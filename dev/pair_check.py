"""Developer GPU check of the pair-table kernel (run on the GPU box): correctness of both numerics against f64
references on a sweep of shapes, then a timing.  TG_PAIR must be set in the environment (1 group-scaled, 2 exact)."""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tinygemm  # noqa
from tests.conftest import bits16, from_bits16
from oracle import oracle as orc

T = torch.ops.tinygemm
DEV = "cuda:0"
mode = int(os.environ.get("TG_PAIR", "0"))


def problem(n, k, g, m, qtype, seed=0, dtype=torch.bfloat16):
    gen = torch.Generator().manual_seed(seed)
    codes = torch.randint(0, 16, (n, k), dtype=torch.int32, generator=gen)
    x = torch.randn(m, k, generator=gen).to(dtype)
    if qtype == "mx4":
        qinfo = torch.randint(120, 131, (n, k // g), dtype=torch.uint8, generator=gen)
        lut = None
    else:
        scales = (torch.rand(k // g, n, generator=gen) * 0.02 + 0.005).to(dtype)
        zeros = (torch.randn(k // g, n, generator=gen) * 0.01).to(dtype)
        qinfo = torch.stack([scales, zeros], dim=2).contiguous()
        lut = {"int4": None, "any4_global": torch.randn(16, generator=gen).to(dtype), "any4_rowwise": torch.randn(n, 16, generator=gen).to(dtype)}[qtype]
    return codes, x, qinfo, lut


def refs(codes, x, qinfo, lut, g, qtype, dtype):
    q = {"int4": orc.Q_INT4, "any4_global": orc.Q_ANY4_GLOBAL, "any4_rowwise": orc.Q_ANY4_ROWWISE, "mx4": orc.Q_MX4}[qtype]
    qi = qinfo.numpy() if qtype == "mx4" else bits16(qinfo)
    wb = orc.dequant(codes.numpy(), g, q, qi, None if lut is None else bits16(lut), orc.BF16 if dtype == torch.bfloat16 else orc.F16)
    w_exact = from_bits16(wb, dtype).double()
    n, k = codes.shape
    if qtype == "mx4":
        fp4 = torch.tensor([0, .5, 1, 1.5, 2, 3, 4, 6, -0., -.5, -1, -1.5, -2, -3, -4, -6], dtype=torch.float64)
        L = fp4[codes.long()]
        s = torch.exp2(qinfo.double() - 127).repeat_interleave(g, 1)
        w_gs = L * s
    else:
        if qtype == "int4":
            L = (codes.double() - 8)
        elif qtype == "any4_global":
            L = lut.double()[codes.long()]
        else:
            L = torch.gather(lut.double(), 1, codes.long())
        s = qinfo[:, :, 0].double().t().repeat_interleave(g, 1)
        z = qinfo[:, :, 1].double().t().repeat_interleave(g, 1)
        w_gs = L * s + z
    x64 = x.double()
    return x64 @ w_exact.t(), x64 @ w_gs.t(), (x64.abs() @ w_exact.abs().t())


def run(codes, x, qinfo, lut, g, qtype, inner):
    d = lambda t: None if t is None else t.to(DEV)
    w2 = T.convert_matrix_to_m16n8k16_Bint4_layout(d(codes), inner)
    if qtype == "mx4":
        return T.tinygemm_y_f16RM_x_f16RM_w_mx4TC(d(x), w2, g, d(qinfo), True)
    if qtype == "int4":
        return T.tinygemm_y_f16RM_x_f16RM_w_int4TC(d(x), w2, g, d(qinfo), True)
    return T.tinygemm_y_f16RM_x_f16RM_w_any4TC(d(x), w2, g, d(qinfo), d(lut), True)


def ulp16(y64, dtype):
    e = np.floor(np.log2(np.maximum(np.abs(y64), 1e-300)))
    return np.exp2(e - 7) if dtype == torch.bfloat16 else np.exp2(np.maximum(e, -14) - 10)


fails = 0
cases = []
for qtype in ("any4_rowwise", "int4", "any4_global", "mx4"):
    for (n, k, g, m, inner) in [(64, 512, 128, 1, 4), (40, 256, 32, 2, 4), (64, 512, 64, 1, 8), (32, 256, 128, 3, 2), (64, 1024, 256, 1, 4),
                                (4096, 4096, 128, 1, 4), (200, 2048, 128, 5, 4), (72, 4096, 128, 2, 8), (128, 1024, 64, 1, 2), (96, 2048, 32, 1, 8)]:
        if qtype == "mx4":
            g = 32
        cases.append((qtype, n, k, g, m, inner, torch.bfloat16))
cases.append(("any4_rowwise", 128, 1024, 128, 1, 4, torch.float16))
cases.append(("int4", 64, 512, 64, 2, 4, torch.float16))
for (qtype, n, k, g, m, inner, dtype) in cases:
    if k % (16 * inner):
        continue
    codes, x, qinfo, lut = problem(n, k, g, m, qtype, seed=n + k + m, dtype=dtype)
    y = run(codes, x, qinfo, lut, g, qtype, inner).double().cpu().numpy()
    y_ex, y_gs, S = [t.numpy() for t in refs(codes, x, qinfo, lut, g, qtype, dtype)]
    ref = y_ex if mode == 2 else y_gs
    tol = 0.5 * ulp16(ref, dtype) * (1 + 2.0 ** -7) + 4e-6 * S + 1e-37
    err = np.abs(y - ref)
    bad = int((err > tol).sum())
    d_other = np.abs(y - (y_gs if mode == 2 else y_ex)).max()
    print(f"{qtype:13s} n={n:5d} k={k:5d} g={g:3d} m={m} I={inner} {str(dtype)[6:]:8s} max|y|={np.abs(ref).max():7.3f} max err={err.max():.3e} "
          f"bad={bad}/{err.size}  (vs other numerics: {d_other:.3e})")
    fails += bad > 0
print("FAILS", fails)
