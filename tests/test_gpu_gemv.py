"""GPU suite (-m gpu): w4_gemv_kernel (any4_amd/csrc/w4_gemv.cuh) -- ONE layer per launch with 1 ... 4 activation rows, the
kernel behind a decode step's GEMMs and behind Any4Linear.forward / Int4Linear.forward at batch 1 -- against the CPU oracle.

Tolerances are test_gpu_fast.py's (the kernel computes the same group-scaled sum, regrouped per 32-k step):
    |y - y_gs|  <= 0.5 ulp16(y_gs) (1 + 2^-7) + 4e-6 S          against oracle.linear_group_scaled (double)
    |y - y_ref| <= 0.5 ulp16 (1 + 2^-7) + (4e-6 + eps16) S        against the reference-faithful contraction (eps16 = 2^-9 / 2^-12)
with S = sum_k |x_k w_k|.  Shapes walk every decomposition of the kernel: 8 / 16 / 32 rows per pass, one and several passes,
workgroups with unequal ranges, ragged tiles, k slices that are ragged or empty for some waves, every group size, every LUT kind.
"""
import numpy as np
import pytest
import torch

from tests.conftest import bits16, from_bits16
from tests.test_gpu_fast import QT, gs_reference
from tests.test_gpu_parity import DEV, T, oracle_weights, rand_problem, run_rm, ulp16  # noqa: F401  (T is a fixture)

pytestmark = pytest.mark.gpu


def check(oracle, y_hip, codes, x, qinfo, lut, g, qtype, dtype=torch.bfloat16, rows=None):
    """rows: check only these weight rows (large layers: the first, middle and last rows cover every pass and workgroup kind)."""
    n = codes.shape[0]
    sel = np.arange(n) if rows is None else rows
    tsel = torch.from_numpy(sel)
    c = codes[tsel]
    q = qinfo[:, tsel].contiguous()
    lt = lut if lut is None or lut.dim() == 1 else lut[tsel].contiguous()
    w = from_bits16(oracle_weights(oracle, c, g, qtype, q, lt, dtype), dtype).double()
    x64 = x.double()
    y_ref = (x64 @ w.t()).numpy()
    S = (x64.abs() @ w.abs().t()).numpy()
    y_gs = gs_reference(oracle, c, x, q, lt, g, qtype, dtype)
    got = y_hip.detach().double().cpu().numpy()[:, sel]
    tol = 0.5 * ulp16(y_gs, dtype) * (1 + 2.0 ** -7) + 4e-6 * S + 1e-37
    bad = np.abs(got - y_gs) > tol
    assert not bad.any(), f"vs group-scaled oracle: {bad.sum()} / {bad.size} outside tolerance; worst {np.abs(got - y_gs).max()} at {np.argwhere(bad)[:4]}"
    eps16 = 2.0 ** -9 if dtype == torch.bfloat16 else 2.0 ** -12
    tol_ref = 0.5 * ulp16(y_ref, dtype) * (1 + 2.0 ** -7) + (4e-6 + eps16) * S + 1e-37
    assert not (np.abs(got - y_ref) > tol_ref).any()


def plan(m, n, k, g, qtype, dtype=torch.bfloat16):
    from any4_amd import ops

    return ops.gemm_w4_plan(m, -(-n // 8) * 8, k, g, QT[qtype], True, 4, dtype, 1, "fast")


@pytest.mark.parametrize("case", [
    # (n, k, m, g, qtype)                      what the shape exercises
    (64, 1024, 1, 128, "any4_rowwise"),      # 8 workgroups of one tile: 8 rows per pass, 8 sub-slots
    (40, 512, 3, 64, "any4_rowwise"),        # ragged last tile, one super-tile per wave
    (72, 256, 2, 128, "any4_global"),        # k = 256: waves 4 ... 7 hold no super-tile
    (2048, 1024, 1, 128, "any4_rowwise"),    # one tile on each of 256 workgroups
    (4096, 4096, 1, 128, "any4_rowwise"),    # 16 rows per pass (o_proj of Llama-3-8B)
    (2056, 512, 4, 32, "int4"),              # 257 tiles: ranges of one and two tiles; g = 32 splits a step in two groups
    (6144, 4096, 1, 128, "any4_rowwise"),    # 24 rows on 32-row passes (the fused q/k/v projection)
    (6144, 4096, 2, 256, "any4_global"),
    (4096, 2112, 1, 64, "any4_rowwise"),     # 33 super-tiles: slices of 5, 5, 5, 5, 5, 5, 3, 0
    (4096, 2112, 3, 32, "any4_rowwise"),
    (8192, 2048, 4, 128, "any4_rowwise"),    # 4 tiles per workgroup: a full 32-row pass
    (10240, 1024, 1, 128, "any4_rowwise"),   # 5 tiles: two passes, the second a quarter full; LUT rows staged through LDS
    (28672, 4096, 1, 128, "any4_rowwise"),   # gate_up of Llama-3-8B: 14 tiles per workgroup, four passes
    (28672, 4096, 1, 64, "int4"),
    (4096, 14336, 1, 128, "any4_rowwise"),   # down_proj: 28 super-tiles per wave, 14 steps, ring of 8 with two padding steps
    (4096, 14336, 1, 256, "any4_global"),
    (1024, 8192, 2, 128, "any4_rowwise"),
])
def test_gemv_vs_oracle(T, oracle, case):
    n, k, m, g, qtype = case
    assert plan(m, n, k, g, qtype) == "gemv"
    codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, seed=n + k + m)
    y = run_rm(T, codes, x, qinfo, lut, g, qtype, True, 4)
    assert y.shape[0] == m
    rows = None
    if n > 2048:  # every kind of row: the first and last workgroups, a stretch across workgroup and pass boundaries in between
        rows = np.unique(np.concatenate([np.arange(0, 160), np.arange(n // 2 - 80, n // 2 + 80), np.arange(n - 160, n)]))
    check(oracle, y, codes, x, qinfo, lut, g, qtype, rows=rows)


@pytest.mark.parametrize("m", [1, 2, 3, 4])
@pytest.mark.parametrize("g", [32, 64, 128, 256])
def test_gemv_fp16_and_every_m(T, oracle, m, g):
    n, k = 1032, 1024
    for qtype in ("any4_rowwise", "int4"):
        assert plan(m, n, k, g, qtype, torch.float16) == "gemv"
        codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, dtype=torch.float16, seed=m * g)
        y = run_rm(T, codes, x, qinfo, lut, g, qtype, True, 4)
        check(oracle, y, codes, x, qinfo, lut, g, qtype, dtype=torch.float16)


@pytest.mark.parametrize("case", [
    # (n, k, m, g, qtype)  3 ... 8 rows with a group of >= 128 at k <= 4096: the contraction runs on the matrix core (16-row passes)
    (4096, 4096, 4, 128, "any4_rowwise"),   # two tiles per workgroup, one pass
    (4096, 4096, 8, 128, "any4_rowwise"),   # rows 4 ... 7 come from the second lane quarter
    (6144, 4096, 3, 256, "any4_global"),    # three tiles: the second pass is half empty
    (28672, 4096, 5, 128, "any4_rowwise"),  # gate_up of Llama-3-8B: seven passes, the LUT rows staged through LDS
    (2056, 512, 7, 128, "int4"),            # 257 tiles: ranges of one and two tiles; one super-tile per wave
    (72, 256, 6, 128, "any4_rowwise"),      # k = 256: waves 2 ... 7 hold no step
    (1024, 2048, 8, 256, "int4"), (200, 1024, 3, 128, "any4_rowwise"),
    (4096, 14336, 1, 128, "any4_rowwise"), (512, 8192, 2, 256, "int4"),   # long k: a chunk per thread in the staging, ring of eight
    (8192, 8192, 4, 128, "any4_rowwise"), (1024, 12288, 3, 128, "any4_rowwise"),
])
def test_gemv_matrix_core_contraction_vs_oracle(T, oracle, case):
    n, k, m, g, qtype = case
    for dtype in (torch.bfloat16, torch.float16):
        assert plan(m, n, k, g, qtype, dtype) == "gemv"
        codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, dtype=dtype, seed=n + k + m)
        y = run_rm(T, codes, x, qinfo, lut, g, qtype, True, 4)
        rows = None
        if n > 2048:
            rows = np.unique(np.concatenate([np.arange(0, 160), np.arange(n // 2 - 80, n // 2 + 80), np.arange(n - 160, n)]))
        check(oracle, y, codes, x, qinfo, lut, g, qtype, dtype=dtype, rows=rows)


def test_gemv_matrix_core_row_does_not_depend_on_the_batch(T):
    """The same activation row alone and as row 5 of a 6-row launch (another lane quarter and accumulator register of the matrix
    core), and against the v_dot2 contraction (groups of 64 keep it: same codes and LUT, scales repeated): one output step at most."""
    codes, x, qinfo, lut = rand_problem(4096, 4096, 128, 6, "any4_rowwise", seed=8)
    y6 = run_rm(T, codes, x, qinfo, lut, 128, "any4_rowwise", True, 4)
    y1 = run_rm(T, codes, x[5:6].contiguous(), qinfo, lut, 128, "any4_rowwise", True, 4)
    step = torch.exp2(torch.floor(torch.log2(y1.float().abs().clamp_min(1e-30))) - 7)
    assert ((y6[5:6].float() - y1.float()).abs() <= step).all()
    q64 = qinfo.repeat_interleave(2, dim=0).contiguous()       # groups of 64 with the same scale | zero: the same weights
    assert plan(1, 4096, 4096, 64, "any4_rowwise") == "gemv"
    yd = run_rm(T, codes, x[5:6].contiguous(), q64, lut, 64, "any4_rowwise", True, 4)
    assert ((yd.float() - y1.float()).abs() <= step).all()


def test_gemv_identity_within_one_ulp(T):
    """The reference's identity known-answer test (test_tinygemm_any4.py:14-37): w = eye(k) quantised, LUT = 8 - arange(16),
    scales negated.  Reference numerics reproduce x bit for bit (test_gpu_parity.py); the group-scaled default multiplies by
    15 * bf16(1/15) where the reference's weight rounds to exactly 1.0: within one output ulp."""
    from any4_amd.utils import group_quantize_tensor

    k = 512
    w = torch.eye(k, dtype=torch.bfloat16)
    codes, sz = group_quantize_tensor(w, 4, 128)
    lut = (8 - torch.arange(16)).to(torch.bfloat16)
    sz = sz.clone()
    sz[:, :, 0] *= -1
    x = torch.randn(2, k, generator=torch.Generator().manual_seed(5)).bfloat16()
    y = run_rm(T, codes, x, sz.contiguous(), lut, 128, "any4_global", True, 4).cpu()
    step = torch.exp2(torch.floor(torch.log2(x.float().abs().clamp_min(1e-30))) - 7)
    assert ((y.float() - x.float()).abs() <= step).all()


def test_gemv_row_does_not_depend_on_its_neighbours(T):
    """Deterministic, and a weight row's result depends on that row only: the same rows inside a 4096-row and a 6144-row layer
    (16 and 24 rows per workgroup: different passes, lanes and sub-slots) give the same bits -- every row's sum is added in
    the same order whatever the decomposition."""
    codes, x, qinfo, lut = rand_problem(6144, 2048, 128, 1, "any4_rowwise", seed=3)
    y_big = run_rm(T, codes, x, qinfo, lut, 128, "any4_rowwise", True, 4)
    y_big2 = run_rm(T, codes, x, qinfo, lut, 128, "any4_rowwise", True, 4)
    assert torch.equal(y_big.view(torch.int16), y_big2.view(torch.int16))
    y_small = run_rm(T, codes[:4096], x, qinfo[:, :4096].contiguous(), lut[:4096].contiguous(), 128, "any4_rowwise", True, 4)
    d = (y_big[:, :4096].float() - y_small.float()).abs()
    step = torch.exp2(torch.floor(torch.log2(y_small.float().abs().clamp_min(1e-30))) - 7)
    assert (d <= step).all()  # (the order of a row's partial sums differs between 16- and 32-row passes: at most the final rounding)
