"""GPU suite (-m gpu): the library's DEFAULT numerics (TG_NUM_FAST, include/tinygemm_hip.h) -- the pair-table kernel of
any4_amd/csrc/w4_gemm_pair.cuh, which applies scale / zero per quantisation group to the f32 accumulator -- plus what
rides on the same ABI revision: the bias fused into the output store, re-entrancy across host threads, and parity of the
exact launches bench.py times.

Tolerances:
  * against the oracle's group-scaled restatement (oracle.linear_group_scaled, same math in double):
        |y - y64| <= 0.5 ulp16(y64) (1 + 2^-7) + 4e-6 S,   S = sum_k |x_k w_k|
    i.e. the final rounding plus a bound on f32 accumulation error, the same budget as the reference-numerics tests.
  * against the oracle's reference-faithful contraction (oracle.dequant + exact contraction): the fast result may differ
    by the reference's own rounding of every dequantised weight to 16 bits, |dw_k| <= 2^-9 |w_k| (2^-12 for fp16):
        |y - y64_ref| <= 0.5 ulp16 (1 + 2^-7) + 4e-6 S + eps16 S      (worst case; observed ~0.1 of it)
  * north_star: max-abs <= 1e-2 against the reference's own CPU dequant-matmul on the captured fixture (max|y| = 2.2).
"""
import ctypes
import threading

import numpy as np
import pytest
import torch

from tests.conftest import bits16, from_bits16, load_golden
from tests.test_gpu_parity import DEV, T, oracle_weights, rand_problem, run_rm, ulp16  # noqa: F401  (T is a fixture)

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("reference_weight_format")]  # (Aint4 tests here drive the C ABI with the reference's words)

QT = {"int4": 0, "any4_global": 1, "any4_rowwise": 2, "mx4": 3}


def gs_reference(oracle, codes, x, qinfo, lut, g, qtype, dtype=torch.bfloat16):
    q = {"int4": oracle.Q_INT4, "any4_global": oracle.Q_ANY4_GLOBAL, "any4_rowwise": oracle.Q_ANY4_ROWWISE, "mx4": oracle.Q_MX4}[qtype]
    qi = qinfo.numpy() if qtype == "mx4" else bits16(qinfo)
    dt = oracle.BF16 if dtype == torch.bfloat16 else oracle.F16
    _, y32 = oracle.linear_group_scaled(bits16(x), codes.numpy(), g, q, qi, None if lut is None else bits16(lut), dt)
    return y32.astype(np.float64)


def assert_fast_close(oracle, y_hip, codes, x, qinfo, lut, g, qtype, dtype=torch.bfloat16, inner=4, batch=1, expect_pair=True,
                      on_right=True):
    """Both tolerances of the module docstring.  The tight comparison with the group-scaled oracle applies when the
    library says (tg_gemm_w4_plan) that this problem runs the pair-table kernel, which `expect_pair` demands."""
    from any4_amd import ops

    pad = 8 if on_right else 16
    plan = ops.gemm_w4_plan(x.shape[0], -(-codes.shape[0] // pad) * pad, x.shape[1], g, QT[qtype], on_right, inner, dtype, batch, "fast")
    if expect_pair is not None:  # (None: whichever family the library routes this shape to -- the tolerance follows the plan)
        assert (plan in ("pair", "gemv")) == expect_pair, f"kernel plan {plan!r}, expected {'pair / gemv' if expect_pair else 'a reference kernel'}"
    w = from_bits16(oracle_weights(oracle, codes, g, qtype, qinfo, lut, dtype), dtype).double()
    x64 = x.double()
    y_ref = (x64 @ w.t()).numpy()
    S = (x64.abs() @ w.abs().t()).numpy()
    y_gs = gs_reference(oracle, codes, x, qinfo, lut, g, qtype, dtype)
    got = y_hip.detach().double().cpu().numpy()[:, :codes.shape[0]]
    if plan in ("pair", "gemv"):
        tol = 0.5 * ulp16(y_gs, dtype) * (1 + 2.0 ** -7) + 4e-6 * S + 1e-37
        bad = np.abs(got - y_gs) > tol
        assert not bad.any(), f"vs group-scaled oracle: {bad.sum()} / {bad.size} outside tolerance; worst {np.abs(got - y_gs).max()}"
    eps16 = 2.0 ** -9 if dtype == torch.bfloat16 else 2.0 ** -12
    tol_ref = 0.5 * ulp16(y_ref, dtype) * (1 + 2.0 ** -7) + (4e-6 + eps16) * S + 1e-37
    bad = np.abs(got - y_ref) > tol_ref
    assert not bad.any(), f"vs reference-faithful oracle: {bad.sum()} / {bad.size} outside tolerance"


def assert_north_star(oracle, y_hip, codes, x, qinfo, lut, g, qtype):
    """north_star's contract, RAW, as bench.py's check_layers enforces it on every leg of the driver line: with the activations
    calibrated (bench.calibrate_x: a power of two, exact in bf16) so that max|y| < 2 -- the order of the captured fixture's 2.2
    (SURVEY.md 8c/8d), below the binade where one bf16 step alone is 1.6e-2 -- the bf16 output is within 1e-2 max-abs of the
    reference-faithful result (oracle.linear: every weight rounded to bf16 with one fma, fp32 contraction, bf16 output --
    MatrixLayoutB.cuh:1042-1046), and so is the regrouped arithmetic before the output rounding.  No scaled / one-step clause."""
    q = {"int4": oracle.Q_INT4, "any4_global": oracle.Q_ANY4_GLOBAL, "any4_rowwise": oracle.Q_ANY4_ROWWISE, "mx4": oracle.Q_MX4}[qtype]
    qi = qinfo.numpy() if qtype == "mx4" else bits16(qinfo)
    lb = None if lut is None else bits16(lut)
    r16, r32 = oracle.linear(bits16(x), codes.numpy(), g, q, qi, lb)
    _, g32 = oracle.linear_group_scaled(bits16(x), codes.numpy(), g, q, qi, lb)
    ref = from_bits16(r16, torch.bfloat16).double().numpy()
    got = y_hip.detach().double().cpu().numpy()[:, :codes.shape[0]]
    ymax = float(np.abs(ref).max())
    assert ymax < 2.0, f"{qtype}: activations not calibrated (max|y| = {ymax:.3f}): use _stacked_launch(..., calibrate=True)"
    formula = float(np.abs(g32.astype(np.float64) - r32.astype(np.float64)).max())
    assert formula <= 1e-2, f"{qtype}: arithmetic {formula:.3e} from the reference's at max|y| = {ymax:.3f}"
    err = float(np.abs(got - ref).max())
    assert err <= 1e-2, f"{qtype}: {err:.3e} from the reference-faithful result at max|y| = {ymax:.3f}"
    return err, ymax


def run_fast(T, codes, x, qinfo, lut, g, qtype, inner, bias=None, min_items=384, workspace=True, on_right=True):
    """The persistent pair-table kernel is dispatched when a launch has >= 192 work items (64-row blocks x problems; tested here
    with >= 384; smaller launches are latency-bound and go to w4_gemm_pair16_kernel): run `copies` identical problems in ONE
    tg_gemm_w4 call (the C ABI's stacked launch), check that every copy gives the same bits, return (y of copy 0, copies)."""
    from any4_amd import _lib

    L = _lib.load()
    n, k = codes.shape
    m = x.shape[0]
    dt = x.dtype
    if on_right:
        packed1 = T.convert_matrix_to_m16n8k16_Bint4_layout(codes.to(DEV), inner)
        wrows = packed1.shape[0] * 8
        copies = -(-min_items // (-(-wrows // 64) * -(-m // (8 if m <= 8 else 16 if m <= 16 else 32))))
    else:  # Aint4: 32-row work items, one pass of up to 8 activation rows
        packed1 = T.convert_matrix_to_m16n8k16_Aint4_layout(codes.to(DEV), inner)
        wrows = packed1.shape[0] * 16
        copies = -(-min_items // -(-wrows // 32))
    if qtype == "mx4" and qinfo.shape[0] < wrows:  # exponent rows cover the tile padding
        qinfo = torch.cat([qinfo, torch.full((wrows - qinfo.shape[0], qinfo.shape[1]), 127, dtype=qinfo.dtype)])
    elif qtype != "mx4" and qinfo.shape[1] < wrows:
        qinfo = torch.cat([qinfo, torch.zeros(qinfo.shape[0], wrows - qinfo.shape[1], 2, dtype=qinfo.dtype)], dim=1)
    if lut is not None and lut.dim() == 2 and lut.shape[0] < wrows:
        lut = torch.cat([lut, torch.zeros(wrows - lut.shape[0], 16, dtype=lut.dtype)])
    rep = lambda t: None if t is None else t.to(DEV).unsqueeze(0).repeat(copies, *([1] * t.dim())).contiguous()
    packed, xs, qs, luts = rep(packed1.cpu()), rep(x), rep(qinfo), rep(lut)
    bs = rep(bias)
    ys = torch.full((copies, m, wrows), float("nan"), dtype=dt, device=DEV)
    args = _lib.W4Gemm(x=xs.data_ptr(), w=packed.data_ptr(), qinfo=qs.data_ptr(), lut=(luts.data_ptr() if luts is not None else None),
                       y=ys.data_ptr(), m=m, wrows=wrows, k=k, group=g, qtype=QT[qtype],
                       dtype=_lib.TG_BF16 if dt == torch.bfloat16 else _lib.TG_F16, w_on_right=1 if on_right else 0, inner_k_tiles=inner, batch=copies,
                       stride_x=xs.stride(0) * 2, stride_w=packed.stride(0) * 4, stride_qinfo=qs.stride(0) * qs.element_size(),
                       stride_lut=(luts.stride(0) * 2 if luts is not None else 0), stride_y=ys.stride(0) * 2,
                       numerics=_lib.TG_NUM_FAST, bias=(bs.data_ptr() if bs is not None else None),
                       stride_bias=(bs.stride(0) * 2 if bs is not None else 0))
    if workspace:
        need = L.tg_gemm_w4_workspace_bytes(ctypes.byref(args))
        assert need >= 0
        if need:
            ws = torch.empty(need + 64, dtype=torch.uint8, device=DEV)
            ws.fill_(0xff)  # NaN bit patterns: nothing may be read that the call did not write first
            args.workspace, args.workspace_bytes = ws.data_ptr(), need
    _lib.check(L.tg_gemm_w4(ctypes.byref(args), 0, torch.cuda.current_stream().cuda_stream), "stacked fast launch")
    torch.cuda.synchronize()
    assert not torch.isnan(ys.float()).any() or qtype == "mx4"
    assert torch.equal(ys[0].view(torch.int16), ys[-1].view(torch.int16)) and torch.equal(ys[0].view(torch.int16), ys[copies // 2].view(torch.int16))
    return ys[0], copies


def test_default_numerics_is_fast():
    import any4_amd

    assert any4_amd.get_numerics() == "fast"
    with any4_amd.numerics("reference"):
        assert any4_amd.get_numerics() == "reference"
    assert any4_amd.get_numerics() == "fast"


@pytest.mark.parametrize("qtype", ["any4_rowwise", "any4_global", "int4", "mx4"])
@pytest.mark.parametrize("inner", [2, 4, 8])
@pytest.mark.parametrize("g", [32, 64, 128, 256])
def test_pair_kernel_vs_oracle(T, oracle, qtype, inner, g):
    if qtype == "mx4":
        g = 32
    for (n, k, m) in [(64, 1024, 1), (40, 512, 3), (136, 2048, 1), (200, 4096, 1)]:
        if k % (16 * inner) or k % g:
            continue
        codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, seed=n + k + inner)
        y, copies = run_fast(T, codes, x, qinfo, lut, g, qtype, inner)
        # innerKTiles 2 / 4: always a pair-table kernel; innerKTiles 8: only the m = 1 kernels of int4 / any4 at g >= 128 are
        # instantiated (the others compiled with 70 ... 1100 bytes of scratch per lane and were dropped in round 4)
        assert_fast_close(oracle, y, codes, x, qinfo, lut, g, qtype, inner=inner, batch=copies, expect_pair=True if inner < 8 else None)


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 7, 8])
def test_pair_kernel_m_sweep(T, oracle, m):
    """m = 1..8 on one accumulator register set (rows 0-3 in lane half 0, 4-7 in half 1); k small enough for the
    activation block to stay on chip at every m."""
    n, k, g = 72, 256, 128
    codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_rowwise", seed=m)
    y, copies = run_fast(T, codes, x, qinfo, lut, g, "any4_rowwise", 4)
    assert_fast_close(oracle, y, codes, x, qinfo, lut, g, "any4_rowwise", batch=copies)


@pytest.mark.parametrize("m", [9, 16, 17, 33])
def test_pair_kernel_many_rows(T, oracle, m):
    """m > 8: wider accumulator sets and several activation passes; whether the block still fits the on-chip stage is the
    library's call (the plan decides which tolerance applies)."""
    from any4_amd import ops

    n, k, g = 64, 128, 64
    codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_rowwise", seed=m)
    y, copies = run_fast(T, codes, x, qinfo, lut, g, "any4_rowwise", 4)
    pair = ops.gemm_w4_plan(m, n, k, g, QT["any4_rowwise"], batch=copies) == "pair"
    assert_fast_close(oracle, y, codes, x, qinfo, lut, g, "any4_rowwise", batch=copies, expect_pair=pair)


@pytest.mark.parametrize("case", [
    # (n, k, m, g, inner, qtype): 9 ... 16 activation rows at a k where the block does not fit on chip -> the 16x16x32 kernel for
    # Bint4 weights (launch_pair_b16): every inner-k, group, quantisation type, ragged row tiles, all m of the range
    (64, 4096, 16, 128, 4, "any4_rowwise"), (72, 4096, 9, 128, 4, "any4_rowwise"), (40, 4096, 12, 64, 4, "any4_global"),
    (64, 4096, 16, 32, 4, "mx4"), (64, 4096, 13, 32, 4, "int4"), (96, 2048, 16, 128, 2, "any4_rowwise"), (64, 4096, 10, 256, 8, "int4"),
    (64, 8192, 16, 128, 8, "any4_rowwise"), (136, 4096, 11, 64, 2, "int4"), (64, 14336, 16, 128, 4, "any4_rowwise"),
    (64, 4096, 15, 32, 8, "any4_rowwise"),
])
def test_pair_kernel_9_to_16_rows(T, oracle, case):
    """m = 9 ... 16 with the weights on the right (the reference's full 16-row tile, TinyGemmImpl.cuh:53-54) on the pair-table
    kernel: `tg_gemm_w4_plan` must say pair, results against both oracles."""
    from any4_amd import ops

    n, k, m, g, inner, qtype = case
    codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, seed=n + k + m)
    y, copies = run_fast(T, codes, x, qinfo, lut, g, qtype, inner, min_items=768)
    if inner < 8:  # (innerKTiles 8 on the 16x16x32 tiles is not instantiated: scratch; those shapes take the reference-numerics kernels)
        assert ops.gemm_w4_plan(m, -(-n // 8) * 8, k, g, QT[qtype], True, inner, batch=copies) == "pair"
    assert_fast_close(oracle, y, codes, x, qinfo, lut, g, qtype, inner=inner, batch=copies, expect_pair=True if inner < 8 else None)


def _run_xr(T, codes, x, qinfo, lut, g, qtype, copies, bias=None, residual=False, tc=False, plan=None):
    """One stacked tg_gemm_w4 launch over `copies` problems with the SAME weights and DIFFERENT activations (problem j's rows are
    x rolled by j along the batch: the register-resident activations of w4_gemm_xr_kernel must follow the problem); optional
    fused bias / residual (bias_row_stride = wrows) and A-fragment-order activations / outputs.  Returns y [copies][m][n]."""
    from any4_amd import _lib

    L = _lib.load()
    n, k = codes.shape
    m = x.shape[0]
    dt = x.dtype
    packed1 = T.convert_matrix_to_m16n8k16_Bint4_layout(codes.to(DEV), 4)
    rep = lambda t: None if t is None else t.to(DEV).unsqueeze(0).repeat(copies, *([1] * t.dim())).contiguous()
    packed, qs, luts = rep(packed1.cpu()), rep(qinfo), rep(lut)
    xs = torch.stack([torch.roll(x, j % m, 0) for j in range(copies)]).to(DEV).contiguous()
    xin = xs
    if tc:
        xin = torch.stack([T.convert_matrix_to_m16n8k16_A_layout(xs[j], 1) for j in range(copies)]).contiguous()
    bs = None
    if bias is not None:
        bs = rep(bias)
    ys = torch.full((copies, m, n), float("nan"), dtype=dt, device=DEV)
    yout = ys
    if tc:
        yout = torch.stack([T.convert_matrix_to_m16n8k16_A_layout(ys[j], 1) for j in range(copies)]).contiguous()
    args = _lib.W4Gemm(x=xin.data_ptr(), w=packed.data_ptr(), qinfo=qs.data_ptr(), lut=(luts.data_ptr() if luts is not None else None),
                       y=yout.data_ptr(), m=m, wrows=n, k=k, group=g, qtype=QT[qtype],
                       dtype=_lib.TG_BF16 if dt == torch.bfloat16 else _lib.TG_F16, w_on_right=1, inner_k_tiles=4, batch=copies,
                       stride_x=xin.stride(0) * 2, stride_w=packed.stride(0) * 4, stride_qinfo=qs.stride(0) * qs.element_size(),
                       stride_lut=(luts.stride(0) * 2 if luts is not None else 0), stride_y=yout.stride(0) * 2,
                       numerics=_lib.TG_NUM_FAST, bias=(bs.data_ptr() if bs is not None else None),
                       stride_bias=(bs.stride(0) * 2 if bs is not None else 0), bias_row_stride=(n if residual else 0),
                       x_layout=1 if tc else 0, y_layout=1 if tc else 0)
    need = L.tg_gemm_w4_workspace_bytes(ctypes.byref(args))
    if k == 4096 or m <= 8:
        assert need == 0  # the xr kernel arranges the activations itself: no scratch
    else:                 # 9 ... 16 rows beyond k = 4096: k-windows, f32 partial sums of every window in the caller's workspace
        assert need == (2 if k == 8192 else 3) * copies * m * n * 4
        ws = torch.full((need,), 0xff, dtype=torch.uint8, device=DEV)
        args.workspace, args.workspace_bytes = ws.data_ptr(), need
    assert L.tg_gemm_w4_plan(ctypes.byref(args), 0) == (_lib.TG_PLAN_PAIR_XR if plan is None else plan)
    _lib.check(L.tg_gemm_w4(ctypes.byref(args), 0, torch.cuda.current_stream().cuda_stream), "stacked launch (xr)")
    torch.cuda.synchronize()
    if tc:
        ys = torch.stack([T.convert_matrix_from_m16n8k16_A_layout(yout[j], m, n) for j in range(copies)])
    return xs.cpu(), ys


@pytest.mark.parametrize("case", [
    # (n, m, g, qtype, dtype): w4_gemm_xr_kernel -- Bint4 weights, k = 4096, innerKTiles 4, 2 ... 16 activation rows, all three
    # instantiated group sizes (64, 128, 256), every quantisation type with a 16-bit LUT, both dtypes; n = 64: EVERY work item starts a new
    # problem, n = 192: every third one
    (64, 16, 128, "any4_rowwise", torch.bfloat16), (64, 9, 64, "any4_global", torch.bfloat16), (64, 2, 256, "int4", torch.bfloat16),
    (192, 5, 128, "any4_rowwise", torch.bfloat16), (64, 13, 256, "any4_rowwise", torch.bfloat16), (64, 8, 128, "int4", torch.bfloat16),
    (128, 16, 128, "any4_rowwise", torch.float16), (64, 3, 64, "int4", torch.float16), (64, 12, 128, "any4_global", torch.bfloat16),
    # g = 32: two groups per 64-k super-tile
    (64, 8, 32, "int4", torch.bfloat16), (128, 16, 32, "any4_rowwise", torch.bfloat16), (64, 11, 32, "any4_global", torch.float16),
    # mx4 (g = 32): weights converted in registers, exponent blocks per row, no table
    (64, 16, 32, "mx4", torch.bfloat16), (128, 5, 32, "mx4", torch.bfloat16), (64, 2, 32, "mx4", torch.bfloat16),
    # k = 8192 (9 ... 16 rows): 32 chunks of activations per lane, two super-tiles in flight
    (64, 16, 128, "any4_rowwise", torch.bfloat16, 8192), (64, 9, 256, "int4", torch.float16, 8192),
])
def test_xr_kernel_vs_oracle(T, oracle, case):
    """The register-resident-activation kernel against both oracles, problem by problem (each problem of the stacked launch has
    its own activations), with workgroups that change problem every item / every third item."""
    n, m, g, qtype, dtype = case[:5]
    k = case[5] if len(case) > 5 else 4096
    codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, dtype=dtype, seed=n + m + g)
    copies = 512 * 64 // n + 5
    xs, ys = _run_xr(T, codes, x, qinfo, lut, g, qtype, copies)
    assert not torch.isnan(ys.float()).any()
    for j in (0, 1, copies // 2, copies - 1):
        assert_fast_close(oracle, ys[j], codes, xs[j], qinfo, lut, g, qtype, dtype=dtype, batch=copies)
    # problems with the same activations (j and j + m: the same roll) give the same bits whichever workgroup computed them
    assert torch.equal(ys[0].view(torch.int16), ys[m].view(torch.int16))
    assert torch.equal(ys[1].view(torch.int16), ys[copies - 1 - (copies - 2) % m].view(torch.int16))


@pytest.mark.parametrize("case", [(16384, 16, 128, "any4_rowwise", torch.bfloat16), (16384, 9, 64, "int4", torch.float16),
                                  (28672, 13, 128, "any4_rowwise", torch.bfloat16), (5120, 16, 128, "any4_rowwise", torch.bfloat16),
                                  (11008, 10, 256, "any4_global", torch.bfloat16), (16384, 5, 64, "any4_rowwise", torch.bfloat16),
                                  (16384, 8, 32, "int4", torch.bfloat16)])
def test_xr_kernel_single_large_layer(T, oracle, case):
    """ONE layer per launch at k = 4096 with more 16-row tiles than CUs, 9 ... 16 rows and the 5 ... 8-row launches w4_gemv_kernel declines
    (groups of 32 / 64): w4_gemm_pair16_loop_kernel for row-major operands (up to 8 tiles per CU) -- and, with fragment-order operands
    (m = 16), the xr kernel with one workgroup per 64-row item (tg_xr.hip, `single`: fewer workgroups than CUs at 5120 / 11008 rows, one
    each, one or two at 28672; the activations of the launch's first problem staged through LDS, w4_gemm_xr.cuh XLDS)."""
    from any4_amd import _lib

    n, m, g, qtype, dtype = case
    codes, x, qinfo, lut = rand_problem(n, 4096, g, m, qtype, dtype=dtype, seed=n + m)
    xs, ys = _run_xr(T, codes, x, qinfo, lut, g, qtype, 1, plan=_lib.TG_PLAN_PAIR)
    if m == 16 and qtype != "mx4":
        _, yt = _run_xr(T, codes, x, qinfo, lut, g, qtype, 1, tc=True)   # (asserts the xr plan)
        assert_fast_close(oracle, yt[0], codes, xs[0], qinfo, lut, g, qtype, dtype=dtype, batch=1)
    assert not torch.isnan(ys.float()).any()
    assert_fast_close(oracle, ys[0], codes, xs[0], qinfo, lut, g, qtype, dtype=dtype, batch=1)


def test_xr_kernel_single_large_layer_bias_and_fragment_order(T, oracle):
    """One 6144-row layer at 16 rows with a fused residual (the plain launch's bits plus a separate rounded add) and with fragment-order
    activations / outputs (the xr kernel's single-layer route, fewer workgroups than CUs: 96 items; x_layout = TG_LAYOUT_TC_A takes
    x_prepare's gather path instead of the LDS staging)."""
    from any4_amd import _lib

    n, m, g, k = 6144, 16, 128, 4096
    codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_rowwise", seed=321)
    xs, y0 = _run_xr(T, codes, x, qinfo, lut, g, "any4_rowwise", 1, plan=_lib.TG_PLAN_PAIR)   # (row-major: w4_gemm_pair16_loop_kernel)
    assert_fast_close(oracle, y0[0], codes, xs[0], qinfo, lut, g, "any4_rowwise", batch=1)
    res = torch.randn(m, n, generator=torch.Generator().manual_seed(6)).bfloat16()
    _, y1 = _run_xr(T, codes, x, qinfo, lut, g, "any4_rowwise", 1, bias=res, residual=True, plan=_lib.TG_PLAN_PAIR)
    assert torch.equal((y0.float() + res.to(DEV).float()).bfloat16().view(torch.int16), y1.view(torch.int16))
    _, y2 = _run_xr(T, codes, x, qinfo, lut, g, "any4_rowwise", 1, tc=True)                   # (fragment order: the xr kernel, one item per workgroup)
    assert_fast_close(oracle, y2[0], codes, xs[0], qinfo, lut, g, "any4_rowwise", batch=1)


def test_xr_kernel_mx4_nan_exponent(T, oracle):
    """e = 255 is NaN (Dequantization.cuh:331-339): one such group makes its weight row's outputs NaN -- for every activation row and
    every problem -- and nothing else; the other outputs are bit-equal to the run without it."""
    n, m, g, k = 64, 9, 32, 4096
    codes, x, qinfo, lut = rand_problem(n, k, g, m, "mx4", seed=9)
    copies = 520
    _, y0 = _run_xr(T, codes, x, qinfo, lut, g, "mx4", copies)
    q2 = qinfo.clone()
    q2[5, 77] = 255
    _, y1 = _run_xr(T, codes, x, q2, lut, g, "mx4", copies)
    assert torch.isnan(y1[:, :, 5].float()).all()
    keep = [c for c in range(n) if c != 5]
    assert torch.equal(y1[:, :, keep].view(torch.int16), y0[:, :, keep].view(torch.int16))


def test_xr_kernel_bias_and_residual(T, oracle):
    """Fused bias and fused residual (one bias row per activation row) on the xr kernel: bit-equal to the plain launch's result
    plus a separate rounded add."""
    n, m, g, k = 128, 11, 128, 4096
    codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_rowwise", seed=77)
    copies = 300
    xs, y0 = _run_xr(T, codes, x, qinfo, lut, g, "any4_rowwise", copies)
    gen = torch.Generator().manual_seed(5)
    bias = torch.randn(n, generator=gen).bfloat16()
    _, y1 = _run_xr(T, codes, x, qinfo, lut, g, "any4_rowwise", copies, bias=bias)
    assert torch.equal((y0.float() + bias.to(DEV).float()).bfloat16().view(torch.int16), y1.view(torch.int16))
    res = torch.randn(m, n, generator=gen).bfloat16()
    _, y2 = _run_xr(T, codes, x, qinfo, lut, g, "any4_rowwise", copies, bias=res, residual=True)
    assert torch.equal((y0.float() + res.to(DEV).float()).bfloat16().view(torch.int16), y2.view(torch.int16))


def test_xr_kernel_fragment_order(T, oracle):
    """A-fragment-order activations and outputs (x_layout = y_layout = TG_LAYOUT_TC_A; m a multiple of 16) on the xr kernel: the
    row-major launch's bits, re-laid out."""
    n, m, g, k = 128, 16, 128, 4096
    codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_rowwise", seed=78)
    copies = 300
    xs, y_tc = _run_xr(T, codes, x, qinfo, lut, g, "any4_rowwise", copies, tc=True)
    _, y_rm = _run_xr(T, codes, x, qinfo, lut, g, "any4_rowwise", copies)
    assert torch.equal(y_tc.view(torch.int16), y_rm.view(torch.int16))
    assert_fast_close(oracle, y_tc[7], codes, xs[7], qinfo, lut, g, "any4_rowwise", batch=copies)


@pytest.mark.parametrize("case", [
    # (n, k, m, g, qtype, copies): more than 16 activation rows = ceil(m / 16) launches of up to 16 rows each (tinygemm_hip.hip, row blocks)
    (4096, 4096, 17, 128, "any4_rowwise", 12),   # stacked: xr for the 16-row block, the m = 1 kernel for the ragged block
    (4096, 4096, 40, 64, "int4", 10),            # three blocks (16 + 16 + 8)
    (256, 4096, 33, 128, "any4_rowwise", 1),     # one layer per launch: pair16 twice, the gemv kernel for the last row
    (4096, 8192, 25, 128, "any4_global", 10),    # k-windows (f32 partial sums in the workspace, reused by the blocks) + packed rows
    (64, 4096, 64, 32, "mx4", 40),               # mx4, four full blocks
])
def test_more_than_16_rows_in_row_blocks(T, oracle, case):
    """Default numerics with more than 16 activation rows: every row within the group-scaled tolerance, every problem of a stacked
    launch, a fused per-row residual bit-identical to the separate add."""
    from any4_amd import _lib

    L = _lib.load()
    n, k, m, g, qtype, copies = case
    codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, seed=n + k + m)
    packed1 = T.convert_matrix_to_m16n8k16_Bint4_layout(codes.to(DEV), 4)
    rep = lambda t: None if t is None else t.to(DEV).unsqueeze(0).repeat(copies, *([1] * t.dim())).contiguous()
    packed, qs, luts = rep(packed1.cpu()), rep(qinfo), rep(lut)
    xs = torch.stack([torch.roll(x, j % m, 0) for j in range(copies)]).to(DEV).contiguous()
    res = torch.randn(copies, m, n, generator=torch.Generator().manual_seed(3)).bfloat16().to(DEV)

    def run(bias):
        ys = torch.full((copies, m, n), float("nan"), dtype=torch.bfloat16, device=DEV)
        args = _lib.W4Gemm(x=xs.data_ptr(), w=packed.data_ptr(), qinfo=qs.data_ptr(), lut=(luts.data_ptr() if luts is not None else None),
                           y=ys.data_ptr(), m=m, wrows=n, k=k, group=g, qtype=QT[qtype], dtype=_lib.TG_BF16, w_on_right=1, inner_k_tiles=4,
                           batch=copies, stride_x=xs.stride(0) * 2, stride_w=packed.stride(0) * 4, stride_qinfo=qs.stride(0) * qs.element_size(),
                           stride_lut=(luts.stride(0) * 2 if luts is not None else 0), stride_y=ys.stride(0) * 2, numerics=_lib.TG_NUM_FAST,
                           bias=(bias.data_ptr() if bias is not None else None), stride_bias=(bias.stride(0) * 2 if bias is not None else 0),
                           bias_row_stride=(n if bias is not None else 0))
        need = L.tg_gemm_w4_workspace_bytes(ctypes.byref(args))
        assert need >= 0
        if need and copies > 1:   # (ONE layer with a workspace would take the split-K tile launch: tests/test_gpu_parity.py; without one: row blocks)
            ws = torch.full((need,), 0xff, dtype=torch.uint8, device=DEV)
            args.workspace, args.workspace_bytes = ws.data_ptr(), need
        assert L.tg_gemm_w4_plan(ctypes.byref(args), 0) in (_lib.TG_PLAN_PAIR, _lib.TG_PLAN_PAIR_XR, _lib.TG_PLAN_GEMV)
        _lib.check(L.tg_gemm_w4(ctypes.byref(args), 0, torch.cuda.current_stream().cuda_stream), "row-blocked launch")
        torch.cuda.synchronize()
        return ys

    y0 = run(None)
    assert not torch.isnan(y0.float()).any()
    rows = slice(None) if n <= 256 else torch.cat([torch.arange(0, 64), torch.arange(n - 64, n)])
    q = qinfo[rows] if qtype == "mx4" else qinfo[:, rows].contiguous()
    lt = lut if lut is None or lut.dim() == 1 else lut[rows].contiguous()
    for j in sorted({0, copies // 2, copies - 1}):
        _check_rows(oracle, y0[j][:, rows], codes[rows], xs[j].cpu(), q, lt, g, qtype, torch.bfloat16)
    y1 = run(res)
    assert torch.equal((y0.float() + res.float()).bfloat16().view(torch.int16), y1.view(torch.int16))


@pytest.mark.parametrize("case", [
    # (n, k, m, g, inner, qtype): activation blocks that do not fit next to the table -> workspace variant
    (64, 4096, 8, 64, 4, "any4_rowwise"), (72, 4096, 5, 128, 4, "any4_rowwise"), (64, 4096, 7, 256, 4, "int4"),
    (64, 8192, 1, 128, 4, "any4_rowwise"), (40, 8192, 3, 64, 4, "any4_global"), (64, 14336, 1, 128, 4, "any4_rowwise"),
    (64, 4096, 8, 32, 4, "mx4"), (48, 4096, 8, 32, 2, "int4"), (64, 8192, 2, 128, 2, "int4"), (64, 4096, 6, 256, 4, "any4_rowwise"),
    (64, 2048, 8, 128, 2, "any4_rowwise"),
])
def test_pair_kernel_workspace_variant(T, oracle, case):
    """tg_gemm_w4_workspace_bytes > 0: the activations are re-arranged into the caller's scratch and streamed by the waves.
    Without the scratch the same call succeeds on a reference-numerics kernel."""
    from any4_amd import ops

    n, k, m, g, inner, qtype = case
    codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, seed=n + k + m)
    y, copies = run_fast(T, codes, x, qinfo, lut, g, qtype, inner)
    assert ops.gemm_w4_plan(m, n, k, g, QT[qtype], True, inner, batch=copies, workspace=False) != "pair"
    assert_fast_close(oracle, y, codes, x, qinfo, lut, g, qtype, inner=inner, batch=copies)
    y0, _ = run_fast(T, codes, x, qinfo, lut, g, qtype, inner, workspace=False)
    w = from_bits16(oracle_weights(oracle, codes, g, qtype, qinfo, lut, torch.bfloat16), torch.bfloat16).double()
    S = (x.double().abs() @ w.abs().t()).numpy()
    d = np.abs(y0.double().cpu().numpy()[:, :n] - y.double().cpu().numpy()[:, :n])
    assert (d <= ulp16(y.double().cpu().numpy()[:, :n], torch.bfloat16) + 2.0 ** -8 * S + 1e-37).all()


@pytest.mark.parametrize("qtype", ["any4_rowwise", "any4_global", "int4", "mx4"])
@pytest.mark.parametrize("inner", [2, 4])
@pytest.mark.parametrize("g", [32, 64, 128, 256])
def test_pair_kernel_a_side(T, oracle, qtype, inner, g):
    """Aint4 weights (weightOnRight = false) on the pair-table kernel: 16x16x32 MFMA tiles, duplicated table, m <= 16."""
    from any4_amd import ops

    if qtype == "mx4":
        g = 32
    for (n, k, m) in [(64, 1024, 1), (40, 512, 3), (136, 2048, 8), (80, 8192, 8), (48, 4096, 5), (64, 2048, 12), (48, 1024, 16)]:
        if k % (16 * inner) or k % g:
            continue
        codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, seed=n + k + inner + 1)
        y, copies = run_fast(T, codes, x, qinfo, lut, g, qtype, inner, on_right=False)
        pair = ops.gemm_w4_plan(m, -(-n // 16) * 16, k, g, QT[qtype], False, inner, batch=copies) == "pair"
        assert pair or (g == 32 and k >= 8192), (n, k, m)  # (8 KiB of activation sums do not fit next to the table)
        assert_fast_close(oracle, y, codes, x, qinfo, lut, g, qtype, inner=inner, batch=copies, expect_pair=pair, on_right=False)


def test_pair_kernel_a_side_fp16_bias(T, oracle):
    codes, x, qinfo, lut = rand_problem(96, 1024, 128, 6, "any4_rowwise", dtype=torch.float16, seed=14)
    y, copies = run_fast(T, codes, x, qinfo, lut, 128, "any4_rowwise", 4, on_right=False)
    assert_fast_close(oracle, y, codes, x, qinfo, lut, 128, "any4_rowwise", dtype=torch.float16, batch=copies, on_right=False)


@pytest.mark.parametrize("qtype", ["any4_rowwise", "any4_global", "int4", "mx4"])
@pytest.mark.parametrize("inner", [2, 4, 8])
@pytest.mark.parametrize("g", [32, 64, 128, 256])
def test_pair16_single_launch(T, oracle, qtype, inner, g):
    """One problem per launch (what Any4Linear.forward issues): w4_gemm_pair16_kernel, 16 weight rows per workgroup, m <= 16."""
    if qtype == "mx4":
        g = 32
    for (n, k, m) in [(64, 1024, 1), (40, 512, 3), (136, 2048, 8), (24, 4096, 1), (200, 4096, 2), (48, 1024, 16), (16, 14336, 1)]:
        if k % (16 * inner) or k % g:
            continue
        codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, seed=n + k + inner + 7)
        y, copies = run_fast(T, codes, x, qinfo, lut, g, qtype, inner, min_items=1)
        assert copies == 1
        # (innerKTiles 8 has no single-launch pair-table kernel since round 4 -- scratch --: the streaming kernels take it)
        assert_fast_close(oracle, y, codes, x, qinfo, lut, g, qtype, inner=inner, batch=1, expect_pair=True if inner < 8 else None)


@pytest.mark.parametrize("qtype,g", [("any4_rowwise", 128), ("any4_rowwise", 32), ("any4_global", 64), ("int4", 256), ("mx4", 32)])
@pytest.mark.parametrize("shape", [
    # (n, k, m)            what it exercises in the register-resident-activation path of w4_gemm_pair16_kernel (m >= 5, one launch per layer)
    (4096, 4096, 16),      # the whole slice in one block (four super-tiles per wave): one launch where the LDS path staged k in two parts
    (4096, 4096, 5), (200, 4096, 11),   # rows >= m load row m - 1 again; a ragged last 16-row tile
    (48, 1024, 16), (136, 2048, 8),     # slices of one / two super-tiles (shorter than a block)
    (64, 8192, 9), (32, 14336, 16),     # long slices: blocks of two super-tiles, two register sets; a ragged last block (14 = 7 x 2)
    (16, 6144, 7),                      # 96 super-tiles over 16 waves: six each
])
def test_pair16_register_resident_activations(T, oracle, qtype, g, shape):
    n, k, m = shape
    for dtype in (torch.bfloat16, torch.float16):
        if qtype == "mx4" and dtype == torch.float16:
            continue  # mx4 is bf16-only (TinyGemm_int4.cu:758)
        codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, dtype=dtype, seed=n + k + m)
        y, copies = run_fast(T, codes, x, qinfo, lut, g, qtype, 4, min_items=1)
        assert copies == 1
        rows = slice(None) if n <= 256 else torch.cat([torch.arange(0, 64), torch.arange(n // 2 - 32, n // 2 + 32), torch.arange(n - 64, n)])
        q = qinfo[rows] if qtype == "mx4" else qinfo[:, rows].contiguous()
        lt = lut if lut is None or lut.dim() == 1 else lut[rows].contiguous()
        # same plan for the sliced problem? the tolerance follows the FULL problem's plan: check it here
        from any4_amd import ops
        # (5 ... 8 rows at k <= 4096 with groups of 128 / 256 go to w4_gemv_kernel's matrix-core contraction since it exists; every
        #  other shape of this list -- more rows, longer k, groups of 32 / 64, mx4 -- is this kernel's)
        assert ops.gemm_w4_plan(m, n, k, g, QT[qtype], True, 4, dtype, 1, "fast") == ("gemv" if m <= 8 and k <= 4096 and g >= 128 and qtype != "mx4" else "pair")
        yy = y[:, rows] if n > 256 else y
        _check_rows(oracle, yy, codes[rows], x, q, lt, g, qtype, dtype)


def _check_rows(oracle, y_hip, codes, x, qinfo, lut, g, qtype, dtype):
    """assert_fast_close's two tolerances on a subset of the weight rows (the plan was checked by the caller)."""
    w = from_bits16(oracle_weights(oracle, codes, g, qtype, qinfo, lut, dtype), dtype).double()
    x64 = x.double()
    y_ref = (x64 @ w.t()).numpy()
    S = (x64.abs() @ w.abs().t()).numpy()
    y_gs = gs_reference(oracle, codes, x, qinfo, lut, g, qtype, dtype)
    got = y_hip.detach().double().cpu().numpy()[:, :codes.shape[0]]
    tol = 0.5 * ulp16(y_gs, dtype) * (1 + 2.0 ** -7) + 4e-6 * S + 1e-37
    bad = np.abs(got - y_gs) > tol
    assert not bad.any(), f"vs group-scaled oracle: {bad.sum()} / {bad.size} outside tolerance; worst {np.abs(got - y_gs).max()} at {np.argwhere(bad)[:4]}"
    eps16 = 2.0 ** -9 if dtype == torch.bfloat16 else 2.0 ** -12
    assert not (np.abs(got - y_ref) > 0.5 * ulp16(y_ref, dtype) * (1 + 2.0 ** -7) + (4e-6 + eps16) * S + 1e-37).any()


@pytest.mark.parametrize("m", [5, 11, 16])
def test_pair16_register_resident_residual_and_fragment_order(T, oracle, m):
    """w4_gemm_pair16_kernel's register-resident activations hold the A operand's rows in a ROTATED order (quad-contiguous loads:
    accumulator register r of lane (n, q) = activation row 4 r + q): a fused residual (one bias row per activation row) and the
    fragment-order output must follow it -- the plain launch's bits plus a separate rounded add / re-laid out."""
    from any4_amd import _lib

    n, g, k = 4096 if m > 8 else 208, 64, 4096   # (groups of 64: the gemv kernel declines them at 5 ... 8 rows)
    codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_rowwise", seed=500 + m)
    xs, y0 = _run_xr(T, codes, x, qinfo, lut, g, "any4_rowwise", 1, plan=_lib.TG_PLAN_PAIR)
    rows = torch.cat([torch.arange(0, 64), torch.arange(n - 64, n)])
    _check_rows(oracle, y0[0][:, rows], codes[rows], xs[0], qinfo[:, rows].contiguous(), lut[rows].contiguous(), g, "any4_rowwise", torch.bfloat16)
    res = torch.randn(m, n, generator=torch.Generator().manual_seed(8)).bfloat16()
    _, y1 = _run_xr(T, codes, x, qinfo, lut, g, "any4_rowwise", 1, bias=res, residual=True, plan=_lib.TG_PLAN_PAIR)
    assert torch.equal((y0.float() + res.to(DEV).float()).bfloat16().view(torch.int16), y1.view(torch.int16))
    if m == 16:  # (fragment order: whole 16-row tiles; x in fragment order takes the XTC loads, y in fragment order the rotated rows)
        _, y2 = _run_xr(T, codes, x, qinfo, lut, g, "any4_rowwise", 1, tc=True, plan=_lib.TG_PLAN_PAIR)
        assert torch.equal(y2.view(torch.int16), y0.view(torch.int16))


@pytest.mark.parametrize("case", [
    # (n, m, g, qtype, dtype): one layer per launch with more 16-row tiles than CUs at k = 4096 (w4_gemm_pair16_loop_kernel: a workgroup walks
    # 1 ... 8 tiles; beyond that w4_gemm_xr_kernel's 64-row items): uneven ranges (5120 rows = 320 tiles over 256 workgroups), a ragged
    # last tile (8200 rows), every group size, the three LUT kinds, both dtypes
    (5120, 16, 128, "any4_rowwise", torch.bfloat16), (8200, 9, 128, "any4_rowwise", torch.bfloat16), (11008, 13, 64, "int4", torch.bfloat16),
    (16384, 16, 256, "any4_global", torch.float16), (6144, 5, 32, "any4_rowwise", torch.bfloat16), (28672, 16, 128, "any4_rowwise", torch.bfloat16),
])
def test_single_layer_with_more_tiles_than_cus(T, oracle, case):
    from any4_amd import _lib, ops

    L = _lib.load()
    n, m, g, qtype, dtype = case
    codes, x, qinfo, lut = rand_problem(n, 4096, g, m, qtype, dtype=dtype, seed=n + m)
    assert ops.gemm_w4_plan(m, n, 4096, g, QT[qtype], True, 4, dtype, 1, "fast", detail=True) in ("pair", "pair_xr")
    packed = T.convert_matrix_to_m16n8k16_Bint4_layout(codes.to(DEV), 4)
    xs, qs, ls = x.to(DEV), qinfo.to(DEV), (None if lut is None else lut.to(DEV))
    res = torch.randn(m, n, generator=torch.Generator().manual_seed(4)).to(dtype).to(DEV)

    def run(bias):  # (no workspace: with one, the 9 ... 16-row workspace variant of the pair kernel may take the launch instead)
        y = torch.full((m, n), float("nan"), dtype=dtype, device=DEV)
        args = _lib.W4Gemm(x=xs.data_ptr(), w=packed.data_ptr(), qinfo=qs.data_ptr(), lut=(ls.data_ptr() if ls is not None else None), y=y.data_ptr(),
                           m=m, wrows=n, k=4096, group=g, qtype=QT[qtype], dtype=_lib.TG_BF16 if dtype == torch.bfloat16 else _lib.TG_F16, w_on_right=1,
                           inner_k_tiles=4, batch=1, numerics=_lib.TG_NUM_FAST, bias=(bias.data_ptr() if bias is not None else None),
                           bias_row_stride=(n if bias is not None else 0))
        assert L.tg_gemm_w4_plan(ctypes.byref(args), 0) in (_lib.TG_PLAN_PAIR, _lib.TG_PLAN_PAIR_XR)
        _lib.check(L.tg_gemm_w4(ctypes.byref(args), 0, torch.cuda.current_stream().cuda_stream), "single layer")
        torch.cuda.synchronize()
        return y

    y = run(None)
    assert not torch.isnan(y.float()).any()
    rows = torch.cat([torch.arange(0, 96), torch.arange(n // 2 - 40, n // 2 + 40), torch.arange(n - 96, n)])
    q = qinfo[:, rows].contiguous()
    lt = lut if lut is None or lut.dim() == 1 else lut[rows].contiguous()
    _check_rows(oracle, y[:, rows], codes[rows], x, q, lt, g, qtype, dtype)
    assert torch.equal(y.view(torch.int16), run(None).view(torch.int16))   # deterministic
    # a fused per-row residual: the same bits as the separate add
    assert torch.equal((y.float() + res.float()).to(dtype).view(torch.int16), run(res).view(torch.int16))


def test_single_layer_with_more_tiles_than_cus_random_shapes(T, oracle):
    """Seeded random shapes through the same launches: rows any multiple of 8 in (4096, 33000) -- ragged last tiles, ranges of 1 ... 8 tiles,
    workgroups with one tile less than their neighbours --, 5 ... 16 activation rows, every group size / LUT kind / dtype: sampled rows
    against the oracle, the whole output deterministic."""
    import random

    from any4_amd import _lib

    L = _lib.load()
    rng = random.Random(20260930)
    for _ in range(10):
        n = rng.randrange(4104, 33000, 8)
        m = rng.randint(5, 16)
        g = rng.choice([32, 64, 128, 256])
        qtype = rng.choice(["any4_rowwise", "any4_rowwise", "int4", "any4_global"])
        dtype = rng.choice([torch.bfloat16, torch.float16])
        codes, x, qinfo, lut = rand_problem(n, 4096, g, m, qtype, dtype=dtype, seed=n + m)
        packed = T.convert_matrix_to_m16n8k16_Bint4_layout(codes.to(DEV), 4)
        xs, qs, ls = x.to(DEV), qinfo.to(DEV), (None if lut is None else lut.to(DEV))

        def run():
            y = torch.full((m, n), float("nan"), dtype=dtype, device=DEV)
            args = _lib.W4Gemm(x=xs.data_ptr(), w=packed.data_ptr(), qinfo=qs.data_ptr(), lut=(ls.data_ptr() if ls is not None else None), y=y.data_ptr(),
                               m=m, wrows=n, k=4096, group=g, qtype=QT[qtype], dtype=_lib.TG_BF16 if dtype == torch.bfloat16 else _lib.TG_F16,
                               w_on_right=1, inner_k_tiles=4, batch=1, numerics=_lib.TG_NUM_FAST)
            assert L.tg_gemm_w4_plan(ctypes.byref(args), 0) in (_lib.TG_PLAN_PAIR, _lib.TG_PLAN_PAIR_XR, _lib.TG_PLAN_GEMV), (n, m, g, qtype)
            _lib.check(L.tg_gemm_w4(ctypes.byref(args), 0, torch.cuda.current_stream().cuda_stream), "single layer")
            torch.cuda.synchronize()
            return y

        y = run()
        assert not torch.isnan(y.float()).any(), (n, m, g, qtype)
        rows = torch.cat([torch.arange(0, 48), torch.arange(n // 3, n // 3 + 48), torch.arange(n - 48, n)])
        q = qinfo[:, rows].contiguous()
        lt = lut if lut is None or lut.dim() == 1 else lut[rows].contiguous()
        _check_rows(oracle, y[:, rows], codes[rows], x, q, lt, g, qtype, dtype)
        assert torch.equal(y.view(torch.int16), run().view(torch.int16)), (n, m, g, qtype)


def test_pair16_fp16_bias_and_batch(T, oracle):
    codes, x, qinfo, lut = rand_problem(96, 1024, 128, 5, "any4_rowwise", dtype=torch.float16, seed=21)
    bias = torch.randn(96).half()
    y, copies = run_fast(T, codes, x, qinfo, lut, 128, "any4_rowwise", 4, bias=bias, min_items=6)  # 3 stacked problems
    assert copies == 3
    y0, _ = run_fast(T, codes, x, qinfo, lut, 128, "any4_rowwise", 4, min_items=1)
    assert_fast_close(oracle, y0, codes, x, qinfo, lut, 128, "any4_rowwise", dtype=torch.float16, batch=1)
    want = (y0.float().cpu() + bias.float()[None, :]).half()
    assert torch.equal(want.view(torch.int16), y.cpu().view(torch.int16))


def test_pair_kernel_fp16(T, oracle):
    for qtype in ("any4_rowwise", "int4"):
        codes, x, qinfo, lut = rand_problem(96, 1024, 128, 2, qtype, dtype=torch.float16, seed=4)
        y, copies = run_fast(T, codes, x, qinfo, lut, 128, qtype, 4)
        assert y.dtype == torch.float16
        assert_fast_close(oracle, y, codes, x, qinfo, lut, 128, qtype, torch.float16, batch=copies)


def test_fast_equals_reference_where_no_fast_kernel(T, oracle):
    """Shapes without a group-scaled kernel (small launches of A-side weights, innerKTiles 8 at one layer per launch with more than 16
    activation rows): the fast setting then runs the reference kernels, bit for bit."""
    import any4_amd
    from any4_amd import ops

    for on_right, m, k, inner in ((False, 2, 1024, 4), (True, 17, 4096, 8), (False, 1, 4096, 4)):
        codes, x, qinfo, lut = rand_problem(64, k, 128, m, "any4_rowwise", seed=3)
        assert ops.gemm_w4_plan(m, 64, k, 128, QT["any4_rowwise"], on_right, inner) != "pair"
        y_fast = run_rm(T, codes, x, qinfo, lut, 128, "any4_rowwise", on_right, inner)
        with any4_amd.numerics("reference"):
            y_ref = run_rm(T, codes, x, qinfo, lut, 128, "any4_rowwise", on_right, inner)
        assert torch.equal(y_fast, y_ref)


@pytest.mark.parametrize("g", [32, 128])
def test_identity_fast_within_one_ulp(T, g):
    """The reference's identity known-answer case (test_tinygemm_any4.py:14-37).  Bit-equal in reference numerics
    (test_gpu_parity.test_identity_bit_exact); the group-scaled path multiplies x by 15 * bf16(1/15) without the
    reference's rounding of that weight to 1.0, so y is x within one 16-bit ulp."""
    import any4_amd.utils as U

    k = 512
    x = torch.randn(2, k, generator=torch.Generator().manual_seed(2)).bfloat16()
    codes, sz = U.group_quantize_tensor(torch.eye(k, dtype=torch.bfloat16), 4, g)
    lut = -(torch.arange(16, dtype=torch.bfloat16) - 8)
    sz[:, :, 0] *= -1.0
    y, _ = run_fast(T, codes, x, sz, lut, g, "any4_global", 4)
    y = y.cpu()
    err = (y.double() - x.double()).abs().numpy()
    assert (err <= ulp16(x.double().numpy(), torch.bfloat16)).all()


def test_identity_mx4_fast_is_exact(T):
    """mx4 weights (fp4 * 2^e) are exact in 16 bits, so the fast numerics lose nothing: bit-equal (test_tinygemm_mx4.py:14-39)."""
    import any4_amd.utils as U

    k = 256
    x = torch.randn(2, k, generator=torch.Generator().manual_seed(3)).bfloat16()
    q, e = U.quantize_mx4(torch.eye(k), 32)
    e = e + (torch.arange(k) % 4).to(torch.uint8).unsqueeze(1)
    expect = (x.float() * (2.0 ** (torch.arange(k) % 4).float())).bfloat16()
    y, _ = run_fast(T, q, x, e, None, 32, "mx4", 4)
    assert torch.equal(y.cpu(), expect)
    e2 = e.clone()
    e2[5, :] = 255  # NaN exponent: that weight row only (test_tinygemm_mx4.py:443-506)
    y, _ = run_fast(T, q, x, e2, None, 32, "mx4", 4)
    y = y.cpu()
    assert torch.isnan(y[:, 5]).all() and not torch.isnan(y[:, :5]).any() and not torch.isnan(y[:, 6:]).any()


def test_reference_fixture_fast(T):
    """north_star: within 1e-2 max-abs of the reference's own CPU dequant-matmul on the captured any4 tensors."""
    g = load_golden("any4_n1024_k1024_g128_seed1234.npz")
    n, k, grp = int(g["n"]), int(g["k"]), int(g["g"])
    nib = g["codes_nib"]
    codes = np.empty((n, k), np.int32)
    codes[:, 0::2] = nib & 15
    codes[:, 1::2] = nib >> 4
    x = from_bits16(g["x_bits"], torch.bfloat16)
    lut = from_bits16(g["lut_m8_bits"], torch.bfloat16)
    sz = from_bits16(g["sz_bits"], torch.bfloat16)
    from any4_amd import ops

    y, copies = run_fast(T, torch.from_numpy(codes), x, sz, lut, grp, "any4_rowwise", 4)
    assert ops.gemm_w4_plan(1, n, k, grp, QT["any4_rowwise"], batch=copies) == "pair"
    y_ref = from_bits16(g["y_bits"], torch.bfloat16)
    assert (y.float().cpu() - y_ref.float()).abs().max().item() <= 1e-2


# ------------------------------------------------------------------------------------------------
# fragment-order ("TC") activations and outputs read / written by the GEMM kernel itself
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("qtype", ["any4_rowwise", "int4", "mx4"])
@pytest.mark.parametrize("case", [(64, 1024, 16, 128, 4), (40, 512, 16, 64, 2), (136, 4096, 16, 128, 4), (24, 2048, 16, 32, 8)])
def test_tc_ops_native_fragment_order(T, oracle, qtype, case, monkeypatch):
    """tinygemm_y_f16TC_x_f16TC_w_*TC with the weights on the right (TinyGemm_int4.cu:28-292): the pair-table kernels take the
    A-fragment-order activations and write the A-fragment-order output (no converter launches: they are made to fail here), and
    the result is the row-major op's, re-laid out, bit for bit."""
    from any4_amd import ops

    n, k, m, g, inner = case
    if qtype == "mx4":
        g = 32
    if k % (16 * inner) or k % g:
        pytest.skip("shape does not divide")
    codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, seed=n + m)
    d = lambda t: None if t is None else t.to(DEV)
    w2 = T.convert_matrix_to_m16n8k16_Bint4_layout(d(codes), inner)
    wrows = w2.shape[0] * 8
    if qtype == "mx4":
        qi = torch.cat([qinfo, torch.full((wrows - n, qinfo.shape[1]), 127, dtype=qinfo.dtype)]) if wrows > n else qinfo
    else:
        qi = qinfo
    xa = T.convert_matrix_to_m16n8k16_A_layout(d(x), 1)
    y_rm = run_rm(T, codes, x, qi, lut, g, qtype, True, inner)                       # [m][wrows]
    want = T.convert_matrix_to_m16n8k16_A_layout(y_rm, 1)
    def boom(*a, **k):
        raise AssertionError("converter launched: the fragment-order path is not native")
    if inner < 8:  # (innerKTiles 8 has no pair-table kernel for 16 rows: that op converts around a row-major call)
        monkeypatch.setattr(ops, "convert_matrix_from_m16n8k16_A_layout", boom)
        monkeypatch.setattr(ops, "convert_matrix_to_m16n8k16_A_layout", boom)
    if qtype == "mx4":
        got = T.tinygemm_y_f16TC_x_f16TC_w_mx4TC(xa, w2, g, d(qi), True)
    elif qtype == "int4":
        got = T.tinygemm_y_f16TC_x_f16TC_w_int4TC(xa, w2, g, d(qi), True)
    else:
        got = T.tinygemm_y_f16TC_x_f16TC_w_any4TC(xa, w2, g, d(qi), d(lut), True)
    assert got.shape == want.shape
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


# ------------------------------------------------------------------------------------------------
# the launches bench.py times, at their own shape: stacked tg_gemm_w4 over >= 16 layers of 4096 x 4096
# ------------------------------------------------------------------------------------------------

def _stacked_launch(layers, m, n, k, g, qtype, numerics, seed=0, on_right=True, native=False, calibrate=False, dtype=torch.bfloat16):
    """One tg_gemm_w4 call over `layers` independent problems, built as bench.py builds its legs (make_batch / make_args).
    native (weights on the left): `w` is the Bint4 tensor of the rows, tg_w4_gemm.w_format = TG_WFMT_ROWS -- what bench.py's
    config3 leg launches; otherwise the reference's Aint4 words.  calibrate: bench.calibrate_x on the activations (max|y| in
    (0.95, 1.9]), then the launch the caller checks."""
    from any4_amd import _lib

    L = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(seed)
    inner = 4
    assert not (native and on_right)
    wshape = (layers, n // 8, k // (16 * inner), 32, inner // 2) if (on_right or native) else (layers, n // 16, k // (16 * inner), 32, inner)
    w = torch.randint(-2 ** 31, 2 ** 31 - 1, wshape, dtype=torch.int64, device=DEV, generator=gen).to(torch.int32)
    x = torch.randn(layers, m, k, device=DEV, generator=gen).to(dtype)
    if qtype == "mx4":
        q = torch.randint(120, 131, (layers, n, k // g), dtype=torch.uint8, device=DEV, generator=gen)
        qstride = q.stride(0)
    else:
        scales = torch.rand(layers, k // g, n, device=DEV, generator=gen) * 0.02 + 0.005
        zeros = torch.randn(layers, k // g, n, device=DEV, generator=gen) * 0.01
        q = torch.stack([scales, zeros], dim=3).to(dtype).contiguous()
        qstride = q.stride(0) * 2
    lut = {"any4_rowwise": torch.randn(layers, n, 16, device=DEV, generator=gen).to(dtype),
           "any4_global": torch.randn(layers, 16, device=DEV, generator=gen).to(dtype)}.get(qtype)
    y = torch.full((layers, m, n), float("nan"), device=DEV, dtype=dtype)
    args = _lib.W4Gemm(x=x.data_ptr(), w=w.data_ptr(), qinfo=q.data_ptr(), lut=(lut.data_ptr() if lut is not None else None),
                       y=y.data_ptr(), m=m, wrows=n, k=k, group=g, qtype=QT[qtype], dtype=_lib.TG_BF16 if dtype == torch.bfloat16 else _lib.TG_F16, w_on_right=1 if on_right else 0,
                       inner_k_tiles=inner, batch=layers, stride_x=x.stride(0) * 2, stride_w=w.stride(0) * 4,
                       stride_qinfo=qstride, stride_lut=(lut.stride(0) * 2 if lut is not None else 0), stride_y=y.stride(0) * 2,
                       numerics=numerics, w_format=_lib.TG_WFMT_ROWS if native else _lib.TG_WFMT_M16N8K16)
    need = L.tg_gemm_w4_workspace_bytes(ctypes.byref(args))  # as bench.py does
    assert need >= 0
    if need:
        ws = torch.empty(need, dtype=torch.uint8, device=DEV)
        args.workspace, args.workspace_bytes = ws.data_ptr(), need

    def launch():
        _lib.check(L.tg_gemm_w4(ctypes.byref(args), 0, torch.cuda.current_stream().cuda_stream), "stacked")

    if calibrate:
        import bench

        bench.calibrate_x(launch, x, y)
        y.fill_(float("nan"))
    launch()
    torch.cuda.synchronize()
    return w, x, q, lut, y


@pytest.mark.parametrize("qtype", ["any4_rowwise", "int4", "any4_global"])
def test_m1_dot2_contraction_pinned_to_the_mfma_contraction(oracle, qtype):
    """The default m = 1 kernel contracts with per-lane v_dot2_f32_bf16 (round 3); TG_NUM_FAST_MFMA runs the same kernel with the
    32x32x16 MFMA.  Same table, same group scaling, another adder tree: both within the fast-numerics tolerance of the oracle, and
    of each other within one output ulp + twice the f32 accumulation slack -- with ordinary activations and with activations that
    contain bf16 DENORMALS and signed zeros (v_dot2 and the MFMA do not promise the same denormal handling: the products of a
    denormal activation with O(1) table values stay far below the slack either way, which is what this pins)."""
    from any4_amd import _lib

    layers, m, n, k, g = 8, 1, 4096, 4096, 128
    for denorm in (False, True):
        w, x, q, lut, y_dot = _stacked_launch(layers, m, n, k, g, qtype, _lib.TG_NUM_FAST, seed=5)
        if denorm:
            xi = x.view(torch.int16)
            xi[:, :, ::7] &= 0x807f       # exponent 0: denormals and signed zeros
            x = xi.view(torch.bfloat16)
        args = dict(layers=layers, m=m, n=n, k=k, g=g, qtype=qtype)
        y = {}
        for name, num in (("dot", _lib.TG_NUM_FAST), ("mfma", _lib.TG_NUM_FAST_MFMA)):
            yy = torch.full((layers, m, n), float("nan"), device=DEV, dtype=torch.bfloat16)
            L = _lib.load()
            a = _lib.W4Gemm(x=x.data_ptr(), w=w.data_ptr(), qinfo=q.data_ptr(), lut=(lut.data_ptr() if lut is not None else None),
                            y=yy.data_ptr(), m=m, wrows=n, k=k, group=g, qtype=QT[qtype], dtype=_lib.TG_BF16, w_on_right=1, inner_k_tiles=4,
                            batch=layers, stride_x=x.stride(0) * 2, stride_w=w.stride(0) * 4, stride_qinfo=q.stride(0) * 2,
                            stride_lut=(lut.stride(0) * 2 if lut is not None else 0), stride_y=yy.stride(0) * 2, numerics=num)
            # (the MFMA contraction of a stacked m = 1 launch runs on w4_gemm_xr_kernel's 16x16x32 tiles since round 5)
            assert L.tg_gemm_w4_plan(ctypes.byref(a), 0) == (_lib.TG_PLAN_PAIR if name == "dot" else _lib.TG_PLAN_PAIR_XR)
            _lib.check(L.tg_gemm_w4(ctypes.byref(a), 0, torch.cuda.current_stream().cuda_stream), name)
            torch.cuda.synchronize()
            y[name] = yy
        for b in (0, layers - 1):
            codes = torch.from_numpy(oracle.unpack_Bint4(w[b].cpu().numpy(), n, k))[:256]
            lb = None if lut is None else (lut[b][:256].cpu() if qtype == "any4_rowwise" else lut[b].cpu())
            qb = q[b][:, :256].contiguous().cpu()
            for name in ("dot", "mfma"):
                assert_fast_close(oracle, y[name][b][:, :256], codes, x[b].cpu(), qb, lb, g, qtype, batch=layers)
            wq = from_bits16(oracle_weights(oracle, codes, g, qtype, qb, lb), torch.bfloat16).double()
            S = (x[b].cpu().double().abs() @ wq.abs().t()).numpy()
            d, e = y["dot"][b][:, :256].double().cpu().numpy(), y["mfma"][b][:, :256].double().cpu().numpy()
            assert (np.abs(d - e) <= ulp16(e, torch.bfloat16) * (1 + 2.0 ** -7) + 8e-6 * S).all(), (qtype, denorm, np.abs(d - e).max())


@pytest.mark.parametrize("qtype,g", [("any4_rowwise", 128), ("int4", 128), ("any4_global", 128), ("mx4", 32)])
@pytest.mark.parametrize("m", [1, 8, 16])
@pytest.mark.parametrize("numerics", ["fast", "reference", "fast_mfma"])
def test_benchmarked_launch_shape(T, oracle, qtype, g, m, numerics):
    """BASELINE configs 2 and 4 exactly as bench.py launches them (legs `value`, m8, m16, int4 / nf4 / mx4, mx4_m16,
    reference_numerics, m1_mfma): ONE tg_gemm_w4 call over 16 independent layers of n = k = 4096, activations calibrated as the
    bench calibrates them.  Three layers are checked in full against the oracle; every output must have been written; the raw
    1e-2 contract of north_star on every one."""
    from any4_amd import _lib, ops

    if numerics == "fast_mfma" and (m != 1 or qtype == "mx4"):
        pytest.skip("TG_NUM_FAST_MFMA differs from TG_NUM_FAST only for the m = 1 pair-table launch (bench leg m1_mfma)")
    layers, n, k = 16, 4096, 4096
    num = {"fast": _lib.TG_NUM_FAST, "reference": _lib.TG_NUM_REFERENCE, "fast_mfma": _lib.TG_NUM_FAST_MFMA}[numerics]
    w, x, q, lut, y = _stacked_launch(layers, m, n, k, g, qtype, num, seed=m, calibrate=True)
    assert not torch.isnan(y.float()).any()
    if numerics != "reference":  # the kernel family the bench line's leg names
        assert ops.gemm_w4_plan(m, n, k, g, QT[qtype], True, 4, batch=layers, numerics=numerics, detail=True) == \
            ("pair" if m == 1 and numerics == "fast" else "pair_xr")
    for b in (0, 7, 15):
        codes = torch.from_numpy(oracle.unpack_Bint4(w[b].cpu().numpy(), n, k))
        xb, qb = x[b].cpu(), q[b].cpu()
        lb = None if lut is None else lut[b].cpu()
        if numerics != "reference":
            assert_fast_close(oracle, y[b], codes, xb, qb, lb, g, qtype, batch=layers)
        else:
            from tests.test_gpu_parity import assert_gemm_close

            assert_gemm_close(y[b], xb, oracle_weights(oracle, codes, g, qtype, qb, lb))
        # ... and both numerics inside north_star's tolerance against the reference-faithful result at THIS (the benchmarked) shape
        assert_north_star(oracle, y[b], codes, xb, qb, lb, g, qtype)


@pytest.mark.parametrize("qtype,g,m,on_right", [("any4_global", 128, 1, True), ("any4_global", 128, 3, True), ("int4", 128, 1, True),
                                                 ("mx4", 32, 1, True), ("any4_global", 128, 1, False), ("int4", 64, 2, False)])
def test_persistent_workgroups_cross_problem_boundaries(T, oracle, qtype, g, m, on_right):
    """The persistent kernel keeps its lookup table across work items when the LUT cannot change (int4, mx4) and rebuilds it
    for a global LUT only when the PROBLEM changes: 700 small problems with their own global LUTs, 3 (B side) / 6 (A side)
    work items each, so that every persistent workgroup walks across several problem boundaries."""
    from any4_amd import _lib, ops

    layers, n, k = 700, 192, 1024
    assert ops.gemm_w4_plan(m, n, k, g, QT[qtype], on_right, 4, torch.bfloat16, layers, "fast") == "pair"
    w, x, q, lut, y = _stacked_launch(layers, m, n, k, g, qtype, _lib.TG_NUM_FAST, seed=11, on_right=on_right)
    assert not torch.isnan(y.float()).any()
    for b in (0, 1, 2, 349, 350, 698, 699):
        codes = torch.from_numpy((oracle.unpack_Bint4 if on_right else oracle.unpack_Aint4)(w[b].cpu().numpy(), n, k))
        lb = None if lut is None else lut[b].cpu()
        assert_fast_close(oracle, y[b], codes, x[b].cpu(), q[b].cpu(), lb, g, qtype, batch=layers, on_right=on_right)


@pytest.mark.parametrize("qtype,g", [("any4_rowwise", 128), ("any4_global", 64)])
def test_workspace_variant_chunked_item_dealing(T, oracle, qtype, g):
    """Launches of >= 8192 work items of the workspace variant deal the items in chunks of 4 consecutive ones per workgroup visit
    (8400 items here: 4200 problems x 2 row blocks, m = 8 at k = 1024 does not fit next to the table), so a chunk straddles
    problems; first / middle / last problems against the oracle, every output written."""
    from any4_amd import _lib, ops

    layers, m, n, k = 4200, 8, 128, 1024
    assert ops.gemm_w4_plan(m, n, k, g, QT[qtype], True, 4, torch.bfloat16, layers, "fast") == "pair"
    w, x, q, lut, y = _stacked_launch(layers, m, n, k, g, qtype, _lib.TG_NUM_FAST, seed=5)
    assert not torch.isnan(y.float()).any()
    for b in (0, 1, 2, 3, 2099, 2100, 4198, 4199):
        codes = torch.from_numpy(oracle.unpack_Bint4(w[b].cpu().numpy(), n, k))
        lb = None if lut is None else lut[b].cpu()
        assert_fast_close(oracle, y[b], codes, x[b].cpu(), q[b].cpu(), lb, g, qtype, batch=layers)


@pytest.mark.parametrize("numerics", ["fast", "reference"])
@pytest.mark.parametrize("words", ["native", "reference"])
def test_benchmarked_launch_shape_config3(T, oracle, numerics, words):
    """BASELINE config 3 as bench.py launches it: m = 8, n = k = 8192, g = 128, weights on the A side, stacked over 4 layers.
    words = 'native': the bench's `config3` leg (the Bint4 tensor of the rows, TG_WFMT_ROWS: what the convert op returns by
    default); 'reference': its `config3_reference_words` leg (the reference's Aint4 words, innerKTiles 4).  256 rows of two layers
    against the oracle, raw 1e-2 contract."""
    from any4_amd import _lib, ops
    from tests.test_gpu_parity import assert_gemm_close

    layers, m, n, k, g = 4, 8, 8192, 8192, 128
    native = words == "native"
    w, x, q, lut, y = _stacked_launch(layers, m, n, k, g, "any4_rowwise", {"fast": _lib.TG_NUM_FAST, "reference": _lib.TG_NUM_REFERENCE}[numerics],
                                      seed=3, on_right=False, native=native, calibrate=True)
    assert not torch.isnan(y.float()).any()
    if numerics == "fast":
        assert ops.gemm_w4_plan(m, n, k, g, QT["any4_rowwise"], False, 4, batch=layers, weight_format=words) == "pair"
    rows = 256
    for b in (0, 3):
        codes = torch.from_numpy((oracle.unpack_Bint4 if native else oracle.unpack_Aint4)(w[b].cpu().numpy(), n, k))[:rows]
        xb, qb, lb = x[b].cpu(), q[b].cpu()[:, :rows].contiguous(), lut[b].cpu()[:rows]
        wq = from_bits16(oracle_weights(oracle, codes, g, "any4_rowwise", qb, lb), torch.bfloat16).double()
        if numerics == "fast":
            y_gs = gs_reference(oracle, codes, xb, qb, lb, g, "any4_rowwise")
            S = (xb.double().abs() @ wq.abs().t()).numpy()
            got = y[b][:, :rows].double().cpu().numpy()
            assert (np.abs(got - y_gs) <= 0.5 * ulp16(y_gs, torch.bfloat16) * (1 + 2.0 ** -7) + 4e-6 * S + 1e-37).all()
        else:
            assert_gemm_close(y[b][:, :rows], xb, oracle_weights(oracle, codes, g, "any4_rowwise", qb, lb))
        assert_north_star(oracle, y[b][:, :rows], codes, xb, qb, lb, g, "any4_rowwise")


@pytest.mark.parametrize("m,n,k,layers", [(8, 8192, 8192, 4), (16, 8192, 8192, 4), (16, 4096, 14336, 8), (8, 4096, 14336, 8), (12, 14336, 4096, 4)])
def test_stacked_weights_on_the_right_beyond_k4096(T, oracle, m, n, k, layers):
    """Weights on the right at the other Llama-3-8B shapes, stacked (the workspace / register-resident variants of the pair-table
    family at k = 8192 / 14336; TinyGemmImpl.cuh:132-217 takes any k % 32 == 0): 192 rows of the first and last layer against the
    oracle, raw 1e-2 contract."""
    from any4_amd import _lib, ops

    g, qtype = 128, "any4_rowwise"
    w, x, q, lut, y = _stacked_launch(layers, m, n, k, g, qtype, _lib.TG_NUM_FAST, seed=m + k, calibrate=True)
    assert not torch.isnan(y.float()).any()
    assert ops.gemm_w4_plan(m, n, k, g, QT[qtype], True, 4, batch=layers) == "pair"
    if k in (8192, 14336) and n % 64 == 0:
        # activations resident in registers at every one of these shapes: packed rows up to 8 rows (k = 14336: at 8), k-windows with
        # f32 partial sums in the workspace from 9 rows on (w4_gemm_xr.cuh; TinyGemmImpl.cuh:132-217 takes any k % 32 == 0)
        assert ops.gemm_w4_plan(m, n, k, g, QT[qtype], True, 4, batch=layers, detail=True) == "pair_xr"
    for b in (0, layers - 1):
        for r0 in (0, n - 192):
            codes = torch.from_numpy(oracle.unpack_Bint4(w[b].cpu().numpy(), n, k))[r0:r0 + 192]
            xb, qb, lb = x[b].cpu(), q[b].cpu()[:, r0:r0 + 192].contiguous(), lut[b].cpu()[r0:r0 + 192]
            assert_fast_close(oracle, y[b][:, r0:r0 + 192], codes, xb, qb, lb, g, qtype, batch=layers, expect_pair=None)
            assert_north_star(oracle, y[b][:, r0:r0 + 192], codes, xb, qb, lb, g, qtype)


@pytest.mark.parametrize("case", [
    # (m, n, k, g, qtype, dtype, layers): every launch >= 512 64-row work items, so that w4_gemm_xr_kernel takes it
    (3, 1024, 8192, 128, "any4_rowwise", torch.bfloat16, 32),   # packed rows (two chunks per register set), odd row count
    (8, 1024, 8192, 32, "int4", torch.bfloat16, 32),            # ... one group per chunk: the two halves of a row hold different groups
    (5, 1024, 8192, 64, "any4_global", torch.bfloat16, 32),
    (8, 1024, 8192, 256, "any4_rowwise", torch.float16, 32),    # ... fp16
    (8, 1024, 14336, 128, "any4_rowwise", torch.bfloat16, 32),  # ... 56 chunks per slice
    (8, 1024, 14336, 64, "int4", torch.float16, 32),
    (11, 1024, 14336, 128, "any4_rowwise", torch.bfloat16, 32),  # k-windows 16 + 16 + 24 chunks, f32 partial sums in the workspace
    (16, 1024, 14336, 64, "any4_global", torch.bfloat16, 32),
    (9, 1024, 14336, 32, "int4", torch.float16, 32),
    (13, 1024, 8192, 128, "any4_rowwise", torch.float16, 32),    # k-windows 16 + 16
    (16, 2048, 8192, 256, "int4", torch.bfloat16, 16),           # (groups of 256: windows of 4096 hold whole groups)
])
def test_register_resident_kernel_beyond_k4096_variants(T, oracle, case):
    """The register-resident-activation kernel's round-5 paths at k = 8192 / 14336 over group sizes, quantisation variants, both
    dtypes and odd row counts: packed rows (m <= 8) and k-windows (m >= 9), first / last layer, first / last rows, both oracles."""
    from any4_amd import _lib, ops

    m, n, k, g, qtype, dtype, layers = case
    assert ops.gemm_w4_plan(m, n, k, g, QT[qtype], True, 4, dtype, layers, detail=True) == "pair_xr"
    w, x, q, lut, y = _stacked_launch(layers, m, n, k, g, qtype, _lib.TG_NUM_FAST, seed=m * 7 + g, dtype=dtype)
    assert not torch.isnan(y.float()).any()
    for b in (0, layers - 1):
        for r0 in (0, n - 128):
            codes = torch.from_numpy(oracle.unpack_Bint4(w[b].cpu().numpy(), n, k))[r0:r0 + 128]
            qb = q[b].cpu()[:, r0:r0 + 128].contiguous()
            lb = None if lut is None else (lut[b].cpu()[r0:r0 + 128] if qtype == "any4_rowwise" else lut[b].cpu())
            assert_fast_close(oracle, y[b][:, r0:r0 + 128], codes, x[b].cpu(), qb, lb, g, qtype, dtype=dtype, batch=layers, expect_pair=None)


def test_k_windows_bias_and_workspace_protocol(T, oracle):
    """k-windows (m >= 9 at k = 14336): the fused bias is added by the kernel that sums the windows (rounded sum + bias, rounded
    again: the bits of the separate add); without the workspace the call still works (on an older kernel) and plan says so."""
    import ctypes

    from any4_amd import _lib, ops

    layers, m, n, k, g = 32, 12, 1024, 14336, 128
    L = _lib.load()
    w, x, q, lut, y = _stacked_launch(layers, m, n, k, g, "any4_rowwise", _lib.TG_NUM_FAST, seed=21)
    bias = torch.randn(layers, n, device=DEV).to(torch.bfloat16)
    yb = torch.full_like(y, float("nan"))
    args = _lib.W4Gemm(x=x.data_ptr(), w=w.data_ptr(), qinfo=q.data_ptr(), lut=lut.data_ptr(), y=yb.data_ptr(), m=m, wrows=n, k=k, group=g,
                       qtype=QT["any4_rowwise"], dtype=_lib.TG_BF16, w_on_right=1, inner_k_tiles=4, batch=layers, stride_x=x.stride(0) * 2,
                       stride_w=w.stride(0) * 4, stride_qinfo=q.stride(0) * 2, stride_lut=lut.stride(0) * 2, stride_y=yb.stride(0) * 2,
                       numerics=_lib.TG_NUM_FAST, bias=bias.data_ptr(), stride_bias=bias.stride(0) * 2)
    need = L.tg_gemm_w4_workspace_bytes(ctypes.byref(args))
    assert need == 3 * layers * m * n * 4            # three windows of f32 partial sums
    assert L.tg_gemm_w4_plan(ctypes.byref(args), 0) != _lib.TG_PLAN_PAIR_XR   # no workspace attached: another kernel would run
    ws = torch.full((need,), 0xff, dtype=torch.uint8, device=DEV)
    args.workspace, args.workspace_bytes = ws.data_ptr(), need
    assert L.tg_gemm_w4_plan(ctypes.byref(args), 0) == _lib.TG_PLAN_PAIR_XR
    _lib.check(L.tg_gemm_w4(ctypes.byref(args), 0, torch.cuda.current_stream().cuda_stream), "k-windows + bias")
    torch.cuda.synchronize()
    assert torch.equal(yb, y + bias[:, None, :])
    args.workspace, args.workspace_bytes = None, 0    # ... and without the workspace: the same numbers within the tolerance, older kernel
    y2 = torch.full_like(y, float("nan"))
    args.y, args.bias = y2.data_ptr(), None
    _lib.check(L.tg_gemm_w4(ctypes.byref(args), 0, torch.cuda.current_stream().cuda_stream), "no workspace")
    torch.cuda.synchronize()
    assert not torch.isnan(y2.float()).any() and (y2.float() - y.float()).abs().max() <= 0.02 * y.float().abs().max()


# ------------------------------------------------------------------------------------------------
# bias fused into the output store
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("numerics", ["fast", "reference"])
@pytest.mark.parametrize("kernel,cls", [("linear_y_f16RM_x_f16RM_W_any4TC", "Any4Linear"), ("linear_y_f16RM_W_any4TC_x_f16RM", "Any4Linear"),
                                        ("linear_y_f16RM_W_int4TC_x_f16RM", "Int4Linear"), ("linear_y_f16RM_x_f16RM_W_int8TC", "Int8Linear")])
def test_module_bias_is_fused_and_bit_identical(T, kernel, cls, numerics):
    """modules.py:221-222 computes y = gemm(x); y = y + bias as two 16-bit ops.  The fused store must give the same bits."""
    import any4_amd
    import modules
    from any4_amd import ops

    n, k, g, m = 64, 512, 128, 3
    gen = torch.Generator().manual_seed(8)
    mod = getattr(modules, cls)(k, n, bias=True, device=DEV, dtype=torch.bfloat16, group_size=g, kernel=kernel)
    hi = 256 if cls == "Int8Linear" else 16
    mod.weight.data = torch.randint(0, hi, (n, k), dtype=torch.int32, generator=gen).to(DEV)
    mod.scales_and_zeros.data = (torch.rand(k // g, n, 2, generator=gen) * 0.02).bfloat16().to(DEV)
    if cls == "Any4Linear":
        mod.lut.data = torch.randn(n, 16, generator=gen).bfloat16().to(DEV)
    mod.bias.data = torch.randn(n, generator=gen).bfloat16().to(DEV)
    mod.reshape_weight()
    x = torch.randn(2, m, k, generator=gen).bfloat16().to(DEV)
    with any4_amd.numerics(numerics):
        calls = []
        orig = ops._take_bias
        ops._take_bias = lambda wrows, xx: (calls.append(1), orig(wrows, xx))[1]
        try:
            y = mod(x)
        finally:
            ops._take_bias = orig
        bias = mod.bias
        mod.bias = None
        y_plain = mod(x)
        mod.bias = bias
    assert calls, "the GEMM op never looked for a fused bias"
    assert y.shape == (2, m, n)
    assert torch.equal(y, y_plain + bias)


@pytest.mark.parametrize("kernel,cls", [("linear_y_f16RM_x_f16RM_W_any4TC", "Any4Linear"), ("linear_y_f16RM_W_any4TC_x_f16RM", "Any4Linear"),
                                        ("linear_y_f16RM_W_int4TC_x_f16RM", "Int4Linear")])
def test_module_more_than_16_rows(T, oracle, kernel, cls):
    """A module forward with 2 x 21 = 42 activation rows (default numerics): the split-K tile launch with the op's workspace, three launches of
    up to 16 rows inside ONE op call without, on both
    operand sides (weights on the left = the native row-per-lane words, a B-side call), bias fused into every block's store -- every row
    within the group-scaled tolerance of the oracle, the same bits as the separate add."""
    import any4_amd
    import modules
    from any4_amd import ops

    n, k, g, m = 256, 4096, 128, 21
    gen = torch.Generator().manual_seed(18)
    mod = getattr(modules, cls)(k, n, bias=True, device=DEV, dtype=torch.bfloat16, group_size=g, kernel=kernel)
    qtype = "any4_rowwise" if cls == "Any4Linear" else "int4"
    codes, x2, qinfo, lut = rand_problem(n, k, g, 2 * m, qtype, seed=77)
    mod.weight.data = codes.to(DEV)
    mod.scales_and_zeros.data = qinfo.to(DEV)
    if cls == "Any4Linear":
        mod.lut.data = lut.to(DEV)
    mod.bias.data = torch.randn(n, generator=gen).bfloat16().to(DEV)
    with any4_amd.weight_format("native"):   # (this file's fixture packs the reference's Aint4 words, which keep their own kernels)
        mod.reshape_weight()
    on_right = "x_f16RM_W" in kernel
    assert mod.weight_format == (None if on_right else "native")
    # (round 6: ONE layer from 17 rows with the op's workspace = the split-K tile launch; without a workspace: 16-row blocks -- both checked here)
    assert ops.gemm_w4_plan(2 * m, n, k, g, QT[qtype], on_right, 4, weight_format="native") == "tile"
    assert ops.gemm_w4_plan(2 * m, n, k, g, QT[qtype], on_right, 4, weight_format="native", workspace=False) in ("pair", "pair_xr", "gemv")
    x = x2.view(2, m, k).to(DEV)
    y = mod(x)
    bias = mod.bias
    mod.bias = None
    y_plain = mod(x)
    mod.bias = bias
    assert y.shape == (2, m, n)
    assert torch.equal(y, y_plain + bias)
    # (the tile GEMM computes the reference's own weights: the GEMM tolerance against the reference-faithful oracle)
    from tests.test_gpu_parity import assert_gemm_close, oracle_weights
    assert_gemm_close(y_plain.view(2 * m, n).cpu(), x2, oracle_weights(oracle, codes, g, qtype, qinfo, lut, torch.bfloat16), torch.bfloat16)
    # the same forward without the workspace: three launches of up to 16 rows inside ONE op call
    saved = dict(ops._WS_BYTES)
    ops._WS_BYTES.clear()
    real = ops._L.tg_gemm_w4_workspace_bytes
    ops._L.tg_gemm_w4_workspace_bytes = lambda a: 0
    mod.__dict__.pop("_plan", None)
    mod.__dict__.pop("_no_plan", None)
    try:
        yb = mod(x)
        mod.bias = None
        yb_plain = mod(x)
    finally:
        mod.bias = bias
        ops._L.tg_gemm_w4_workspace_bytes = real
        ops._WS_BYTES.clear()
        ops._WS_BYTES.update(saved)
        mod.__dict__.pop("_plan", None)
    assert torch.equal(yb, yb_plain + bias)
    _check_rows(oracle, yb_plain.view(2 * m, n), codes, x2, qinfo, lut, g, qtype, torch.bfloat16)


def test_pair_kernel_fused_bias(T, oracle):
    """The same store-side bias in the pair-table kernel (reached through the stacked C-ABI launch)."""
    codes, x, qinfo, lut = rand_problem(128, 1024, 128, 2, "any4_rowwise", seed=12)
    bias = torch.randn(128, generator=torch.Generator().manual_seed(1)).bfloat16()
    y0, _ = run_fast(T, codes, x, qinfo, lut, 128, "any4_rowwise", 4)
    y1, _ = run_fast(T, codes, x, qinfo, lut, 128, "any4_rowwise", 4, bias=bias)
    assert torch.equal(y1, y0 + bias.to(DEV))


def test_bias_not_fused_into_fragment_layouts(T):
    """A TC-layout functional returns fragment-order output: an offered bias must not be consumed by its inner GEMM."""
    from any4_amd import ops

    codes, x, qinfo, lut = rand_problem(32, 256, 128, 4, "any4_rowwise", seed=1)
    d = lambda t: t.to(DEV)
    w2 = T.convert_matrix_to_m16n8k16_Bint4_layout(d(codes), 4)
    x2 = T.convert_matrix_to_m16n8k16_A_layout(d(x), 1)
    with ops.fused_bias(torch.ones(32, dtype=torch.bfloat16, device=DEV)) as fb:
        T.tinygemm_y_f16TC_x_f16TC_w_any4TC(x2, w2, 128, d(qinfo), d(lut), True)
    assert not fb.consumed


# ------------------------------------------------------------------------------------------------
# re-entrancy: two host threads, two streams (reference contract: TinyGemm_int4.cu:41-42, stateless)
# ------------------------------------------------------------------------------------------------

def test_two_host_threads_two_streams(T, oracle):
    import any4_amd

    probs = [rand_problem(128, 1024, 128, 1 + t, "any4_rowwise", seed=20 + t) for t in range(2)]
    dev = [[None if v is None else v.to(DEV) for v in p] for p in probs]
    packed = [T.convert_matrix_to_m16n8k16_Bint4_layout(d[0], 4) for d in dev]
    expect = []
    for t in range(2):
        with any4_amd.numerics("reference" if t else "fast"):
            expect.append(T.tinygemm_y_f16RM_x_f16RM_w_any4TC(dev[t][1], packed[t], 128, dev[t][2], dev[t][3], True).clone())
    torch.cuda.synchronize()
    results, errors = [None, None], []

    def work(t):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s), any4_amd.numerics("reference" if t else "fast"):
                out = []
                for _ in range(200):
                    out.append(T.tinygemm_y_f16RM_x_f16RM_w_any4TC(dev[t][1], packed[t], 128, dev[t][2], dev[t][3], True))
                s.synchronize()
            results[t] = out
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for t in range(2):
        assert all(torch.equal(o, expect[t]) for o in results[t])


# ------------------------------------------------------------------------------------------------
# BASELINE config 5 at its own size: one decoder layer of Llama-3-8B shapes through the decode harness
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("bs", [1, 8])
def test_decode_layer_at_llama3_8b_shapes(oracle, bs):
    """hidden 4096, 32 / 8 heads of 128, intermediate 14336, g = 128 (q/k/v 6144 x 4096, o 4096 x 4096, gate|up 28672 x 4096,
    down 4096 x 14336: the launches of config 5), default numerics, against the same stack on nn.Linear with the oracle's
    dequantised weights.  (The vocabulary is cut to 1024 to keep the LM head small.)"""
    from any4_amd.decode import DecodeConfig, DecodeStack
    from tests.test_gpu_decode import _PairedFactories

    cfg = DecodeConfig(hidden=4096, inter=14336, layers=1, heads=32, kv_heads=8, head_dim=128, vocab=1024, max_seq=64, group_size=128)
    fac = _PairedFactories(oracle, cfg, "linear_y_f16RM_x_f16RM_W_any4TC")
    q = DecodeStack(cfg, fac.any4, DEV, torch.bfloat16, bs=bs, seed=5, fused=True)
    d = DecodeStack(cfg, fac.dense, DEV, torch.bfloat16, bs=bs, seed=5, fused=False)
    toks = torch.randint(0, cfg.vocab, (4, bs), generator=torch.Generator().manual_seed(1)).to(DEV)
    for i, t in enumerate(toks):
        a, b = q.decode(t, i).float(), d.decode(t, i).float()
        assert torch.isfinite(a).all()
        assert (a - b).abs().max() <= 0.03 * b.abs().max() + 1e-3, (i, (a - b).abs().max(), b.abs().max())


@pytest.mark.parametrize("qtype,g", [("any4_rowwise", 128), ("int4", 128), ("mx4", 32)])
def test_default_numerics_row_does_not_depend_on_batch_shape(T, oracle, qtype, g):
    """in the default numerics the kernel family changes with the launch shape (m = 1 / m <= 8 / m <= 16 / larger, one
    layer or many), so the SAME activation row may come back with different low bits.  The contract: whatever m the row travels in,
    its outputs stay inside the module's stated bound around the group-scaled sum -- hence within one rounding + accumulation slack
    of each other -- and mx4 (exact weights) stays within the f32 accumulation slack alone."""
    n, k = 256, 4096
    codes, x, qinfo, lut = rand_problem(n, k, g, 33, qtype, seed=77)
    y_gs = gs_reference(oracle, codes, x[:1], qinfo, lut, g, qtype)
    w = from_bits16(oracle_weights(oracle, codes, g, qtype, qinfo, lut), torch.bfloat16).double()
    S = (x[:1].double().abs() @ w.abs().t()).numpy()
    tol = 0.5 * ulp16(y_gs, torch.bfloat16) * (1 + 2.0 ** -7) + (4e-6 + 2.0 ** -9) * S + 1e-37  # (reference kernels at m > 16)
    rows = {}
    for m in (1, 2, 8, 9, 16, 33):
        y = run_rm(T, codes, x[:m].contiguous(), qinfo, lut, g, qtype, True, 4)
        rows[m] = y[:1, :n].double().cpu().numpy()
        bad = np.abs(rows[m] - y_gs) > tol
        assert not bad.any(), f"m = {m}: row 0 is {np.abs(rows[m] - y_gs).max():.3e} from the group-scaled sum"
    for m in (2, 8, 9, 16, 33):
        assert (np.abs(rows[m] - rows[1]) <= 2 * tol).all(), f"row 0 at m = {m} against m = 1"


def test_module_launch_plan_is_the_full_path_and_follows_its_inputs():
    """Any4Linear / Int4Linear.forward re-issue a recorded launch (ops.LaunchPlan) for a repeated (module, activation shape): the same
    bits as the fully validated path, for fresh inputs, and the plan is dropped when anything it depends on changes -- a parameter
    re-assigned, another numerics setting, another shape, a non-contiguous input."""
    import any4_amd
    import modules
    from any4_amd import ops

    n, k, g = 1024, 2048, 128
    gen = torch.Generator().manual_seed(2)
    for cls, kw in ((modules.Any4Linear, {}), (modules.Int4Linear, {})):
        lin = cls(k, n, bias=False, device=DEV, dtype=torch.bfloat16, group_size=g, **kw)
        codes, x, qinfo, lut = rand_problem(n, k, g, 3, "any4_rowwise", seed=4)
        lin.weight.data = codes.to(DEV)
        lin.scales_and_zeros.data = qinfo.to(DEV)
        if hasattr(lin, "lut") and lin.lut is not None:
            lin.lut.data = lut.to(DEV)
        lin.reshape_weight()
        full = lambda xx: lin._gemm(xx)                      # the fully validated path (no plan)
        xs = [torch.randn(3, k, generator=gen).bfloat16().to(DEV) for _ in range(4)]
        ys = [lin(xx) for xx in xs]                          # first call records, the others run the plan
        assert lin.__dict__["_plan"] is not None
        for xx, yy in zip(xs, ys):
            assert torch.equal(yy, full(xx))
        # another shape, a 3-D input, a non-contiguous input
        x1 = torch.randn(2, 5, k, generator=gen).bfloat16().to(DEV)
        assert torch.equal(lin(x1), full(x1.view(-1, k)).view(2, 5, n))
        xt = torch.randn(k, 3, generator=gen).bfloat16().to(DEV).t()
        with pytest.raises(RuntimeError, match="contiguous"):      # (as in the reference: TinyGemm_int4.cu checks is_contiguous)
            lin(xt)
        # a re-assigned parameter: the old plan must not be used
        q2 = (qinfo * 2).to(DEV)
        lin.scales_and_zeros.data = q2
        y_new = lin(xs[0])
        assert torch.equal(y_new, full(xs[0])) and not torch.equal(y_new, ys[0])
        # another numerics setting
        with any4_amd.numerics("reference"):
            y_ref = lin(xs[0])
            assert torch.equal(y_ref, full(xs[0]))
        assert torch.equal(lin(xs[0]), y_new)

