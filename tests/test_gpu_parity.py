"""GPU suite (-m gpu): parity of the HIP path (through torch.ops.tinygemm -> C ABI) against the CPU
oracle on identical seeded inputs, against the committed golden fixtures, and -- at BASELINE's full
sizes -- through size-independent properties.

Every test of this file runs in the library's REFERENCE numerics (conftest.reference_numerics): dequantised weights
bit-identical to the reference kernels'.  The default fast (group-scaled) numerics are covered by test_gpu_fast.py.

Tolerances (stated once, used everywhere):
  * packing / layout conversion / debug dequant: bit-exact.
  * GEMM: the dequantised bf16 weights are bit-identical by construction (f32 fma + RNE == the oracle),
    so the only freedom is fp32 summation order and the final rounding.  With y64 = exact (float64)
    contraction of the bf16 inputs and S = sum_k |x_k w_k|:
        |y_hip - y64| <= 0.5 ulp_bf16(y64) * (1 + 2^-7) + 4e-6 * S
    (half an output ulp for the final RNE, plus a bound on fp32 accumulation error; the observed
    accumulation error is ~2e-7 * S).
  * against the reference's own CPU dequant-matmul (fixture H-Q): max-abs <= 1e-2 at max|y| ~ 2.2
    (north_star).
"""
import os

import numpy as np
import pytest
import torch

from tests.conftest import bf16_ulp, bits16, from_bits16, load_golden

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("reference_numerics", "reference_weight_format")]

DEV = "cuda:0"


@pytest.fixture(scope="module")
def T():
    import tinygemm  # noqa: F401  registers the ops; ImportError if the HIP library is missing

    assert torch.cuda.is_available(), "the gpu suite needs a GPU"
    return torch.ops.tinygemm


def ulp16(y64, dtype):
    y = np.abs(np.asarray(y64, np.float64))
    e = np.floor(np.log2(np.maximum(y, 1e-300)))
    if dtype == torch.bfloat16:
        return np.exp2(e - 7)
    return np.exp2(np.maximum(e, -14) - 10)


def assert_gemm_close(y_hip, x, w_bits, dtype=torch.bfloat16):
    """x: torch [m][k] 16-bit (cpu); w_bits: oracle-dequantised weights [rows][k] as uint16 bits."""
    w = from_bits16(w_bits, dtype).double()
    x64 = x.double().cpu()
    y64 = (x64 @ w.t()).numpy()
    S = (x64.abs() @ w.abs().t()).numpy()
    got = y_hip.detach().double().cpu().numpy()
    tol = 0.5 * ulp16(y64, dtype) * (1 + 2.0 ** -7) + 4e-6 * S + 1e-37
    bad = np.abs(got - y64) > tol
    assert not bad.any(), f"{bad.sum()} / {bad.size} outside tolerance; worst {np.abs(got - y64).max()}"


def rand_problem(n, k, g, m, qtype, dtype=torch.bfloat16, seed=0):
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
        lut = {"int4": None, "any4_global": torch.randn(16, generator=gen).to(dtype),
               "any4_rowwise": torch.randn(n, 16, generator=gen).to(dtype)}[qtype]
    return codes, x, qinfo, lut


def oracle_weights(oracle, codes, g, qtype, qinfo, lut, dtype=torch.bfloat16):
    q = {"int4": oracle.Q_INT4, "any4_global": oracle.Q_ANY4_GLOBAL, "any4_rowwise": oracle.Q_ANY4_ROWWISE, "mx4": oracle.Q_MX4}[qtype]
    qi = qinfo.numpy() if qtype == "mx4" else bits16(qinfo)
    return oracle.dequant(codes.numpy(), g, q, qi, None if lut is None else bits16(lut),
                          oracle.BF16 if dtype == torch.bfloat16 else oracle.F16)


def run_rm(T, codes, x, qinfo, lut, g, qtype, on_right, inner):
    d = lambda t: None if t is None else t.to(DEV)
    if on_right:
        w2 = T.convert_matrix_to_m16n8k16_Bint4_layout(d(codes), inner)
        A, B = d(x), w2
    else:
        w2 = T.convert_matrix_to_m16n8k16_Aint4_layout(d(codes), inner)
        A, B = w2, d(x)
    if qtype == "mx4":
        return T.tinygemm_y_f16RM_x_f16RM_w_mx4TC(A, B, g, d(qinfo), on_right)
    if qtype == "int4":
        return T.tinygemm_y_f16RM_x_f16RM_w_int4TC(A, B, g, d(qinfo), on_right)
    return T.tinygemm_y_f16RM_x_f16RM_w_any4TC(A, B, g, d(qinfo), d(lut), on_right)


# ------------------------------------------------------------------------------------------------
# packing and layout conversion: bit-exact
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("inner", [2, 4, 8])
@pytest.mark.parametrize("n,k", [(8, 128), (19, 256), (64, 1024), (100, 640), (256, 4096)])
def test_pack_Bint4_bit_exact(T, oracle, inner, n, k):
    if k % (16 * inner):
        pytest.skip("k not a multiple of 16*innerKTiles")
    codes = torch.randint(0, 16, (n, k), dtype=torch.int32, generator=torch.Generator().manual_seed(n + k))
    got = T.convert_matrix_to_m16n8k16_Bint4_layout(codes.to(DEV), inner).cpu().numpy()
    assert np.array_equal(got, oracle.pack_Bint4(codes.numpy(), inner))


@pytest.mark.parametrize("inner", [1, 2, 4])
@pytest.mark.parametrize("m,k", [(16, 64), (21, 96), (48, 512), (33, 1000), (256, 4096), (7, 17)])
def test_pack_Aint4_bit_exact(T, oracle, inner, m, k):
    codes = torch.randint(0, 16, (m, k), dtype=torch.int32, generator=torch.Generator().manual_seed(m + k))
    got = T.convert_matrix_to_m16n8k16_Aint4_layout(codes.to(DEV), inner).cpu().numpy()
    assert np.array_equal(got, oracle.pack_Aint4(codes.numpy(), inner))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k", [(16, 16), (5, 40), (33, 100), (48, 256), (1, 4096), (19, 7)])
def test_layout16_bit_exact_and_roundtrip(T, oracle, dtype, m, k):
    x = torch.randn(m, k, generator=torch.Generator().manual_seed(1)).to(dtype)
    a = T.convert_matrix_to_m16n8k16_A_layout(x.to(DEV), 1)
    assert np.array_equal(bits16(a), oracle.to_A16(bits16(x)))
    assert torch.equal(T.convert_matrix_from_m16n8k16_A_layout(a, m, k).cpu(), x)
    for inner in (1, 2):
        b = T.convert_matrix_to_m16n8k16_B_layout(x.to(DEV), inner)
        assert np.array_equal(bits16(b), oracle.to_B16(bits16(x), inner))
        assert torch.equal(T.convert_matrix_from_m16n8k16_B_layout(b, m, k).cpu(), x)


@pytest.mark.parametrize("inner_b,inner_a", [(2, 1), (4, 2), (8, 4)])
def test_pack_out_of_range_codes_bit_exact(T, oracle, inner_b, inner_a):
    """The reference ORs the shifted UNMASKED 32-bit inputs (TinyGemmConvertB.cu:302-303, TinyGemmConvertA.cu:280-281):
    codes outside 0..15 -- negative ones included -- must give the same words as that expression."""
    gen = torch.Generator().manual_seed(5)
    codes = torch.randint(-2**31, 2**31 - 1, (24, 256), dtype=torch.int64, generator=gen).to(torch.int32)
    codes[::3] = torch.randint(0, 300, (8, 256), dtype=torch.int32, generator=gen)
    got = T.convert_matrix_to_m16n8k16_Bint4_layout(codes.to(DEV), inner_b).cpu().numpy()
    assert np.array_equal(got, oracle.pack_Bint4(codes.numpy(), inner_b))
    got = T.convert_matrix_to_m16n8k16_Aint4_layout(codes.to(DEV), inner_a).cpu().numpy()
    assert np.array_equal(got, oracle.pack_Aint4(codes.numpy(), inner_a))


def test_dequant_int4_debug_bit_exact(T, oracle):
    words = torch.randint(-2**31, 2**31 - 1, (5000,), dtype=torch.int64).to(torch.int32)
    got = T.tinygemm_dequant_int4(words.to(DEV))
    assert np.array_equal(bits16(got), oracle.dequant_int4_debug(words.numpy()))


def test_packed_full_size_roundtrip(T, oracle):
    """BASELINE full size (4096x4096): GPU pack == oracle pack on sampled tiles, and unpack(pack) == codes."""
    n = k = 4096
    codes = torch.randint(0, 16, (n, k), dtype=torch.int32, generator=torch.Generator().manual_seed(9))
    pb = T.convert_matrix_to_m16n8k16_Bint4_layout(codes.to(DEV), 4).cpu().numpy()
    assert np.array_equal(oracle.unpack_Bint4(pb, n, k), codes.numpy())
    pa = T.convert_matrix_to_m16n8k16_Aint4_layout(codes.to(DEV), 4).cpu().numpy()
    assert np.array_equal(oracle.unpack_Aint4(pa, n, k), codes.numpy())


# ------------------------------------------------------------------------------------------------
# GEMM vs oracle
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("qtype", ["any4_rowwise", "any4_global", "int4", "mx4"])
@pytest.mark.parametrize("on_right,inner", [(True, 2), (True, 4), (True, 8), (False, 1), (False, 2), (False, 4)])
@pytest.mark.parametrize("g", [32, 128])
def test_gemm_rm_vs_oracle(T, oracle, qtype, on_right, inner, g):
    n, k, m = 48, 1024, 5
    codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, seed=inner + g)
    y = run_rm(T, codes, x, qinfo, lut, g, qtype, on_right, inner)
    assert y.shape == (m, n)
    assert_gemm_close(y, x, oracle_weights(oracle, codes, g, qtype, qinfo, lut))


@pytest.mark.parametrize("m", [1, 2, 8, 16, 17, 33, 48])
@pytest.mark.parametrize("on_right", [True, False])
def test_gemm_m_sweep(T, oracle, m, on_right):
    n, k, g = 64, 512, 64
    codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_rowwise", seed=m)
    y = run_rm(T, codes, x, qinfo, lut, g, "any4_rowwise", on_right, 4)
    assert_gemm_close(y, x, oracle_weights(oracle, codes, g, "any4_rowwise", qinfo, lut))


@pytest.mark.parametrize("on_right,inner", [(True, 2), (True, 4), (True, 8), (False, 1), (False, 2), (False, 4)])
def test_gemm_general_k(T, oracle, on_right, inner):
    """k from the smallest legal value upwards in layout-granularity steps
    (reference test_tinygemm_any4.py:141-163), including k that leaves the last step ragged."""
    step = max(32, 16 * inner)
    for k in list(range(step, 8 * step + 1, step)) + [1024 + step]:
        g = 32
        n, m = 32, 3
        codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_global", seed=k)
        y = run_rm(T, codes, x, qinfo, lut, g, "any4_global", on_right, inner)
        assert_gemm_close(y, x, oracle_weights(oracle, codes, g, "any4_global", qinfo, lut))


@pytest.mark.parametrize("n", [8, 24, 40])
def test_gemm_ragged_row_tiles(T, oracle, n):
    """Bint4 with an odd number of 8-row tiles: the second half of the last 16-row MFMA tile is padding."""
    k, g, m = 256, 128, 2
    codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_rowwise", seed=n)
    y = run_rm(T, codes, x, qinfo, lut, g, "any4_rowwise", True, 4)
    assert y.shape == (m, n)
    assert_gemm_close(y, x, oracle_weights(oracle, codes, g, "any4_rowwise", qinfo, lut))


@pytest.mark.parametrize("g", [32, 64, 128, 256])
@pytest.mark.parametrize("on_right,inner", [(True, 4), (False, 4), (True, 8), (False, 1)])
def test_identity_bit_exact(T, g, on_right, inner):
    """w = eye(k) group-quantised, LUT = 8 - arange(16) with negated scales: y must equal x bit for bit
    (reference tests/tinygemm/test_tinygemm_any4.py:14-37, 360-382, 442-464)."""
    import any4_amd.utils as U

    k = 512
    for m in (5, 16, 19):
        x = torch.randn(m, k, generator=torch.Generator().manual_seed(m)).bfloat16()
        codes, sz = U.group_quantize_tensor(torch.eye(k, dtype=torch.bfloat16), 4, g)
        lut = -(torch.arange(16, dtype=torch.bfloat16) - 8)
        sz[:, :, 0] *= -1.0
        y = run_rm(T, codes, x, sz, lut, g, "any4_global", on_right, inner)
        assert torch.equal(y.cpu(), x)
        # int4 path, un-negated
        codes, sz = U.group_quantize_tensor(torch.eye(k, dtype=torch.bfloat16), 4, g)
        y = run_rm(T, codes, x, sz, None, g, "int4", on_right, inner)
        assert torch.equal(y.cpu(), x)


def test_identity_mx4_and_nan(T):
    import any4_amd.utils as U

    k = 256
    x = torch.randn(7, k, generator=torch.Generator().manual_seed(3)).bfloat16()
    q, e = U.quantize_mx4(torch.eye(k), 32)
    e = e + (torch.arange(k) % 4).to(torch.uint8).unsqueeze(1)
    expect = (x.float() * (2.0 ** (torch.arange(k) % 4).float())).bfloat16()
    for on_right, inner in ((True, 4), (False, 2)):
        y = run_rm(T, q, x, e, None, 32, "mx4", on_right, inner)
        assert torch.equal(y.cpu(), expect)
    # exponent 254 finite, 255 -> NaN for that weight row only (reference test_tinygemm_mx4.py:443-506)
    e2 = e.clone()
    e2[5, :] = 255
    y = run_rm(T, q, x, e2, None, 32, "mx4", True, 4).cpu()
    assert torch.isnan(y[:, 5]).all() and not torch.isnan(y[:, :5]).any() and not torch.isnan(y[:, 6:]).any()


def test_zero_one_weights(T, oracle):
    """0/1 weights, mean-abs-err < 0.1 in the reference (test_tinygemm_any4.py:165-192); here exact to tolerance."""
    import any4_amd.utils as U

    gen = torch.Generator().manual_seed(11)
    n, k, m, g = 64, 1024, 16, 64
    w = torch.randint(0, 2, (n, k), generator=gen).bfloat16()
    x = torch.randn(m, k, generator=gen).bfloat16()
    codes, sz = U.group_quantize_tensor(w, 4, g)
    y = run_rm(T, codes, x, sz, None, g, "int4", True, 4)
    assert_gemm_close(y, x, oracle_weights(oracle, codes, g, "int4", sz, None))
    assert (y.float().cpu() - x.float() @ w.float().t()).abs().mean() < 0.1


@pytest.mark.parametrize("qtype", ["any4_rowwise", "int4"])
def test_gemm_fp16(T, oracle, qtype):
    n, k, g, m = 32, 512, 64, 4
    codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, dtype=torch.float16, seed=2)
    for on_right, inner in ((True, 4), (False, 4)):
        y = run_rm(T, codes, x, qinfo, lut, g, qtype, on_right, inner)
        assert y.dtype == torch.float16
        assert_gemm_close(y, x, oracle_weights(oracle, codes, g, qtype, qinfo, lut, torch.float16), torch.float16)


# ------------------------------------------------------------------------------------------------
# the reference's own outputs (golden fixtures)
# ------------------------------------------------------------------------------------------------

def test_reference_fixture_any4_n1024(T, oracle):
    """BASELINE config 1 inputs through the GPU path vs the reference's CPU dequant-matmul: <= 1e-2 max-abs."""
    g = load_golden("any4_n1024_k1024_g128_seed1234.npz")
    n, k, gs = int(g["n"]), int(g["k"]), int(g["g"])
    nib = g["codes_nib"]
    codes = np.empty((n, k), np.int32)
    codes[:, 0::2] = nib & 0xF
    codes[:, 1::2] = nib >> 4
    codes = torch.from_numpy(codes)
    x = from_bits16(g["x_bits"], torch.bfloat16)
    lut = from_bits16(g["lut_m8_bits"], torch.bfloat16)
    sz = from_bits16(g["sz_bits"], torch.bfloat16)
    y_ref = oracle.bf16_to_f32(g["y_bits"])
    for on_right, inner in ((True, 4), (False, 4)):
        y = run_rm(T, codes, x, sz, lut, gs, "any4_rowwise", on_right, inner)
        err = np.abs(y.float().cpu().numpy() - y_ref).max()
        assert err <= 1e-2, err
        assert_gemm_close(y, x, oracle_weights(oracle, codes, gs, "any4_rowwise", sz, lut))


def test_reference_fixture_anyq_linspace_all_apis(T):
    """tests/test_anyq.py:63-108: the four functional APIs on the captured quantiser output."""
    import tinygemm_lib.functional as F

    g = load_golden("anyq_linspace64.npz")
    for gs in (32, 64):
        p = f"bf16_g{gs}_"
        x = from_bits16(g[p + "x_bits"], torch.bfloat16, DEV)
        y_ref = from_bits16(g[p + "y_bits"], torch.bfloat16, DEV)
        codes = torch.from_numpy(g[p + "codes"].astype(np.int32)).to(DEV)
        lut = from_bits16(g[p + "lut_bits"], torch.bfloat16, DEV)
        sz = from_bits16(g[p + "sz_bits"], torch.bfloat16, DEV)
        for api in ("linear_y_f16TC_x_f16TC_W_any4TC", "linear_y_f16TC_W_any4TC_x_f16TC",
                    "linear_y_f16RM_x_f16RM_W_any4TC", "linear_y_f16RM_W_any4TC_x_f16RM"):
            for w_inner_k in (1, 2, 4):  # 8 needs k % 128 == 0 (k = 64 here), as in the reference
                if not F.valid_tinygemm_kernel_call(api, w_inner_k):
                    continue
                if "f16TC_x" in api or "x_f16TC" in api:
                    y = getattr(F, api)(x, codes, lut, sz, gs, w_inner_k, x_inner_k=1)
                else:
                    y = getattr(F, api)(x, codes, lut, sz, gs, w_inner_k)
                torch.testing.assert_close(y, y_ref)


# ------------------------------------------------------------------------------------------------
# TC-layout ops, f16 weights, modules
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("m", [5, 16, 19])
def test_tc_variants_match_rm(T, oracle, m):
    import tinygemm_lib.functional as F

    n, k, g = 32, 256, 32
    codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_rowwise", seed=m)
    xd, cd, qd, ld = x.to(DEV), codes.to(DEV), qinfo.to(DEV), lut.to(DEV)
    w = oracle_weights(oracle, codes, g, "any4_rowwise", qinfo, lut)
    for x_inner in (1, 2):
        y = F.linear_y_f16TC_W_any4TC_x_f16TC(xd, cd, ld, qd, g, w_inner_k=2, x_inner_k=x_inner)
        assert y.shape == (m, n)
        assert_gemm_close(y, x, w)
    y = F.linear_y_f16TC_x_f16TC_W_any4TC(xd, cd, ld, qd, g, w_inner_k=4, x_inner_k=1)
    assert_gemm_close(y, x, w)
    sz_i, _ = qinfo, None
    y = F.linear_y_f16TC_x_f16TC_W_int4TC(xd, cd, qd, g, w_inner_k=8)
    assert_gemm_close(y, x, oracle_weights(oracle, codes, g, "int4", qinfo, None))
    y = F.linear_y_f16TC_W_int4TC_x_f16TC(xd, cd, qd, g, w_inner_k=1, x_inner_k=2)
    assert_gemm_close(y, x, oracle_weights(oracle, codes, g, "int4", qinfo, None))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_f16_weight_gemm(T, dtype):
    """16-bit weights (reference tests/tinygemm/test_tinygemm_f16.py): identity exact, random assert_close(0.01, 0.1)."""
    import tinygemm_lib.functional as F

    gen = torch.Generator().manual_seed(0)
    for m, n, k in ((5, 64, 256), (16, 48, 1024), (33, 16, 64)):
        x = (torch.randn(m, k, generator=gen) * 0.1).to(dtype).to(DEV)
        w = (torch.randn(n, k, generator=gen) * 0.1).to(dtype).to(DEV)
        y_ref = (x.double() @ w.double().t())
        outs = [F.linear_y_f16RM_x_f16RM_W_f16TC(x, w, 1), F.linear_y_f16RM_x_f16RM_W_f16TC(x, w, 2),
                F.linear_y_f16RM_W_f16TC_x_f16RM(x, w, 1), F.linear_y_f16TC_x_f16TC_W_f16TC(x, w, 2),
                F.linear_y_f16TC_W_f16TC_x_f16TC(x, w, 1)]
        for y in outs:
            assert y.shape == (m, n)
            torch.testing.assert_close(y.double(), y_ref, atol=0.01, rtol=0.1)
    x = torch.randn(19, 256, generator=gen).to(dtype).to(DEV)
    eye = torch.eye(256, dtype=dtype, device=DEV)
    assert torch.equal(F.linear_y_f16RM_x_f16RM_W_f16TC(x, eye, 2), x)
    assert torch.equal(F.linear_y_f16RM_W_f16TC_x_f16RM(x, eye, 1), x)


@pytest.mark.parametrize("kernel", ["linear_y_f16RM_x_f16RM_W_any4TC", "linear_y_f16RM_W_any4TC_x_f16RM"])
@pytest.mark.parametrize("per_row", [True, False])
def test_any4linear_module(T, oracle, kernel, per_row):
    import modules

    n, k, g = 64, 256, 128
    codes, x, qinfo, lut = rand_problem(n, k, g, 6, "any4_rowwise" if per_row else "any4_global", seed=4)
    mod = modules.Any4Linear(k, n, bias=True, device=DEV, dtype=torch.bfloat16, group_size=g, kernel=kernel, per_row=per_row)
    mod.weight.data = codes.to(DEV)
    mod.scales_and_zeros.data = qinfo.to(DEV)
    mod.lut.data = lut.to(DEV)
    bias = torch.randn(n).bfloat16()
    mod.bias.data = bias.to(DEV)
    x3 = x.view(2, 3, k).to(DEV)
    y_unpacked = mod(x3)                       # packs on the fly (reshape_weight=True path)
    mod.reshape_weight(4)
    assert mod.weight_reshaped and mod.weight.dim() == 4
    y = mod(x3)
    assert y.shape == (2, 3, n) and torch.equal(y, y_unpacked)
    w = oracle_weights(oracle, codes, g, "any4_rowwise" if per_row else "any4_global", qinfo, lut)
    y_nobias = torch.ops.tinygemm.tinygemm_y_f16RM_x_f16RM_w_any4TC(
        *( (x.to(DEV), mod.weight) if "x_f16RM_W" in kernel else (mod.weight, x.to(DEV)) ), g, mod.scales_and_zeros, mod.lut,
        "x_f16RM_W" in kernel)
    assert_gemm_close(y_nobias, x, w)
    assert torch.equal(y.view(-1, n), y_nobias + bias.to(DEV))
    # state_dict round trip keeps the packed weight usable
    # state_dict round trip into a FRESH module (unpacked shape, weight_reshaped False): the packed weight stays usable
    import copy

    sd = copy.deepcopy(mod.state_dict())
    mod2 = modules.Any4Linear(k, n, bias=True, device=DEV, dtype=torch.bfloat16, group_size=g, kernel=kernel, per_row=per_row)
    mod2.load_state_dict(sd)
    assert mod2.weight_reshaped and mod2.w_inner_k == 4
    assert torch.equal(mod2(x3), y)


def test_int4linear_module(T, oracle):
    import modules
    import any4_amd.utils as U

    n, k, g = 32, 512, 128
    w = (torch.randn(n, k, generator=torch.Generator().manual_seed(1)) * 0.05).bfloat16()
    codes, sz = U.group_quantize_tensor(w, 4, g)
    x = torch.randn(3, k).bfloat16()
    for kernel in ("linear_y_f16RM_W_int4TC_x_f16RM", "linear_y_f16RM_x_f16RM_W_int4TC", "linear_y_f16TC_x_f16TC_W_int4TC"):
        mod = modules.Int4Linear(k, n, bias=False, device=DEV, dtype=torch.bfloat16, group_size=g, kernel=kernel)
        mod.weight.data = codes.to(DEV)
        mod.scales_and_zeros.data = sz.to(DEV)
        if kernel != "linear_y_f16TC_x_f16TC_W_int4TC":
            mod.reshape_weight()
        y = mod(x.to(DEV))
        assert_gemm_close(y, x, oracle_weights(oracle, codes, g, "int4", sz, None))


# ------------------------------------------------------------------------------------------------
# error behaviour: RuntimeError before launch, like TORCH_CHECK
# ------------------------------------------------------------------------------------------------

def test_errors(T):
    codes = torch.randint(0, 16, (16, 96), dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):
        T.convert_matrix_to_m16n8k16_Bint4_layout(codes, 4)          # k % 64 != 0
    with pytest.raises(RuntimeError):
        T.convert_matrix_to_m16n8k16_Bint4_layout(codes, 3)
    with pytest.raises(RuntimeError):
        T.convert_matrix_to_m16n8k16_Aint4_layout(codes.float(), 1)  # dtype
    codes = torch.randint(0, 16, (16, 128), dtype=torch.int32, device=DEV)
    w2 = T.convert_matrix_to_m16n8k16_Bint4_layout(codes, 4)
    x = torch.randn(2, 128, device=DEV).bfloat16()
    sz = torch.zeros(1, 16, 2, device=DEV).bfloat16()
    lut = torch.zeros(16, 16, device=DEV).bfloat16()
    with pytest.raises(RuntimeError):
        T.tinygemm_y_f16RM_x_f16RM_w_any4TC(x, w2, 48, sz, lut, True)        # bad group
    with pytest.raises(RuntimeError):
        T.tinygemm_y_f16RM_x_f16RM_w_any4TC(x, w2, 128, sz[:, :8], lut, True)  # rows mismatch
    with pytest.raises(RuntimeError):
        T.tinygemm_y_f16RM_x_f16RM_w_any4TC(x, w2, 128, sz, lut.half(), True)  # LUT dtype
    with pytest.raises(RuntimeError):
        T.tinygemm_y_f16RM_x_f16RM_w_any4TC(x.float(), w2, 128, sz, lut, True)
    with pytest.raises(RuntimeError):
        T.tinygemm_y_f16RM_x_f16RM_w_mx4TC(x.half(), w2, 32, torch.zeros(16, 4, dtype=torch.uint8, device=DEV), True)
    with pytest.raises(RuntimeError):
        T.tinygemm_y_f16RM_x_f16RM_w_int8TC(x, w2, 128, sz, True)            # a Bint4 tensor is not a Bint8 one (k mismatch)
    with pytest.raises((RuntimeError, NotImplementedError)):
        T.convert_matrix_to_m16n8k16_Bint4_layout(codes.cpu(), 4)            # no CPU fallback


# ------------------------------------------------------------------------------------------------
# BASELINE full sizes: properties that need no full-size oracle run
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("m,n,k,on_right", [(1, 4096, 4096, True), (8, 4096, 4096, True), (8, 8192, 8192, False)])
def test_full_size_properties(T, oracle, m, n, k, on_right):
    g = 128
    codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_rowwise", seed=99)
    y = run_rm(T, codes, x, qinfo, lut, g, "any4_rowwise", on_right, 4)
    # (1) run-to-run determinism
    y2 = run_rm(T, codes, x, qinfo, lut, g, "any4_rowwise", on_right, 4)
    assert torch.equal(y, y2)
    # (2) A-side and B-side packings of the same weights agree to summation order
    y_other = run_rm(T, codes, x, qinfo, lut, g, "any4_rowwise", not on_right, 4)
    assert (y.float() - y_other.float()).abs().max() <= 2 * float(bf16_ulp(y.float().abs().max().item()))
    # (3) a sampled block of rows against the oracle at full k
    rows = slice(n - 64, n)
    wq = oracle_weights(oracle, codes[rows], g, "any4_rowwise", qinfo[:, rows].contiguous(), lut[rows])
    assert_gemm_close(y[:, rows], x, wq)
    rows = slice(1000, 1032)
    wq = oracle_weights(oracle, codes[rows], g, "any4_rowwise", qinfo[:, rows].contiguous(), lut[rows])
    assert_gemm_close(y[:, rows], x, wq)
    # (4) scaling x by a power of two scales y exactly (linearity in the exactly-representable case)
    y4 = run_rm(T, codes, x * 4, qinfo, lut, g, "any4_rowwise", on_right, 4)
    assert torch.equal(y4, y * 4)
    # (5) zero activations give exactly zero
    y0 = run_rm(T, codes, torch.zeros_like(x), qinfo, lut, g, "any4_rowwise", on_right, 4)
    assert not y0.any()


def test_batched_launch_matches_single(T, oracle):
    """The stacked C-ABI launch (batch > 1) equals per-matrix launches bit for bit."""
    import ctypes

    from any4_amd import _lib

    L = _lib.load()
    n, k, g, m, nb = 64, 1024, 128, 2, 3
    probs = [rand_problem(n, k, g, m, "any4_rowwise", seed=100 + b) for b in range(nb)]
    packed = torch.stack([T.convert_matrix_to_m16n8k16_Bint4_layout(p[0].to(DEV), 4) for p in probs])
    xs = torch.stack([p[1] for p in probs]).to(DEV)
    szs = torch.stack([p[2] for p in probs]).to(DEV)
    luts = torch.stack([p[3] for p in probs]).to(DEV)
    ys = torch.empty(nb, m, n, dtype=torch.bfloat16, device=DEV)
    args = _lib.W4Gemm(x=xs.data_ptr(), w=packed.data_ptr(), qinfo=szs.data_ptr(), lut=luts.data_ptr(), y=ys.data_ptr(),
                       m=m, wrows=n, k=k, group=g, qtype=_lib.TG_Q_ANY4_ROWWISE, dtype=_lib.TG_BF16, w_on_right=1,
                       inner_k_tiles=4, batch=nb, stride_x=xs.stride(0) * 2, stride_w=packed.stride(0) * 4,
                       stride_qinfo=szs.stride(0) * 2, stride_lut=luts.stride(0) * 2, stride_y=ys.stride(0) * 2,
                       numerics=_lib.TG_NUM_REFERENCE)
    _lib.check(L.tg_gemm_w4(ctypes.byref(args), 0, torch.cuda.current_stream().cuda_stream), "batched")
    for b in range(nb):
        y1 = T.tinygemm_y_f16RM_x_f16RM_w_any4TC(xs[b], packed[b], g, szs[b], luts[b], True)
        assert torch.equal(ys[b], y1)


# ------------------------------------------------------------------------------------------------
# streaming kernel ("lane owns group", any4_amd/csrc/w4_gemm_stream.cuh): reached when a launch has
# more than 512 tiles -- here through stacked launches of many small problems
# ------------------------------------------------------------------------------------------------

def _stacked(T, probs, copies, g, qtype, on_right, inner, dtype=torch.bfloat16):
    """Run len(probs) * copies problems in ONE tg_gemm_w4 launch; returns y [B][m][rows]."""
    import ctypes

    from any4_amd import _lib

    L = _lib.load()
    conv = T.convert_matrix_to_m16n8k16_Bint4_layout if on_right else T.convert_matrix_to_m16n8k16_Aint4_layout
    packed = torch.stack([conv(p[0].to(DEV), inner) for p in probs]).repeat(copies, 1, 1, 1, 1).contiguous()
    xs = torch.stack([p[1] for p in probs]).to(DEV).repeat(copies, 1, 1).contiguous()
    qs = torch.stack([p[2] for p in probs]).to(DEV)
    qs = qs.repeat(copies, *([1] * (qs.dim() - 1))).contiguous()
    has_lut = probs[0][3] is not None
    luts = torch.stack([p[3] for p in probs]).to(DEV) if has_lut else None
    if has_lut:
        luts = luts.repeat(copies, *([1] * (luts.dim() - 1))).contiguous()
    B = packed.shape[0]
    m, k = xs.shape[1], xs.shape[2]
    rows = packed.shape[1] * (8 if on_right else 16)
    ys = torch.full((B, m, rows), float("nan"), dtype=dtype, device=DEV)
    qt = {"int4": _lib.TG_Q_INT4, "any4_global": _lib.TG_Q_ANY4_GLOBAL, "any4_rowwise": _lib.TG_Q_ANY4_ROWWISE, "mx4": _lib.TG_Q_MX4}[qtype]
    args = _lib.W4Gemm(x=xs.data_ptr(), w=packed.data_ptr(), qinfo=qs.data_ptr(), lut=(luts.data_ptr() if has_lut else None),
                       y=ys.data_ptr(), m=m, wrows=rows, k=k, group=g, qtype=qt,
                       dtype=_lib.TG_BF16 if dtype == torch.bfloat16 else _lib.TG_F16, w_on_right=1 if on_right else 0,
                       inner_k_tiles=inner, batch=B, stride_x=xs.stride(0) * 2, stride_w=packed.stride(0) * 4,
                       stride_qinfo=qs.stride(0) * qs.element_size(), stride_lut=(luts.stride(0) * 2 if has_lut else 0),
                       stride_y=ys.stride(0) * 2, numerics=_lib.TG_NUM_REFERENCE)
    _lib.check(L.tg_gemm_w4(ctypes.byref(args), 0, torch.cuda.current_stream().cuda_stream), "stacked")
    return ys


@pytest.mark.parametrize("qtype,g", [("any4_rowwise", 128), ("any4_global", 256), ("int4", 128), ("mx4", 128), ("any4_rowwise", 64)])
@pytest.mark.parametrize("on_right,inner", [(True, 2), (True, 4), (True, 8), (False, 1), (False, 2), (False, 4)])
@pytest.mark.parametrize("m", [1, 5, 16])
def test_stream_kernel_vs_oracle(T, oracle, qtype, g, on_right, inner, m):
    """> 512 tiles in one launch -> w4_gemm_stream_kernel (when g >= its unit: 128 on the B side, 64 on the
    A side; otherwise the same launch exercises the split-K-1 path of w4_gemm_kernel)."""
    n, k = 48, 1024  # 3 tiles per problem; ragged last 16-row MFMA tile on the B side (6 x 8 rows)
    nprob, copies = 4, 48  # 4 * 48 * 3 = 576 tiles
    probs = [rand_problem(n, k, g, m, qtype, seed=1000 + 17 * b + m) for b in range(nprob)]
    ys = _stacked(T, probs, copies, g, qtype, on_right, inner)
    assert not torch.isnan(ys.float()).any()
    for b in range(nprob):
        w = oracle_weights(oracle, probs[b][0], g, qtype, probs[b][2], probs[b][3])
        for cpy in (0, copies // 2, copies - 1):
            assert_gemm_close(ys[cpy * nprob + b], probs[b][1], w)
    # every copy of a problem gives the identical result
    assert torch.equal(ys[:nprob], ys[-nprob:])


@pytest.mark.parametrize("k", [128, 192, 640, 1152])
@pytest.mark.parametrize("on_right,inner", [(True, 4), (False, 4), (True, 2), (False, 1)])
def test_stream_kernel_ragged_k(T, oracle, k, on_right, inner):
    """k that is not a multiple of the lanes' walk (4 quarters x units): padding units must contribute zeros."""
    if k % (16 * inner):
        pytest.skip("k not a multiple of 16*innerKTiles")
    n, g, m = 32, 64 if not on_right else 128, 3
    if k % g:
        pytest.skip("group does not divide k")
    probs = [rand_problem(n, k, g, m, "any4_rowwise", seed=k + b) for b in range(2)]
    ys = _stacked(T, probs, 160, g, "any4_rowwise", on_right, inner)
    for b in range(2):
        w = oracle_weights(oracle, probs[b][0], g, "any4_rowwise", probs[b][2], probs[b][3])
        assert_gemm_close(ys[b], probs[b][1], w)
        assert_gemm_close(ys[-2 + b], probs[b][1], w)


def test_stream_kernel_identity_and_fp16(T, oracle):
    import any4_amd.utils as U

    k, g = 512, 128
    x = torch.randn(7, k, generator=torch.Generator().manual_seed(5)).bfloat16()
    codes, sz = U.group_quantize_tensor(torch.eye(k, dtype=torch.bfloat16), 4, g)
    lut = -(torch.arange(16, dtype=torch.bfloat16) - 8)
    sz[:, :, 0] *= -1.0
    for on_right, inner in ((True, 4), (False, 4)):
        ys = _stacked(T, [(codes, x, sz, lut)], 20, g, "any4_global", on_right, inner)  # 32 tiles * 20 = 640
        assert torch.equal(ys[0].cpu(), x) and torch.equal(ys[-1].cpu(), x)
    probs = [rand_problem(64, 512, 128, 4, "any4_rowwise", dtype=torch.float16, seed=b) for b in range(2)]
    ys = _stacked(T, probs, 80, 128, "any4_rowwise", True, 4, dtype=torch.float16)
    for b in range(2):
        w = oracle_weights(oracle, probs[b][0], 128, "any4_rowwise", probs[b][2], probs[b][3], torch.float16)
        assert_gemm_close(ys[b], probs[b][1], w, torch.float16)


def test_stream_kernel_mx4_nan_row(T):
    import any4_amd.utils as U

    k = 256
    x = torch.randn(2, k, generator=torch.Generator().manual_seed(3)).bfloat16()
    q, e = U.quantize_mx4(torch.eye(k), 128)
    e[5, :] = 255
    ys = _stacked(T, [(q, x, e, None)], 40, 128, "mx4", True, 4).cpu()  # 16 tiles * 40 = 640
    assert torch.isnan(ys[:, :, 5]).all()
    keep = [c for c in range(k) if c != 5]
    assert torch.equal(ys[0][:, keep], x[:, keep]) and torch.equal(ys[-1][:, keep], x[:, keep])


@pytest.mark.parametrize("k,expect_sk", [(1024, 2), (2048, 4), (4096, 8)])
@pytest.mark.parametrize("on_right,inner", [(True, 4), (False, 4), (True, 8), (False, 1)])
@pytest.mark.parametrize("qtype", ["any4_rowwise", "int4"])
def test_stream_private_slab_splitk(T, oracle, k, expect_sk, on_right, inner, qtype):
    """m = 1 single launches: streaming kernel, private X slabs, the `expect_sk` waves of a workgroup are the
    k-slices of one tile (reduction in LDS, fixed order)."""
    n, g = 80, 128
    codes, x, qinfo, lut = rand_problem(n, k, g, 1, qtype, seed=k + inner)
    y = run_rm(T, codes, x, qinfo, lut, g, qtype, on_right, inner)
    assert_gemm_close(y[:, :n], x, oracle_weights(oracle, codes, g, qtype, qinfo, lut))
    y2 = run_rm(T, codes, x, qinfo, lut, g, qtype, on_right, inner)
    assert torch.equal(y, y2)  # deterministic


@pytest.mark.parametrize("on_right,inner,m", [(True, 4, 8), (True, 2, 11), (False, 4, 16), (False, 2, 16)])
def test_stream_resident_x(T, oracle, on_right, inner, m):
    """>= 8192 wave-tiles with m >= 8 (B side) / 16 (A side): the activation block is staged once per 16-wave
    workgroup (XRES variant of w4_gemm_stream_kernel)."""
    n, k, g = 48, 1024, 128
    nprob, copies = 4, 700  # 4 * 700 * 3 = 8400 tiles
    probs = [rand_problem(n, k, g, m, "any4_rowwise", seed=77 + 3 * b + m) for b in range(nprob)]
    ys = _stacked(T, probs, copies, g, "any4_rowwise", on_right, inner)
    assert not torch.isnan(ys.float()).any()
    for b in range(nprob):
        w = oracle_weights(oracle, probs[b][0], g, "any4_rowwise", probs[b][2], probs[b][3])
        for cpy in (0, copies - 1):
            assert_gemm_close(ys[cpy * nprob + b], probs[b][1], w)
    assert torch.equal(ys[:nprob], ys[-nprob:])


# ------------------------------------------------------------------------------------------------
# int8 weights: packers bit-exact, GEMM vs the oracle, identity bit-exact
# ------------------------------------------------------------------------------------------------

def _rand_int8_problem(n, k, g, m, dtype=torch.bfloat16, seed=0):
    gen = torch.Generator().manual_seed(seed)
    codes = torch.randint(0, 256, (n, k), dtype=torch.int32, generator=gen)
    x = torch.randn(m, k, generator=gen).to(dtype)
    scales = (torch.rand(k // g, n, generator=gen) * 0.002 + 0.0005).to(dtype)
    zeros = (torch.randn(k // g, n, generator=gen) * 0.01).to(dtype)
    return codes, x, torch.stack([scales, zeros], dim=2).contiguous()


@pytest.mark.parametrize("n,k", [(8, 64), (19, 256), (100, 640), (256, 4096)])
def test_pack_int8_bit_exact(T, oracle, n, k):
    codes = torch.randint(0, 256, (n, k), dtype=torch.int32, generator=torch.Generator().manual_seed(n + k))
    codes[0, 0] = 0x1234567  # the packers OR the shifted 32-bit inputs without masking (ConvertB.cu:404)
    for inner in (1, 2, 4):
        if k % (16 * inner) == 0:
            got = T.convert_matrix_to_m16n8k16_Bint8_layout(codes.to(DEV), inner).cpu().numpy()
            assert np.array_equal(got, oracle.pack_Bint8(codes.numpy(), inner)), ("B", inner)
    for inner in (1, 2):
        got = T.convert_matrix_to_m16n8k16_Aint8_layout(codes.to(DEV), inner).cpu().numpy()
        assert np.array_equal(got, oracle.pack_Aint8(codes.numpy(), inner)), ("A", inner)


@pytest.mark.parametrize("on_right,inner", [(True, 1), (True, 2), (True, 4), (False, 1), (False, 2)])
@pytest.mark.parametrize("g", [32, 128])
@pytest.mark.parametrize("m,n,k", [(1, 64, 256), (5, 48, 1024), (16, 200, 512), (33, 24, 4096)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_int8_vs_oracle(T, oracle, on_right, inner, g, m, n, k, dtype):
    if k % (16 * inner) or (dtype == torch.float16 and (g, m) != (128, 5)):
        pytest.skip("covered elsewhere")
    codes, x, sz = _rand_int8_problem(n, k, g, m, dtype, seed=n + k + m)
    d = lambda t: t.to(DEV)
    if on_right:
        w2 = T.convert_matrix_to_m16n8k16_Bint8_layout(d(codes), inner)
        n_pad = w2.size(0) * 8
    else:
        w2 = T.convert_matrix_to_m16n8k16_Aint8_layout(d(codes), inner)
        n_pad = w2.size(0) * 16
    szp = torch.zeros(k // g, n_pad, 2, dtype=dtype)
    szp[:, :n] = sz
    y = T.tinygemm_y_f16RM_x_f16RM_w_int8TC(d(x), w2, g, d(szp), True) if on_right else \
        T.tinygemm_y_f16RM_x_f16RM_w_int8TC(w2, d(x), g, d(szp), False)
    assert y.shape == (m, n_pad)
    w = oracle.dequant(codes.numpy(), g, oracle.Q_INT8, bits16(sz), None, oracle.BF16 if dtype == torch.bfloat16 else oracle.F16)
    assert_gemm_close(y[:, :n], x, w, dtype)
    assert (y[:, n:] == 0).all()  # padded rows: zero codes... with scale = zero = 0 they dequantise to 0


@pytest.mark.parametrize("on_right", [True, False])
@pytest.mark.parametrize("g", [64, 128, 256])
@pytest.mark.parametrize("m,n,k,dtype", [(17, 200, 512, torch.bfloat16), (7, 200, 512, torch.bfloat16), (33, 264, 2048, torch.bfloat16), (64, 4096, 2048, torch.float16), (130, 1008, 4096, torch.bfloat16),
                                         (512, 528, 1024, torch.float16), (100, 48, 192, torch.bfloat16)])
def test_gemm_int8_many_rows_on_the_tile_gemm(T, oracle, on_right, g, m, n, k, dtype):
    """tinygemm_y_f16RM_x_f16RM_w_int8TC (TinyGemm_int8.cu:216-399) from 17 activation rows on the tile GEMM's int8 flavour (innerKTiles 2, the
    packing Int8Linear defaults to; groups of 64 or more): w = RNE16(fma(byte - 128, scale, zero)) computed by the dequantising waves, unsplit
    and split-K launches, both operand sides, ragged rows, a k of one step and a half (192), fused bias."""
    from any4_amd import ops

    if k % g:
        pytest.skip("group does not divide k")
    codes, x, sz = _rand_int8_problem(n, k, g, m, dtype, seed=n + k + m)
    d = lambda t: t.to(DEV)
    if on_right:
        w2 = T.convert_matrix_to_m16n8k16_Bint8_layout(d(codes), 2)
        n_pad = w2.size(0) * 8
    else:
        w2 = T.convert_matrix_to_m16n8k16_Aint8_layout(d(codes), 2)
        n_pad = w2.size(0) * 16
    szp = torch.zeros(k // g, n_pad, 2, dtype=dtype)
    szp[:, :n] = sz
    run = lambda: T.tinygemm_y_f16RM_x_f16RM_w_int8TC(d(x), w2, g, d(szp), True) if on_right else T.tinygemm_y_f16RM_x_f16RM_w_int8TC(w2, d(x), g, d(szp), False)
    y = run()
    assert y.shape == (m, n_pad)
    w = oracle.dequant(codes.numpy(), g, oracle.Q_INT8, bits16(sz), None, oracle.BF16 if dtype == torch.bfloat16 else oracle.F16)
    assert_gemm_close(y[:, :n], x, w, dtype)
    assert (y[:, n:] == 0).all()
    assert torch.equal(y, run())
    bias = torch.randn(n_pad, generator=torch.Generator().manual_seed(5)).to(dtype).to(DEV)
    with ops.fused_bias(bias) as fb:
        yb = run()
    assert fb.consumed and torch.equal(yb, y + bias)


@pytest.mark.parametrize("api", ["RM_right", "RM_left", "TC_right", "TC_left"])
@pytest.mark.parametrize("k,g,inner", [(256, 32, 1), (1024, 64, 2), (2048, 256, 2), (1024, 128, 4)])
def test_int8_identity_bit_exact(T, api, k, g, inner):
    """test_tinygemm_int8.py:23-50: w = eye(k) through group_quantize_tensor(n_bit=8) -> y == x bit for bit (bf16)."""
    import tinygemm_lib.functional as F
    from tinygemm_lib.utils import group_quantize_tensor

    if api.endswith("left") and inner == 4:
        pytest.skip("Aint8 has innerKTiles 1, 2")
    x = torch.randn(29, k, generator=torch.Generator().manual_seed(k)).to(torch.bfloat16).to(DEV)
    w = torch.eye(k, dtype=torch.bfloat16, device=DEV)
    w_int32, sz = group_quantize_tensor(w, n_bit=8, q_group_size=g)
    fn = {"RM_right": F.linear_y_f16RM_x_f16RM_W_int8TC, "RM_left": F.linear_y_f16RM_W_int8TC_x_f16RM,
          "TC_right": F.linear_y_f16TC_x_f16TC_W_int8TC, "TC_left": F.linear_y_f16TC_W_int8TC_x_f16TC}[api]
    y = fn(x, w_int32, sz, g, inner)
    assert torch.equal(y, x @ w.t())


def test_int8linear_module(T, oracle):
    import modules

    n, k, g = 96, 512, 128
    codes, x, sz = _rand_int8_problem(n, k, g, 7, seed=5)
    for kernel in ("linear_y_f16RM_x_f16RM_W_int8TC", "linear_y_f16RM_W_int8TC_x_f16RM"):
        mod = modules.Int8Linear(k, n, bias=True, device=DEV, dtype=torch.bfloat16, group_size=g, kernel=kernel)
        mod.weight.data = codes.to(DEV)
        mod.scales_and_zeros.data = sz.to(DEV)
        mod.bias.data = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
        mod.reshape_weight()
        assert mod.weight.dim() == 4 and mod.weight_reshaped
        y = mod(x.to(DEV).view(1, 7, k))
        assert y.shape == (1, 7, n)
        assert_gemm_close(y.view(7, n), x, oracle.dequant(codes.numpy(), g, oracle.Q_INT8, bits16(sz), None))


@pytest.mark.parametrize("qtype,g,inner,dtype", [("any4_rowwise", 128, 4, torch.bfloat16), ("int4", 32, 2, torch.float16), ("any4_global", 256, 8, torch.bfloat16),
                                                 ("any4_rowwise", 64, 4, torch.float16)])
def test_dequant_w4_is_the_reference_weights_bit_for_bit(T, oracle, qtype, g, inner, dtype):
    """tg_dequant_w4 (what a call with ~100 activation rows or more multiplies with the GEMM library): every element equal to the
    oracle's reference-faithful weight RNE16(fma(lut[code], scale, zero)) (MatrixLayoutB.cuh:1042-1046); the op on 130 rows within
    the GEMM tolerance of the oracle, on both operand sides."""
    import any4_amd
    from any4_amd import ops

    n, k, m = 144, 1024, 130
    codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, dtype=dtype, seed=n + g + inner)
    packed = T.convert_matrix_to_m16n8k16_Bint4_layout(codes.to(DEV), inner)
    wrows = packed.shape[0] * 8
    w = ops.dequant_w4(packed, qinfo.to(DEV), None if lut is None else lut.to(DEV), g, {"int4": 0, "any4_global": 1, "any4_rowwise": 2}[qtype], k, inner, wrows)
    want = oracle_weights(oracle, codes, g, qtype, qinfo, lut, dtype)
    assert np.array_equal(bits16(w[:n].cpu()), np.asarray(want).reshape(n, k))
    # a PANEL of weight rows (what the opt-in library route dequantises at a time): the same bits
    if wrows >= 48:
        wp = ops.dequant_w4(packed, qinfo.to(DEV), None if lut is None else lut.to(DEV), g, {"int4": 0, "any4_global": 1, "any4_rowwise": 2}[qtype], k, inner, wrows, rows=(16, 48))
        assert torch.equal(wp, w[16:48])
    y = run_rm(T, codes, x, qinfo, lut, g, qtype, True, inner)
    assert_gemm_close(y[:, :n], x, want, dtype)
    if inner <= 4:
        with any4_amd.weight_format("native"):
            y2 = run_rm(T, codes, x, qinfo, lut, g, qtype, False, inner)
        assert_gemm_close(y2[:, :n], x, want, dtype)
    # the OPT-IN library route (dequantise in bounded panels + the GEMM library) gives the same result within the GEMM tolerance
    saved = ops._LARGE_M
    try:
        ops._LARGE_M = 65
        os.environ["ANY4_DEQUANT_PANEL_MB"] = "0.1"   # 0.1 MB / (2 k) = 48 weight rows per panel at k = 1024: three panels
        y3 = run_rm(T, codes, x, qinfo, lut, g, qtype, True, inner)
        assert_gemm_close(y3[:, :n], x, want, dtype)
        if k % 512 == 0:
            yp = ops._library_gemm_w4(x.to(DEV), packed, qinfo.to(DEV), None if lut is None else lut.to(DEV), g, {"int4": 0, "any4_global": 1, "any4_rowwise": 2}[qtype], k, inner, wrows)
            assert_gemm_close(yp[:, :n].cpu(), x, want, dtype)
    finally:
        ops._LARGE_M = saved
        os.environ.pop("ANY4_DEQUANT_PANEL_MB", None)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("qtype,g", [("any4_rowwise", 128), ("any4_rowwise", 32), ("int4", 64), ("any4_global", 256)])
@pytest.mark.parametrize("m,n,k", [(65, 200, 512), (128, 64, 64), (130, 1008, 1024), (512, 528, 2048), (33, 200, 2048), (64, 528, 2048), (48, 1008, 4096), (130, 1008, 4096), (256, 2056, 2048), (17, 72, 2048), (20, 4096, 2048)])
def test_tile_gemm_many_rows_against_oracle(T, oracle, dtype, qtype, g, m, n, k):
    """TinyGemmImpl.cuh:379-392: the reference's one kernel walks any m.  Here more than 64 activation rows run w4_gemm_tile_kernel (plan
    'tile'): an LDS-tiled MFMA GEMM whose weight tile is dequantised on the way in -- from 33 rows on as a split-K launch (2 / 4 / 8 k ranges,
    f32 partial tiles in the op's workspace, summed in split order) when the tiles do not fill the chip.  Ragged m (not a multiple of 128), ragged weight rows
    (not a multiple of 64 / 128), k of one step and of many, every group size, both 16-bit types, both numerics settings, both operand
    sides, with and without a fused bias -- against the oracle's reference-faithful weights at the GEMM tolerance."""
    import any4_amd
    from any4_amd import ops

    if g > k:
        pytest.skip("group larger than k")
    QT = {"int4": 0, "any4_global": 1, "any4_rowwise": 2}
    codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, dtype=dtype, seed=m + n + g)
    want = oracle_weights(oracle, codes, g, qtype, qinfo, lut, dtype)
    assert ops.gemm_w4_plan(m, -(-n // 8) * 8, k, g, QT[qtype], True, 4, dtype=dtype) == "tile"
    for num in ("fast", "reference"):
        with any4_amd.numerics(num):
            y = run_rm(T, codes, x, qinfo, lut, g, qtype, True, 4)
            assert y.shape[0] == m and torch.isfinite(y.float()).all()
            assert_gemm_close(y[:, :n], x, want, dtype)
    if n % 16 == 0:   # (weights on the left: the quantisation info has one row per 16-row-padded weight row)
        with any4_amd.weight_format("native"):
            y2 = run_rm(T, codes, x, qinfo, lut, g, qtype, False, 4)
        assert_gemm_close(y2[:, :n], x, want, dtype)
        assert torch.equal(y2[:, :n], y[:, :n])        # the same words, the same kernel: the same bits on either operand side
    # a fused bias: bit-identical to the separate rounded add of the reference module (modules.py:221-222)
    wrows = y.shape[1]
    bias = torch.randn(wrows, generator=torch.Generator().manual_seed(5)).to(dtype).to(DEV)
    with ops.fused_bias(bias) as fb:
        yb = run_rm(T, codes, x, qinfo, lut, g, qtype, True, 4)
    assert fb.consumed and torch.equal(yb, y + bias)


@pytest.mark.parametrize("m,n,k", [(65, 200, 512), (130, 1008, 1024), (512, 528, 2048), (40, 264, 2048), (100, 4096, 4096)])
def test_tile_gemm_many_rows_mx4(T, oracle, m, n, k):
    """mx4 (bf16, groups of 32: TinyGemm_int4.cu:758) at many rows on the tile GEMM: the table of a (row, group) holds fp4[code] * 2^(e - 127),
    exact in bf16 -- the reference's weights bit for bit (Dequantization.cuh:331-346, MatrixLayoutB.cuh:1086-1088); unsplit and split-K
    launches; an exponent of 255 makes exactly its rows' outputs NaN."""
    from any4_amd import ops

    codes, x, qinfo, lut = rand_problem(n, k, 32, m, "mx4", seed=m + n)
    want = oracle_weights(oracle, codes, 32, "mx4", qinfo, None, torch.bfloat16)
    assert ops.gemm_w4_plan(m, -(-n // 8) * 8, k, 32, 3, True, 4) == "tile"
    y = run_rm(T, codes, x, qinfo, None, 32, "mx4", True, 4)
    assert y.shape[0] == m and torch.isfinite(y.float()).all()
    assert_gemm_close(y[:, :n], x, want, torch.bfloat16)
    bias = torch.randn(y.shape[1], generator=torch.Generator().manual_seed(5)).bfloat16().to(DEV)
    with ops.fused_bias(bias) as fb:
        yb = run_rm(T, codes, x, qinfo, None, 32, "mx4", True, 4)
    assert fb.consumed and torch.equal(yb, y + bias)
    q2 = qinfo.clone()
    q2[3, (k // 32) - 1] = 255          # weight row 3: NaN in its last group
    q2[n - 1, 0] = 255
    yn = run_rm(T, codes, x, q2, None, 32, "mx4", True, 4)
    nan_rows = torch.isnan(yn.float()).all(dim=0).nonzero().flatten().tolist()
    assert nan_rows == [3, n - 1]
    keep = [r for r in range(n) if r not in (3, n - 1)]
    assert torch.equal(yn[:, keep], y[:, keep])


def test_tile_gemm_without_a_workspace(T, oracle, monkeypatch):
    """The C ABI's workspace is optional (include/tinygemm_hip.h): without one the tile GEMM runs unsplit from 65 rows (below: 16-row passes).
    Same weights, another summation order: both within the GEMM tolerance of the oracle; and the split launch is deterministic."""
    from any4_amd import ops

    m, n, k, g = 128, 1008, 4096, 128
    codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_rowwise", seed=77)
    want = oracle_weights(oracle, codes, g, "any4_rowwise", qinfo, lut, torch.bfloat16)
    y_split = run_rm(T, codes, x, qinfo, lut, g, "any4_rowwise", True, 4)
    assert torch.equal(y_split, run_rm(T, codes, x, qinfo, lut, g, "any4_rowwise", True, 4))
    saved = dict(ops._WS_BYTES)
    ops._WS_BYTES.clear()
    monkeypatch.setattr(ops._L, "tg_gemm_w4_workspace_bytes", lambda a: 0, raising=False)
    try:
        y_plain = run_rm(T, codes, x, qinfo, lut, g, "any4_rowwise", True, 4)
        y48 = run_rm(T, codes, x[:48].contiguous(), qinfo, lut, g, "any4_rowwise", True, 4)      # (no workspace, 48 rows: three 16-row passes)
    finally:
        ops._WS_BYTES.clear()
        ops._WS_BYTES.update(saved)
    assert_gemm_close(y_split[:, :n], x, want, torch.bfloat16)
    assert_gemm_close(y_plain[:, :n], x, want, torch.bfloat16)
    assert_gemm_close(y48[:, :n], x[:48], want, torch.bfloat16)


@pytest.mark.parametrize("on_right,inner", [(True, 4), (False, 2)])
@pytest.mark.parametrize("m,copies", [(33, 50), (20, 700)])
def test_stream_kernel_several_column_tiles(T, oracle, on_right, inner, m, copies):
    """m > 16: the launch has several 16-row activation tiles (grid.y); the last one is ragged (33 = 16 + 16 + 1,
    20 = 16 + 4), so workgroups of one launch see different numbers of live MFMA columns.  copies = 50: shared-slab
    variant with split-K; copies = 700 (>= 8192 wave-tiles): resident-X variant."""
    n, k, g = 48, 1024, 128
    probs = [rand_problem(n, k, g, m, "any4_rowwise", seed=500 + 7 * b + m) for b in range(4)]
    ys = _stacked(T, probs, copies, g, "any4_rowwise", on_right, inner)
    assert not torch.isnan(ys.float()).any()
    for b in range(4):
        w = oracle_weights(oracle, probs[b][0], g, "any4_rowwise", probs[b][2], probs[b][3])
        assert_gemm_close(ys[b], probs[b][1], w)
        assert_gemm_close(ys[(copies - 1) * 4 + b], probs[b][1], w)


@pytest.mark.parametrize("k", [1024, 4096])
def test_stream_private_slab_splitk_fp16(T, oracle, k):
    n, g = 80, 128
    for on_right, inner in ((True, 4), (False, 4)):
        codes, x, qinfo, lut = rand_problem(n, k, g, 1, "any4_rowwise", dtype=torch.float16, seed=k)
        y = run_rm(T, codes, x, qinfo, lut, g, "any4_rowwise", on_right, inner)
        assert_gemm_close(y[:, :n], x, oracle_weights(oracle, codes, g, "any4_rowwise", qinfo, lut, torch.float16), torch.float16)
