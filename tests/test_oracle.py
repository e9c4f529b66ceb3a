"""CPU suite, part 1: pin the oracle (oracle/tinygemm_oracle.c) against the golden vectors captured
from the reference's Python (tests/golden/make_golden.py) and against the known-answer patterns the
reference's own kernel tests use (identity / 0-1 / random LUT).  No GPU, no HIP compute calls."""
import hashlib

import numpy as np
import pytest
import torch

from tests.conftest import bf16_ulp, bits16, from_bits16, load_golden

import any4_amd.utils as U


# ---------------------------------------------------------------- packing (P1 / P2 / P3)

@pytest.mark.parametrize("inner", [2, 4, 8])
@pytest.mark.parametrize("n,k", [(8, 128), (19, 256), (64, 1024)])
def test_pack_Bint4_roundtrip_and_formula(oracle, inner, n, k):
    rng = np.random.default_rng(n * k + inner)
    codes = rng.integers(0, 16, (n, k), dtype=np.int32)
    packed = oracle.pack_Bint4(codes, inner)
    assert packed.shape == ((n + 7) // 8, k // (16 * inner), 32, inner // 2)
    assert np.array_equal(oracle.unpack_Bint4(packed, n, k), codes)
    # spot-check words against the formula of TinyGemmConvertB.cu:276-306 written out independently
    for (nt, ks, t, j) in [(0, 0, 0, 0), (packed.shape[0] - 1, packed.shape[1] - 1, 31, inner // 2 - 1), (0, 0, 13, 0)]:
        n0 = nt * 8 + t // 4
        base = (ks * inner + 2 * j) * 16 + (t % 4) * 2
        kk = [base, base + 1, base + 8, base + 9, base + 16, base + 17, base + 24, base + 25]
        v = [int(codes[n0, c]) if n0 < n else 0 for c in kk]
        word = (v[7] << 28) | (v[5] << 24) | (v[3] << 20) | (v[1] << 16) | (v[6] << 12) | (v[4] << 8) | (v[2] << 4) | v[0]
        assert np.uint32(packed[nt, ks, t, j]) == np.uint32(word)


@pytest.mark.parametrize("inner", [1, 2, 4])
@pytest.mark.parametrize("m,k", [(16, 64), (21, 96), (48, 512)])
def test_pack_Aint4_roundtrip_and_formula(oracle, inner, m, k):
    rng = np.random.default_rng(m * k + inner)
    codes = rng.integers(0, 16, (m, k), dtype=np.int32)
    packed = oracle.pack_Aint4(codes, inner)
    assert packed.shape == ((m + 15) // 16, -(-k // (16 * inner)), 32, inner)
    assert np.array_equal(oracle.unpack_Aint4(packed, m, k), codes)
    mt, ks, t, i = 0, 0, 6, inner - 1
    m0, m1 = t // 4, t // 4 + 8
    k0 = (ks * inner + i) * 16 + (t % 4) * 2

    def at(r, c):
        return int(codes[r, c]) if (r < m and c < k) else 0

    v = [at(m0, k0), at(m0, k0 + 1), at(m1, k0), at(m1, k0 + 1), at(m0, k0 + 8), at(m0, k0 + 9), at(m1, k0 + 8), at(m1, k0 + 9)]
    word = (v[7] << 28) | (v[5] << 24) | (v[3] << 20) | (v[1] << 16) | (v[6] << 12) | (v[4] << 8) | (v[2] << 4) | v[0]
    assert np.uint32(packed[mt, ks, t, i]) == np.uint32(word)


@pytest.mark.parametrize("m,k", [(16, 16), (5, 40), (33, 100), (48, 256)])
def test_layout16_roundtrips(oracle, m, k):
    """from_X(to_X(t)) == t, exact-tile and ragged sizes (reference tests/tinygemm/test_tinygemm_convert.py)."""
    rng = np.random.default_rng(7)
    x = rng.integers(0, 1 << 16, (m, k)).astype(np.uint16)
    assert np.array_equal(oracle.from_A16(oracle.to_A16(x), m, k), x)
    for inner in (1, 2):
        assert np.array_equal(oracle.from_B16(oracle.to_B16(x, inner), m, k), x)


# ---------------------------------------------------------------- group quantiser (Q1) and mx4 (Q2)

def test_group_quantize_matches_reference_fixture():
    g = load_golden("group_quant.npz")
    w = from_bits16(g["w_bits"], torch.bfloat16)
    for gs in (32, 64, 128, 256):
        codes, sz = U.group_quantize_tensor(w, 4, gs)
        assert np.array_equal(codes.numpy().astype(np.uint8), g[f"codes_g{gs}"])
        assert np.array_equal(bits16(sz), g[f"sz_bits_g{gs}"])
    codes, sz = U.group_quantize_tensor(torch.eye(256, dtype=torch.bfloat16), 4, 64)
    assert np.array_equal(codes.numpy().astype(np.uint8), g["eye256_codes_g64"])
    assert np.array_equal(bits16(sz), g["eye256_sz_bits_g64"])


def test_group_quantize_int8_matches_reference_fixture(oracle):
    """n_bit = 8 (inputs of the int8 kernels): codes 0..255, zero = min + 128 scale."""
    g = load_golden("group_quant_int8.npz")
    w = from_bits16(g["w_bits"], torch.bfloat16)
    for gs in (32, 128):
        codes, sz = U.group_quantize_tensor(w, 8, gs)
        assert np.array_equal(codes.numpy().astype(np.uint8), g[f"codes_g{gs}"]) and int(codes.max()) <= 255
        assert np.array_equal(bits16(sz), g[f"sz_bits_g{gs}"])
        # oracle dequant (byte - 128) * scale + zero reproduces w to half a grid step
        wd = from_bits16(oracle.dequant(codes.numpy(), gs, oracle.Q_INT8, bits16(sz), None), torch.bfloat16).float()
        scale = from_bits16(bits16(sz), torch.bfloat16).float()[:, :, 0].t().repeat_interleave(gs, dim=1)
        assert ((wd - w.float()).abs() <= scale * 0.5 + 2.0 ** -8 * w.float().abs().max()).all()
    codes, sz = U.group_quantize_tensor(torch.eye(256, dtype=torch.bfloat16), 8, 64)
    assert np.array_equal(codes.numpy().astype(np.uint8), g["eye256_codes_g64"])
    assert np.array_equal(bits16(sz), g["eye256_sz_bits_g64"])
    # the identity known-answer case: dequantised eye is exactly eye (test_tinygemm_int8.py:23-50 relies on it)
    wd = from_bits16(oracle.dequant(codes.numpy(), 64, oracle.Q_INT8, bits16(sz), None), torch.bfloat16)
    assert torch.equal(wd, torch.eye(256, dtype=torch.bfloat16))


def test_pack_int8_layout_formulas(oracle):
    """Bint8 / Aint8 words against the index formulas of TinyGemmConvertB.cu:366-411 / TinyGemmConvertA.cu:337-397."""
    rng = np.random.default_rng(3)
    n, k = 20, 192
    codes = rng.integers(0, 256, (n, k), dtype=np.int32)
    for inner in (1, 2, 4):
        out = oracle.pack_Bint8(codes, inner).view(np.uint32)
        assert out.shape == ((n + 7) // 8, k // (16 * inner), 32, inner)
        for (a, b, t, j) in [(0, 0, 0, 0), (1, 1, 13, inner - 1), (2, k // (16 * inner) - 1, 31, 0)]:
            n0, kb = a * 8 + t // 4, (b * inner + j) * 16 + (t % 4) * 2
            v = [int(codes[n0, kk]) if n0 < n else 0 for kk in (kb, kb + 1, kb + 8, kb + 9)]
            assert out[a, b, t, j] == ((v[3] << 24) | (v[1] << 16) | (v[2] << 8) | v[0])
    for inner in (1, 2):
        out = oracle.pack_Aint8(codes, inner).view(np.uint32)
        assert out.shape == ((n + 15) // 16, (k // 16 + inner - 1) // inner, 32, 2 * inner)
        for (a, kt, t) in [(0, 0, 0), (1, 5, 9), (1, k // 16 - 1, 31)]:
            m0, m1, k0 = a * 16 + t // 4, a * 16 + t // 4 + 8, kt * 16 + (t % 4) * 2
            gv = lambda mm, kk: int(codes[mm, kk]) if mm < n else 0
            w0 = (gv(m1, k0 + 1) << 24) | (gv(m0, k0 + 1) << 16) | (gv(m1, k0) << 8) | gv(m0, k0)
            w1 = (gv(m1, k0 + 9) << 24) | (gv(m0, k0 + 9) << 16) | (gv(m1, k0 + 8) << 8) | gv(m0, k0 + 8)
            assert out[a, kt // inner, t, (kt % inner) * 2] == w0 and out[a, kt // inner, t, (kt % inner) * 2 + 1] == w1
    with pytest.raises(Exception):
        oracle.pack_Bint8(codes, 8)


def test_mx4_quantizer_matches_reference_fixture(oracle):
    g = load_golden("mx4.npz")
    w = torch.from_numpy(g["w"])
    q, e = U.quantize_mx4(w, 32)
    assert np.array_equal(q.numpy().astype(np.uint8), g["q"])
    assert np.array_equal(e.numpy(), g["e"])
    deq = U.dequantize_mx4(q, e)
    assert np.array_equal(deq.numpy(), g["deq"])
    # the oracle's mx4 dequant (bf16 result) agrees with the reference's float dequant wherever that is bf16-exact
    wq = oracle.dequant(q.numpy(), 32, oracle.Q_MX4, e.numpy())
    assert np.array_equal(oracle.bf16_to_f32(wq), g["deq"])
    q, e = U.quantize_mx4(torch.eye(128), 32)
    assert np.array_equal(q.numpy().astype(np.uint8), g["eye128_q"]) and np.array_equal(e.numpy(), g["eye128_e"])


# ---------------------------------------------------------------- dequant (D1-D5)

@pytest.mark.parametrize("rowwise", [False, True])
@pytest.mark.parametrize("group", [32, 128])
def test_dequant_equals_torch_addcmul(oracle, rowwise, group):
    """The reference's own random-data oracle: addcmul(zeros, lut[codes], scales) on bf16 CPU tensors
    (tests/tinygemm/test_tinygemm_any4.py:233-236)."""
    torch.manual_seed(5)
    n, k = 24, 512
    codes = torch.randint(0, 16, (n, k), dtype=torch.int32)
    lut = torch.randn(n, 16).bfloat16() if rowwise else torch.randn(16).bfloat16()
    scales = (torch.rand(k // group, n) * 0.02 + 0.005).bfloat16()
    zeros = (torch.randn(k // group, n) * 0.01).bfloat16()
    sz = torch.stack([scales, zeros], dim=2).contiguous()
    s_full, z_full = U.extract_scales_and_zeros(sz, (n, k), group)
    vals = torch.gather(lut, 1, codes.long()) if rowwise else lut[codes.long()]
    expect = torch.addcmul(z_full, vals, s_full)
    got = oracle.dequant(codes.numpy(), group, oracle.Q_ANY4_ROWWISE if rowwise else oracle.Q_ANY4_GLOBAL,
                         bits16(sz), bits16(lut))
    assert np.array_equal(got, bits16(expect))


def test_dequant_int4_is_any4_with_uniform_lut(oracle):
    rng = np.random.default_rng(0)
    n, k, g = 16, 256, 64
    codes = rng.integers(0, 16, (n, k), dtype=np.int32)
    sz = bits16((torch.randn(k // g, n, 2) * 0.1).bfloat16())
    lut = bits16((torch.arange(16) - 8).bfloat16())
    a = oracle.dequant(codes, g, oracle.Q_INT4, sz)
    b = oracle.dequant(codes, g, oracle.Q_ANY4_GLOBAL, sz, lut)
    assert np.array_equal(a, b)


def test_dequant_int4_debug_order(oracle):
    words = np.array([0x76543210, -1, 0x0F0F0F0F], dtype=np.int64).astype(np.uint32).view(np.int32)
    out = oracle.bf16_to_f32(oracle.dequant_int4_debug(words)).reshape(3, 8)
    assert out[0].tolist() == [-8, -4, -7, -3, -6, -2, -5, -1]  # n0 n4 n1 n5 n2 n6 n3 n7, minus 8
    assert out[1].tolist() == [7] * 8
    assert out[2].tolist() == [7, 7, -8, -8, 7, 7, -8, -8]


def test_mx4_nan_exponent(oracle):
    """exponent 254 is finite, 255 is NaN (reference tests/tinygemm/test_tinygemm_mx4.py:443-506)."""
    codes = np.full((2, 32), 2, np.int32)  # fp4 code 2 = 1.0
    e = np.array([[254], [255]], np.uint8)
    w = oracle.bf16_to_f32(oracle.dequant(codes, 32, oracle.Q_MX4, e))
    assert np.all(w[0] == 2.0 ** 127) and np.all(np.isnan(w[1]))


# ---------------------------------------------------------------- known-answer GEMMs

@pytest.mark.parametrize("group", [32, 64, 128, 256])
def test_identity_any4_is_bit_exact(oracle, group):
    """w = eye(k), group-quantised, LUT = 8 - arange(16), scales negated => y == x bit for bit
    (reference tests/tinygemm/test_tinygemm_any4.py:14-37, 117-139)."""
    torch.manual_seed(0)
    k = 512
    x = torch.randn(5, k).bfloat16()
    codes, sz = U.group_quantize_tensor(torch.eye(k, dtype=torch.bfloat16), 4, group)
    lut = -(torch.arange(16, dtype=torch.bfloat16) - 8)
    sz[:, :, 0] *= -1.0
    y16, _ = oracle.linear(bits16(x), codes.numpy(), group, oracle.Q_ANY4_GLOBAL, bits16(sz), bits16(lut))
    assert np.array_equal(y16, bits16(x))
    # and through the packed layouts
    for inner in (2, 4, 8):
        assert np.array_equal(oracle.unpack_Bint4(oracle.pack_Bint4(codes.numpy(), inner), k, k), codes.numpy())


def test_identity_mx4_row_exponent(oracle):
    """quantize_mx4(eye) with the exponent of row r raised by r % 4 scales output column r by 2^(r%4)
    (reference tests/tinygemm/test_tinygemm_mx4.py:14-39)."""
    torch.manual_seed(1)
    k = 128
    x = torch.randn(3, k).bfloat16()
    q, e = U.quantize_mx4(torch.eye(k), 32)
    e = e + (torch.arange(k) % 4).to(torch.uint8).unsqueeze(1)
    y16, _ = oracle.linear(bits16(x), q.numpy(), 32, oracle.Q_MX4, e.numpy())
    expect = (x.float() * (2.0 ** (torch.arange(k) % 4).float())).bfloat16()
    assert np.array_equal(y16, bits16(expect))


def test_linear_equals_dequant_then_gemm(oracle):
    rng = np.random.default_rng(3)
    n, k, g, m = 32, 256, 64, 4
    codes = rng.integers(0, 16, (n, k), dtype=np.int32)
    lut = bits16(torch.randn(n, 16).bfloat16())
    sz = bits16((torch.randn(k // g, n, 2) * 0.05).bfloat16())
    x = bits16(torch.randn(m, k).bfloat16())
    w = oracle.dequant(codes, g, oracle.Q_ANY4_ROWWISE, sz, lut)
    y_a, f_a = oracle.gemm(x, w)
    y_b, f_b = oracle.linear(x, codes, g, oracle.Q_ANY4_ROWWISE, sz, lut)
    assert np.array_equal(y_a, y_b) and np.array_equal(f_a, f_b)


# ---------------------------------------------------------------- the reference's own dequant-matmul (H-Q, config 1)

def test_reference_fixture_any4_n1024(oracle):
    """BASELINE config 1: any4 per-row LUT from the reference's k-means quantiser, m=1, n=k=1024, g=128.
    The reference's CPU path (pseudo dequant, op-by-op bf16) and the kernel-faithful dequant (single
    rounding) must agree within 1e-2 max-abs on y (north_star tolerance)."""
    g = load_golden("any4_n1024_k1024_g128_seed1234.npz")
    n, k, gs = int(g["n"]), int(g["k"]), int(g["g"])
    nib = g["codes_nib"]
    codes = np.empty((n, k), np.int32)
    codes[:, 0::2] = nib & 0xF
    codes[:, 1::2] = nib >> 4
    y16, y32 = oracle.linear(g["x_bits"], codes, gs, oracle.Q_ANY4_ROWWISE, g["sz_bits"], g["lut_m8_bits"])
    y_ref = oracle.bf16_to_f32(g["y_bits"])
    err = np.abs(oracle.bf16_to_f32(y16) - y_ref).max()
    assert np.abs(y_ref).max() > 1.0
    assert err <= 1e-2, err
    # kernel-faithful dequant vs the reference's pseudo dequant on the captured rows
    w = oracle.dequant(codes[:8], gs, oracle.Q_ANY4_ROWWISE, g["sz_bits"][:, :8, :], g["lut_m8_bits"][:8])
    wr = g["wdeq_rows0_8_bits"]
    dw = np.abs(oracle.bf16_to_f32(w) - oracle.bf16_to_f32(wr))
    assert dw.max() <= 1e-3  # 4.9e-4 observed (op-by-op bf16 vs single rounding)


def test_reference_fixture_anyq_linspace(oracle):
    """tests/test_anyq.py:63-108 inputs: weights are permutations of linspace(-8,7); the reference expects
    torch.testing.assert_close(y, x @ w.T) with default bf16 tolerances (atol 1e-5, rtol 1.6e-2)."""
    g = load_golden("anyq_linspace64.npz")
    for gs in (32, 64):
        p = f"bf16_g{gs}_"
        y16, _ = oracle.linear(g[p + "x_bits"], g[p + "codes"].astype(np.int32), gs, oracle.Q_ANY4_GLOBAL,
                               g[p + "sz_bits"], g[p + "lut_bits"])
        y = torch.from_numpy(oracle.bf16_to_f32(y16))
        y_ref = torch.from_numpy(oracle.bf16_to_f32(g[p + "y_bits"]))
        torch.testing.assert_close(y, y_ref, atol=1e-5, rtol=1.6e-2)


def test_fp16_conversions(oracle):
    L = oracle.lib()
    import ctypes

    L.tgo_f32_to_f16.restype = ctypes.c_uint16
    L.tgo_f32_to_f16.argtypes = [ctypes.c_float]
    L.tgo_f16_to_f32.restype = ctypes.c_float
    L.tgo_f16_to_f32.argtypes = [ctypes.c_uint16]
    vals = torch.cat([torch.randn(2000) * 10, torch.randn(500) * 1e-6, torch.tensor([0.0, -0.0, 65504.0, 1e6, 6e-8, 3e-8])])
    for v in vals.tolist():
        h = L.tgo_f32_to_f16(v)
        exp = torch.tensor(v, dtype=torch.float32).half()
        assert h == int(exp.view(torch.int16).item()) & 0xFFFF, v
        assert L.tgo_f16_to_f32(h) == exp.float().item() or (exp.float().item() != exp.float().item())


# ---------------------------------------------------------------- the group-scaled restatement (TG_NUM_FAST numerics)

def test_group_scaled_restatement_is_pinned_to_the_reference_contraction(oracle):
    """oracle.linear_group_scaled is a DERIVED formula (the reference's sum without its per-weight rounding to 16 bits): it must
    stay within the analytic bound 2^-9 * sum|x w| (2^-12 for fp16) of oracle.linear on every quantisation type, and reproduce
    the reference's captured CPU output of the any4 fixture as closely as the reference-faithful contraction does."""
    rng = np.random.default_rng(5)
    n, k, m = 48, 512, 3
    for dtype, eps in ((oracle.BF16, 2.0 ** -9), (oracle.F16, 2.0 ** -12)):
        to16 = oracle.bf16_bits if dtype == oracle.BF16 else (lambda a: a.astype(np.float16).view(np.uint16))
        to32 = oracle.bf16_to_f32 if dtype == oracle.BF16 else (lambda b: b.view(np.float16).astype(np.float32))
        x = to16(rng.standard_normal((m, k)).astype(np.float32))
        codes = rng.integers(0, 16, (n, k), dtype=np.int32)
        for g in (32, 128):
            sz = to16((rng.random((k // g, n, 2)) * 0.02 + 0.005).astype(np.float32))
            lut = to16(rng.standard_normal((n, 16)).astype(np.float32))
            ex = rng.integers(120, 131, (n, k // g), dtype=np.uint8)
            cases = [(oracle.Q_INT4, sz, None), (oracle.Q_ANY4_GLOBAL, sz, lut[0]), (oracle.Q_ANY4_ROWWISE, sz, lut)]
            if dtype == oracle.BF16:
                cases.append((oracle.Q_MX4, ex, None))
            for q, qi, lt in cases:
                _, y_ref = oracle.linear(x, codes, g, q, qi, lt, dtype)
                _, y_gs = oracle.linear_group_scaled(x, codes, g, q, qi, lt, dtype)
                w = to32(oracle.dequant(codes, g, q, qi, lt, dtype)).astype(np.float64)
                S = np.abs(to32(x).astype(np.float64)) @ np.abs(w).T
                bound = eps * S + 1e-6 * S + 1e-30
                assert (np.abs(y_ref.astype(np.float64) - y_gs) <= bound).all(), (q, g, dtype)
                if q == oracle.Q_MX4:  # exact weights: the two formulas agree to float32 rounding
                    assert np.allclose(y_ref, y_gs, rtol=1e-6, atol=1e-30)
    f = load_golden("any4_n1024_k1024_g128_seed1234.npz")
    nib = f["codes_nib"]
    codes = np.empty((int(f["n"]), int(f["k"])), np.int32)
    codes[:, 0::2] = nib & 15
    codes[:, 1::2] = nib >> 4
    y16, _ = oracle.linear_group_scaled(f["x_bits"], codes, int(f["g"]), oracle.Q_ANY4_ROWWISE, f["sz_bits"], f["lut_m8_bits"])
    err = np.abs(oracle.bf16_to_f32(y16) - oracle.bf16_to_f32(f["y_bits"])).max()
    assert err <= 1e-2, err  # north_star tolerance against the reference's own CPU dequant-matmul


def test_group_scaled_distance_from_reference_at_the_benchmarked_shape(oracle):
    """The numerics contract of the library's default (group-scaled) arithmetic, written down at the BENCHMARKED shape: 512 weight
    rows of the bench recipe (k = 4096, g = 128, per-row LUT; SURVEY.md 8d generator) through both CPU restatements.  The
    group-scaled result is the reference's sum without its per-weight rounding to bf16 (MatrixLayoutB.cuh:1042-1046):
      * before the output rounding: max-abs distance from the reference's f32 sums <= 1e-2 at the fixture's scale (max|y| = 2.2,
        north_star); after it: within max(that, one bf16 step of the largest output) -- a flipped final rounding is one step;
      * never more than one bf16 step away for outputs of the top binade, and equal in most outputs."""
    gen = torch.Generator().manual_seed(0)
    n, k, g, m = 512, 4096, 128, 1
    codes = torch.randint(0, 16, (n, k), dtype=torch.int32, generator=gen).numpy()
    lut = oracle.bf16_bits(torch.randn(n, 16, generator=gen).numpy())
    scales = torch.rand(k // g, n, generator=gen) * 0.02 + 0.005
    zeros = torch.randn(k // g, n, generator=gen) * 0.01
    sz = oracle.bf16_bits(torch.stack([scales, zeros], dim=2).numpy())
    x = oracle.bf16_bits(torch.randn(m, k, generator=gen).numpy())
    r16, r32 = oracle.linear(x, codes, g, oracle.Q_ANY4_ROWWISE, sz, lut)
    g16, g32 = oracle.linear_group_scaled(x, codes, g, oracle.Q_ANY4_ROWWISE, sz, lut)
    ref, gs = oracle.bf16_to_f32(r16).astype(np.float64), oracle.bf16_to_f32(g16).astype(np.float64)
    ymax = np.abs(ref).max()
    err = np.abs(gs - ref).max()
    step = 2.0 ** (np.floor(np.log2(ymax)) - 7)    # one bf16 step of the largest output
    assert err <= max(1e-2 * max(1.0, ymax / 2.2), step), (err, ymax)
    assert np.abs(g32.astype(np.float64) - r32.astype(np.float64)).max() <= 1e-2 * max(1.0, ymax / 2.2)
    differ = (g16 != r16).mean()
    assert differ < 0.6, differ                      # (measured: ~0.4 of the outputs land on the neighbouring bf16 value)
    top = np.abs(ref) >= 2.0 ** np.floor(np.log2(ymax))
    key = lambda u: np.where(u & 0x8000, -(u.astype(np.int32) & 0x7fff), u.astype(np.int32) & 0x7fff)  # noqa: E731
    assert np.abs(key(g16) - key(r16))[top].max() <= 1
