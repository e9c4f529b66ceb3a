"""CPU suite, part 2: host logic, the C-ABI surface, the op/functional/module mirror and the row-shard
all-gather (gloo, world_size 2).  No GPU, no HIP compute."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- C ABI

def _declared_symbols():
    syms = set()
    for name in sorted(os.listdir(os.path.join(ROOT, "include"))):  # every header of the C ABI
        if name.endswith(".h"):
            hdr = open(os.path.join(ROOT, "include", name)).read()
            syms |= set(re.findall(r"TG_API\s+[\w\s\*]+?\b((?:tg|dg)_\w+)\s*\(", hdr))
    return sorted(syms)


def test_c_abi_exports_every_declared_symbol():
    from any4_amd import _lib

    syms = _declared_symbols()
    assert len(syms) >= 11 and "tg_gemm_w4" in syms
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/*.h but not exported"
    assert set(syms) == set(_lib.SYMBOLS), "ctypes binding and header disagree"
    assert _lib.load().tg_abi_version() == _lib.TG_ABI_VERSION == 8


def test_peer_gather_preconditions_fail_before_any_launch():
    """include/peer_gather_hip.h: argument validation of the one-shot gather returns TG_E_* without touching HIP."""
    from any4_amd import _lib

    L = _lib.load()
    a = _lib.PeerGather()
    assert L.tg_peer_gather_launch(ctypes.byref(a), 0, None) == -1               # TG_E_NULL: no src / seq / status
    a.src, a.seq, a.status = 4096, 8192, 8192 + 128
    a.world, a.rank, a.m, a.cols_local = 2, 2, 1, 64
    assert L.tg_peer_gather_launch(ctypes.byref(a), 0, None) == -7               # rank outside the world
    a.rank, a.world = 0, _lib.TG_PEER_MAX_WORLD + 1
    assert L.tg_peer_gather_launch(ctypes.byref(a), 0, None) == -7               # world too large
    a.world, a.cols_local = 2, 60
    assert L.tg_peer_gather_launch(ctypes.byref(a), 0, None) == -8               # row bytes not a multiple of 16
    a.cols_local = 64
    assert L.tg_peer_gather_launch(ctypes.byref(a), 0, None) == -1               # dst / flags of a rank missing
    assert L.tg_peer_alloc(0, 0, ctypes.byref(ctypes.c_void_p())) == -7
    assert L.tg_peer_export(0, None, ctypes.byref(_lib.PeerHandle())) == -1
    assert L.tg_peer_open(0, None, ctypes.byref(ctypes.c_void_p())) == -1
    assert L.tg_peer_close(0, None) == -1 and L.tg_peer_free(0, None) == -1


def test_c_abi_preconditions_fail_before_any_launch():
    """Negative codes are returned by argument validation, which runs before the first HIP call."""
    from any4_amd import _lib

    L = _lib.load()
    assert L.tg_convert_to_Bint4(None, 8, 64, 4, None, 0, None) == -1          # TG_E_NULL
    buf = (ctypes.c_int32 * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.tg_convert_to_Bint4(p, 8, 64, 3, p, 0, None) == -2                # bad innerKTiles
    assert L.tg_convert_to_Bint4(p, 8, 96, 4, p, 0, None) == -3                # k % 64
    assert L.tg_convert_to_Aint4(p, 8, 64, 8, p, 0, None) == -2
    a = _lib.W4Gemm(x=p, w=p, qinfo=p, lut=p, y=p, m=1, wrows=16, k=128, group=48, qtype=2, dtype=0, w_on_right=1,
                    inner_k_tiles=4, batch=1)
    assert L.tg_gemm_w4(ctypes.byref(a), 0, None) == -4                         # bad group
    a.group, a.qtype, a.dtype = 32, 3, 1
    assert L.tg_gemm_w4(ctypes.byref(a), 0, None) == -5                         # mx4 + fp16
    a.dtype, a.qtype, a.k = 0, 2, 144
    assert L.tg_gemm_w4(ctypes.byref(a), 0, None) == -3                         # k % 32
    a.k, a.lut = 128, None
    assert L.tg_gemm_w4(ctypes.byref(a), 0, None) == -1                         # any4 without LUT
    a.lut, a.numerics = p, 7
    assert L.tg_gemm_w4(ctypes.byref(a), 0, None) == -7                         # unknown numerics
    a.numerics, a.m, a.k = 0, 1 << 20, 4096
    assert L.tg_gemm_w4(ctypes.byref(a), 0, None) == -10                        # TG_E_SIZE: activations of 8 GiB
    a.m, a.k, a.wrows = 1, 1 << 17, 1 << 16
    assert L.tg_gemm_w4(ctypes.byref(a), 0, None) == -10                        # TG_E_SIZE: packed weights of 4 GiB
    for code in range(-11, 1):
        assert len(L.tg_error_string(code)) > 0
    with pytest.raises(RuntimeError, match="qGroupSize"):
        _lib.check(-4, "x")


# ---------------------------------------------------------------- op surface

REFERENCE_SCHEMAS = {  # tinygemm_lib/TinyGemm.cpp:17-122
    "convert_matrix_to_m16n8k16_A_layout": "(Tensor t, int innerKTiles) -> Tensor",
    "convert_matrix_to_m16n8k16_Aint4_layout": "(Tensor t, int innerKTiles) -> Tensor",
    "convert_matrix_to_m16n8k16_Aint8_layout": "(Tensor t, int innerKTiles) -> Tensor",
    "convert_matrix_from_m16n8k16_A_layout": "(Tensor t, int m, int k) -> Tensor",
    "convert_matrix_to_m16n8k16_B_layout": "(Tensor t, int innerKTiles) -> Tensor",
    "convert_matrix_to_m16n8k16_Bint4_layout": "(Tensor t, int innerKTiles) -> Tensor",
    "convert_matrix_to_m16n8k16_Bint8_layout": "(Tensor t, int innerKTiles) -> Tensor",
    "convert_matrix_from_m16n8k16_B_layout": "(Tensor t, int n, int k) -> Tensor",
    "tinygemm_y_f16TC_x_f16TC_w_int4TC": "(Tensor A, Tensor B, int qGroupSize, Tensor qScaleAndZeros, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16RM_x_f16RM_w_int4TC": "(Tensor A, Tensor B, int qGroupSize, Tensor qScaleAndZeros, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16TC_x_f16TC_w_any4TC": "(Tensor A, Tensor B, int qGroupSize, Tensor qScaleAndZeros, Tensor int4DequantValues, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16RM_x_f16RM_w_any4TC": "(Tensor A, Tensor B, int qGroupSize, Tensor qScaleAndZeros, Tensor int4DequantValues, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16TC_x_f16TC_w_mx4TC": "(Tensor A, Tensor B, int qGroupSize, Tensor mx4Exponents, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16RM_x_f16RM_w_mx4TC": "(Tensor A, Tensor B, int qGroupSize, Tensor mx4Exponents, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16TC_x_f16TC_w_int8TC": "(Tensor A, Tensor B, int qGroupSize, Tensor qScaleAndZeros, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16RM_x_f16RM_w_int8TC": "(Tensor A, Tensor B, int qGroupSize, Tensor qScaleAndZeros, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16TC_x_f16TC_w_f16TC": "(Tensor A, Tensor B, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16RM_x_f16RM_w_f16TC": "(Tensor A, Tensor B, bool weightOnRight) -> Tensor",
    "tinygemm_dequant_int4": "(Tensor t) -> Tensor",
}


def test_ops_registered_with_reference_schemas():
    import tinygemm  # noqa: F401

    for name, sig in REFERENCE_SCHEMAS.items():
        op = getattr(torch.ops.tinygemm, name)
        schema = str(op.default._schema)
        assert schema == f"tinygemm::{name}{sig}", schema


def test_no_cpu_fallback():
    import tinygemm  # noqa: F401

    with pytest.raises(NotImplementedError):
        torch.ops.tinygemm.convert_matrix_to_m16n8k16_Bint4_layout(torch.zeros(8, 64, dtype=torch.int32), 4)
    with pytest.raises(NotImplementedError):
        torch.ops.tinygemm.tinygemm_y_f16RM_x_f16RM_w_int4TC(
            torch.zeros(1, 64).bfloat16(), torch.zeros(1, 1, 32, 2, dtype=torch.int32), 32, torch.zeros(2, 8, 2).bfloat16(), True)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from any4_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU"):
        _lib.load()


def test_valid_tinygemm_kernel_call_truth_table():
    import tinygemm_lib.functional as F

    right = ["linear_y_f16RM_x_f16RM_W_any4TC", "linear_y_f16TC_x_f16TC_W_any4TC"]
    left = ["linear_y_f16TC_W_any4TC_x_f16TC", "linear_y_f16RM_W_any4TC_x_f16RM"]
    for w in (1, 2, 3, 4, 8, 16):
        for api in right:
            assert bool(F.valid_tinygemm_kernel_call(api, w)) == (w in (2, 4, 8))
        for api in left:
            assert bool(F.valid_tinygemm_kernel_call(api, w)) == (w in (1, 2, 4))
    assert not F.valid_tinygemm_kernel_call("linear_y_f16RM_x_f16RM_W_int4TC", 4)
    names = [n for n in dir(F) if n.startswith("linear_y_")]
    assert len(names) == 16  # 4 int4 + 4 int8 + 4 any4 + 4 f16 (reference functional.py:20-259)


def test_module_surface():
    import modules

    m = modules.Any4Linear(256, 64, bias=True, dtype=torch.bfloat16, group_size=128)
    assert m.weight.shape == (64, 256) and m.weight.dtype == torch.int32 and not m.weight.requires_grad
    assert m.scales_and_zeros.shape == (2, 64, 2) and m.lut.shape == (64, 16) and m.bias.shape == (64,)
    assert m.kernel == "linear_y_f16RM_x_f16RM_W_any4TC" and m.w_inner_k == 4 and not m.weight_reshaped
    assert m.per_row and m.n_bit == 4 and m.N_BIT == 4 and m.group_size == 128
    assert set(m.state_dict()) == {"weight", "scales_and_zeros", "lut", "bias"}
    assert "per_row=True" in repr(m)
    g = modules.Any4Linear(256, 64, bias=False, per_row=False)
    assert g.lut.shape == (16,) and g.bias is None
    i4 = modules.Int4Linear(256, 64, dtype=torch.float16)
    assert i4.kernel == "linear_y_f16RM_W_int4TC_x_f16RM" and i4.weight.abs().sum() == 0
    assert set(i4.state_dict()) == {"weight", "scales_and_zeros", "bias"}
    i8 = modules.Int8Linear(256, 64)
    assert i8.w_inner_k == 2
    bad = modules.Any4Linear(256, 64, kernel="linear_y_bogus")
    with pytest.raises(ValueError, match="Unsupported kernel"):
        bad.reshape_weight()
    with pytest.raises(ValueError, match="Unsupported kernel"):
        bad(torch.zeros(1, 256))


def test_group_quantize_roundtrip():
    from tinygemm_lib.utils import extract_scales_and_zeros, group_quantize_tensor

    torch.manual_seed(0)
    w = torch.randn(32, 256)
    for g in (32, 64, 128):
        codes, sz = group_quantize_tensor(w, 4, g)
        assert codes.dtype == torch.int32 and codes.min() >= 0 and codes.max() <= 15
        assert sz.shape == (256 // g, 32, 2)
        s, z = extract_scales_and_zeros(sz, w.shape, g)
        deq = (codes.float() - 8) * s + z
        assert (deq - w).abs().max() <= s.max() * 0.5 + 1e-6


# ---------------------------------------------------------------- row sharding over gloo (world_size 2)

def _shard_worker(rank, world, port, results):
    import torch.distributed as dist

    from any4_amd.shard import RowShardedLinear, row_range, shard_any4_params

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gen = torch.Generator().manual_seed(0)  # same full problem on every rank
        n, k, g, m = 64, 128, 32, 3
        codes = torch.randint(0, 16, (n, k), dtype=torch.int32, generator=gen)
        lut = torch.randn(n, 16, generator=gen)
        sz = torch.randn(k // g, n, 2, generator=gen) * 0.1
        x = torch.randn(2, m, k, generator=gen)

        def dequant(c, l, s):
            sc = s[:, :, 0].t().repeat_interleave(g, dim=1)
            zc = s[:, :, 1].t().repeat_interleave(g, dim=1)
            return torch.gather(l, 1, c.long()) * sc + zc

        y_full = x @ dequant(codes, lut, sz).t()
        c, l, s = shard_any4_params(codes, lut, sz, rank, world)
        lo, hi = row_range(n, rank, world)
        assert c.shape == (n // world, k) and s.shape == (k // g, n // world, 2) and l.shape == (n // world, 16)

        class Local(torch.nn.Module):  # stand-in for the rank-local Any4Linear (the HIP GEMM needs a GPU)
            def forward(self, inp):
                return inp @ dequant(c, l, s).t()

        y = RowShardedLinear(Local(), n)(x)
        ok = y.shape == y_full.shape and torch.allclose(y, y_full, atol=1e-5)
        y_sharded = RowShardedLinear(Local(), n, gather_output=False)(x)
        ok = ok and torch.allclose(y_sharded, y_full[..., lo:hi], atol=1e-5)
        results[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_row_sharded_all_gather_gloo():
    import torch.multiprocessing as mp

    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_shard_worker, args=(world, port, results), nprocs=world, join=True)
    assert dict(results) == {0: True, 1: True}


def test_row_range_validation():
    from any4_amd.shard import row_range

    assert row_range(4096, 3, 8) == (1536, 2048)
    assert row_range(14336, 7, 8) == (12544, 14336)
    with pytest.raises(ValueError):
        row_range(100, 0, 8)


# ---------------------------------------------------------------- bench.py pieces that run without a GPU

def test_bench_cpu_baseline_leg_and_byte_formula():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # config 2 = 9 060 352 B, m = 8 -> 9 175 040 B
    # SURVEY.md 8d figures
    assert bench.alg_bytes(1, 4096, 4096, 128) == 9060352
    assert bench.alg_bytes(8, 4096, 4096, 128) == 9175040
    assert bench.alg_bytes(8, 8192, 8192, 128) == 36175872
    assert bench.alg_bytes(1, 4096, 4096, 128, "int4") == 8929280
    assert bench.alg_bytes(1, 4096, 4096, 128, "any4_global") == 8929312
    assert bench.alg_bytes(1, 4096, 4096, 32, "mx4") == 8929280
    cb = bench.cpu_baseline_torch(1, 512, 512, 128, budget_s=0.2)
    assert cb["unit"] == "GB/s" and cb["value"] > 0 and cb["kind"] == "port" and cb["cores"] >= 1
    assert "layers of the bench workload" in cb["sample"] and "physical cores" in cb["sample"]
    co = bench.cpu_baseline_oracle(1, 512, 512, 128, budget_s=0.2)
    assert co["value"] > 0 and co["cores"] >= 1


def test_xr_kernel_routing():
    """Which stacked launches the xr kernel takes (tg_gemm_w4_plan, no GPU work): Bint4, k = 4096 / 8192 / 14336, innerKTiles 4, rows a
    multiple of 64, 2 <= m <= 16, every group size (k = 4096), at least two work items per CU; everything else stays where it was."""
    from any4_amd import ops

    plan = lambda m, n, k, g, q, inner=4, batch=64, right=True: ops.gemm_w4_plan(m, n, k, g, {'int4': 0, 'any4_global': 1, 'any4_rowwise': 2, 'mx4': 3}[q], right, inner, batch=batch, detail=True)
    for m in (2, 8, 9, 16):
        for q in ("int4", "any4_global", "any4_rowwise"):
            for g in (32, 64, 128, 256):
                assert plan(m, 4096, 4096, g, q) == "pair_xr", (m, q, g)
    assert plan(1, 4096, 4096, 128, "any4_rowwise") == "pair"             # m = 1: the 32x32x16 kernel
    assert plan(8, 4096, 4096, 32, "mx4") == "pair_xr" and plan(16, 4096, 4096, 32, "mx4") == "pair_xr"  # mx4: bf16, g = 32, k = 4096
    assert plan(8, 4096, 8192, 32, "mx4") != "pair_xr"
    # k = 8192 / 14336 (round 5): packed activation rows up to 8 rows (k = 14336: at 8 rows), k-windows with f32 partial sums in the
    # caller's workspace from 9 rows on (without the workspace those calls stay on the older kernels)
    assert plan(8, 4096, 8192, 128, "any4_rowwise") == "pair_xr" and plan(2, 4096, 8192, 64, "int4") == "pair_xr"
    assert plan(9, 4096, 8192, 128, "any4_rowwise", batch=16) == "pair_xr" and plan(16, 8192, 8192, 256, "int4", batch=8) == "pair_xr"
    assert plan(16, 4096, 14336, 128, "any4_rowwise") == "pair_xr" and plan(8, 4096, 14336, 128, "any4_rowwise") == "pair_xr"
    assert plan(6, 4096, 14336, 128, "any4_rowwise") == "pair"            # (measured slower there: the workspace variant keeps it)
    assert ops.gemm_w4_plan(16, 4096, 14336, 128, 2, True, 4, batch=64, detail=True, workspace=False) != "pair_xr"
    assert plan(16, 4096, 2048, 128, "any4_rowwise") != "pair_xr" and plan(16, 4096, 14336, 256, "any4_rowwise") == "pair_xr"
    assert plan(8, 4096, 4096, 128, "any4_rowwise", inner=8) != "pair_xr"
    assert plan(8, 4104, 4096, 128, "any4_rowwise") != "pair_xr"          # rows not a multiple of 64
    assert plan(8, 4096, 4096, 128, "any4_rowwise", batch=4) != "pair_xr"  # 256 items: fewer than two per CU
    # weights on the left: the reference's Aint4 words stay on their own kernels; the native row-per-lane order IS a B-side call
    q2 = {'int4': 0, 'any4_global': 1, 'any4_rowwise': 2, 'mx4': 3}
    assert ops.gemm_w4_plan(8, 4096, 4096, 128, q2["any4_rowwise"], False, 4, batch=64, detail=True, weight_format="reference") != "pair_xr"
    assert ops.gemm_w4_plan(8, 4096, 4096, 128, q2["any4_rowwise"], False, 4, batch=64, detail=True, weight_format="native") == "pair_xr"
    assert ops.gemm_w4_plan(1, 4096, 4096, 128, q2["int4"], False, 4, weight_format="native") == "gemv"   # Int4Linear's default kernel at batch 1
    assert ops.gemm_w4_plan(1, 4096, 4096, 128, q2["int4"], False, 2, weight_format="native") == "gemv"   # (the Aint4 innerKTiles is a shape only)
    # one layer per launch with 3 ... 8 rows, k <= 4096, groups of 128 / 256: the gemv kernel's matrix-core contraction; otherwise pair16
    assert ops.gemm_w4_plan(8, 4096, 4096, 128, q2["any4_rowwise"], True, 4) == "gemv"
    assert ops.gemm_w4_plan(5, 28672, 4096, 256, q2["int4"], True, 4) == "gemv"
    assert ops.gemm_w4_plan(8, 4096, 4096, 64, q2["any4_rowwise"], True, 4) == "pair"
    assert ops.gemm_w4_plan(8, 4096, 14336, 128, q2["any4_rowwise"], True, 4) == "pair"
    assert ops.gemm_w4_plan(9, 4096, 4096, 128, q2["any4_rowwise"], True, 4) == "pair"
    # ... except one layer per launch at k = 4096 with more 16-row tiles than CUs: w4_gemm_pair16_loop_kernel (plan "pair") up to eight tiles
    # per CU (32768 rows), beyond that -- and for fragment-order operands from 80 64-row items on -- the xr kernel with one workgroup per item
    assert ops.gemm_w4_plan(16, 16384, 4096, 128, q2["any4_rowwise"], True, 4, detail=True) == "pair"
    assert ops.gemm_w4_plan(9, 28672, 4096, 64, q2["int4"], True, 4, detail=True) == "pair"
    assert ops.gemm_w4_plan(16, 5120, 4096, 128, q2["any4_rowwise"], True, 4, detail=True) == "pair"
    assert ops.gemm_w4_plan(16, 32768 + 64, 4096, 128, q2["any4_rowwise"], True, 4, detail=True) == "pair_xr"
    assert ops.gemm_w4_plan(8, 16384, 4096, 128, q2["any4_rowwise"], True, 4, detail=True) == "gemv"
    assert ops.gemm_w4_plan(8, 16384, 4096, 64, q2["any4_rowwise"], True, 4, detail=True) == "pair"
    assert ops.gemm_w4_plan(8, 65536, 4096, 64, q2["any4_rowwise"], True, 4, detail=True) == "pair_xr"
    assert ops.gemm_w4_plan(16, 8192, 8192, 128, q2["any4_rowwise"], True, 4, detail=True) != "pair_xr"    # (k = 4096 only)
    assert ops.gemm_w4_plan(4, 4096, 4096, 32, q2["any4_rowwise"], True, 4) == "gemv"   # (the v_dot2 contraction: any group size)
    # more than 16 rows (default numerics, row-major operands): ceil(m / 16) launches of up to 16 rows on the same kernels -- the plan is
    # the 16-row block's; shapes whose blocks have no group-scaled kernel (innerKTiles 8 at one layer per launch) stay where they were
    assert plan(17, 4096, 4096, 128, "any4_rowwise") == "pair_xr" and plan(64, 4096, 4096, 128, "any4_rowwise") == "pair_xr"
    assert ops.gemm_w4_plan(33, 4096, 4096, 128, q2["any4_rowwise"], True, 4, workspace=False) == "pair"
    # ONE layer at 33 ... 64 rows with the caller's workspace: the tile GEMM as a split-K launch (f32 partial tiles in the workspace)
    assert ops.gemm_w4_plan(33, 4096, 4096, 128, q2["any4_rowwise"], True, 4) == "tile" and ops.gemm_w4_plan(17, 4096, 4096, 128, q2["any4_rowwise"], True, 4) == "tile"
    assert ops.gemm_w4_plan(64, 4096, 4096, 128, q2["any4_rowwise"], True, 4, batch=2) == "pair"
    assert ops.gemm_w4_plan(16, 4096, 4096, 128, q2["any4_rowwise"], True, 4) == "pair"      # (one 16-row pass stays)
    assert ops.gemm_w4_plan(17, 4096, 512, 128, q2["any4_rowwise"], True, 4) == "pair"       # (k too short to split)
    # beyond the 64 rows of the row blocks: the LDS-tiled MFMA GEMM (w4_gemm_tile.cuh), in BOTH numerics settings (it computes the reference's
    # weights), on both operand sides of the native words; not for mx4 / innerKTiles != 4 / fragment-order operands
    for num in ("fast", "reference"):
        assert ops.gemm_w4_plan(65, 4096, 4096, 128, q2["any4_rowwise"], True, 4, numerics=num, detail=True) == "tile"
        assert ops.gemm_w4_plan(512, 14336, 4096, 32, q2["int4"], True, 4, numerics=num) == "tile"
        assert ops.gemm_w4_plan(2048, 4096, 4096, 128, q2["any4_global"], False, 4, numerics=num, weight_format="native") == "tile"
    assert ops.gemm_w4_plan(64, 4096, 4096, 128, q2["any4_rowwise"], True, 4, workspace=False) == "pair"      # (64 rows, no workspace: four 16-row blocks)
    assert ops.gemm_w4_plan(512, 4096, 4096, 32, q2["mx4"], True, 4) == "tile"      # (mx4: the same tables, entries fp4 * 2^e)
    assert ops.gemm_w4_plan(512, 4096, 4096, 32, q2["mx4"], True, 2) in ("stream", "splitk", "pair")
    assert ops.gemm_w4_plan(512, 4096, 4096, 128, q2["any4_rowwise"], True, 8, detail=True) in ("stream", "splitk")
    assert ops.gemm_w4_plan(512, 4096, 4096, 128, q2["any4_rowwise"], False, 4, weight_format="reference", detail=True) in ("stream", "splitk")
    assert ops.gemm_w4_plan(512, 4096, 4096, 128, q2["any4_rowwise"], True, 4, workspace=False) == "tile"          # (no workspace)
    # the split-K tile launch asks for splits x m x rows x 4 bytes of f32 partial tiles (as many splits as keep tiles x splits <= CUs)
    def ws_need(m, n, k, g=128, q=2):
        from any4_amd import _lib

        buf = ctypes.create_string_buffer(256)
        pz = (ctypes.addressof(buf) + 63) & ~63
        a = _lib.W4Gemm(x=pz, w=pz, qinfo=pz, lut=pz, y=pz, m=m, wrows=n, k=k, group=g, qtype=q, dtype=_lib.TG_BF16, w_on_right=1, inner_k_tiles=4, batch=1,
                        stride_x=16, stride_w=16, stride_qinfo=16, stride_lut=16, stride_y=16)
        return _lib.load().tg_gemm_w4_workspace_bytes(ctypes.byref(a))
    assert ws_need(128, 4096, 4096) == 4 * 128 * 4096 * 4 and ws_need(64, 4096, 4096) == 4 * 64 * 4096 * 4      # 64 tiles x 4 splits
    assert ws_need(256, 4096, 4096) == 2 * 256 * 4096 * 4 and ws_need(512, 4096, 4096) == 0                     # 128 tiles x 2; 256 tiles: unsplit
    assert ws_need(512, 1024, 4096) == 4 * 512 * 1024 * 4 and ws_need(128, 4096, 14336) == 4 * 128 * 4096 * 4
    assert ops.large_m_rows(4096 * 4096) > 1 << 40      # the library-GEMM route is opt-in (ANY4_LARGE_M_GEMM=library / ANY4_LARGE_M)
    assert ops.gemm_w4_plan(33, 4096, 4096, 128, q2["any4_rowwise"], True, 8, detail=True) in ("stream", "splitk")
    assert ops.gemm_w4_plan(33, 4096, 4096, 128, q2["any4_rowwise"], True, 4, numerics="reference", detail=True, workspace=False) in ("stream", "splitk")
    assert ops.gemm_w4_plan(33, 4096, 4096, 128, q2["any4_rowwise"], True, 4, numerics="reference", detail=True) == "tile"


def test_integration_md_binding_and_struct_bytes():
    """The reference-side binding shown in INTEGRATION.md is executed as written (its ctypes struct, against the built library): the
    struct matches the header's length, a call described by it plans a kernel; a binding written against an OLDER header (struct
    truncated after y_layout = ABI 3) is accepted with the later fields off; a length below the ABI-1 prefix, above the library's
    struct, or no length at all is refused with TG_E_STRUCT -- the library never reads past what the caller says it allocated."""
    import ctypes
    import re

    from any4_amd import _lib

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"^class tg_w4_gemm\(ctypes.Structure\):.*?\n\n", text, re.S | re.M)
    assert m, "INTEGRATION.md lost its tg_w4_gemm binding"
    ns = {"ctypes": ctypes}
    exec(m.group(0), ns)  # the snippet, not a copy of it
    S = ns["tg_w4_gemm"]
    assert ctypes.sizeof(S) == ctypes.sizeof(_lib.W4Gemm)
    assert [f[0] for f in S._fields_] == [f[0] for f in _lib.W4Gemm._fields_]
    L = ctypes.CDLL(_lib.LIB_PATH)
    L.tg_gemm_w4_plan.argtypes = [ctypes.c_void_p, ctypes.c_int]
    p = 0x1000  # plan only checks NULL / alignment of the data pointers

    def fill(a):
        a.x = a.w = a.qinfo = a.lut = a.y = p
        a.m, a.wrows, a.k, a.group, a.qtype, a.dtype, a.w_on_right, a.inner_k_tiles, a.batch = 1, 4096, 4096, 128, 2, 0, 1, 4, 1
        return a

    full = fill(S(struct_bytes=ctypes.sizeof(S)))
    assert L.tg_gemm_w4_plan(ctypes.byref(full), -1) == _lib.TG_PLAN_GEMV
    # an ABI-3 binding: the same fields up to y_layout, nothing behind them in memory but poison
    upto = [f[0] for f in S._fields_].index("y_layout") + 1
    Old = type("tg_w4_gemm_v3", (ctypes.Structure,), {"_fields_": S._fields_[:upto]})
    n_old = ctypes.sizeof(Old)
    assert n_old < ctypes.sizeof(S)
    buf = (ctypes.c_ubyte * (n_old + 64))(*([0xff] * (n_old + 64)))
    old = fill(Old.from_buffer(buf))
    old.struct_reserved = old.numerics = old.reserved = old.bias = old.stride_bias = old.workspace = old.workspace_bytes = 0
    old.x_layout = old.y_layout = 0
    old.stride_x = old.stride_w = old.stride_qinfo = old.stride_lut = old.stride_y = 0
    old.struct_bytes = n_old
    assert L.tg_gemm_w4_plan(ctypes.byref(buf), -1) == _lib.TG_PLAN_GEMV  # 0xff behind the struct: never read (would be TG_E_SHAPE / TG_E_FUSION)
    for bad in (0, 8, ctypes.sizeof(S) + 8):
        full.struct_bytes = bad
        assert L.tg_gemm_w4_plan(ctypes.byref(full), -1) == _lib.TG_E_STRUCT, bad
    full.struct_bytes = ctypes.sizeof(S)
    full.struct_reserved = 1
    assert L.tg_gemm_w4_plan(ctypes.byref(full), -1) == _lib.TG_E_STRUCT


def test_module_copies_drop_the_launch_plan():
    """copy.deepcopy / pickle of a quantized module must not carry (or choke on) the recorded launch plan's ctypes struct."""
    import copy
    import ctypes
    import pickle

    import modules
    from any4_amd import _lib

    lin = modules.Int4Linear(64, 32, bias=False, dtype=torch.bfloat16, group_size=32)
    lin.__dict__["_plan"] = (("key",), _lib.W4Gemm(x=ctypes.c_void_p(1234).value))
    twin = copy.deepcopy(lin)
    assert "_plan" not in twin.__dict__ and torch.equal(twin.weight, lin.weight)
    again = pickle.loads(pickle.dumps(lin))
    assert "_plan" not in again.__dict__ and "_plan" in lin.__dict__



def test_weights_on_the_left_format_is_read_off_the_tensor():
    """ops.aside_format: the reference's Aint4 tensor [m/16][k/(16 I)][32][I] and the native one (the Bint4 tensor of the rows
    padded to 16) never have the same shape for one k, so no setting is needed to consume either."""
    from any4_amd import ops

    for m in (16, 40, 4096):
        for k in (32, 64, 96, 256, 4096, 14336):
            j = 4 if k % 64 == 0 else 2
            nat = torch.empty((2 * -(-m // 16), k // (16 * j), 32, j // 2), dtype=torch.int32)
            assert ops.aside_format(nat, k) == "native"
            for inner in (1, 2, 4):
                if k % (16 * inner) == 0:
                    ref = torch.empty((-(-m // 16), k // (16 * inner), 32, inner), dtype=torch.int32)
                    assert ops.aside_format(ref, k) == "reference"
                    assert ref.shape != nat.shape
    with pytest.raises(RuntimeError, match="do not match"):
        ops.aside_format(torch.empty((2, 5, 32, 2), dtype=torch.int32), 4096)
    with pytest.raises(RuntimeError, match="do not match"):   # odd number of 8-row tiles: not a native tensor
        ops.aside_format(torch.empty((3, 64, 32, 2), dtype=torch.int32), 4096)


def test_state_dict_carries_what_decides_how_weight_is_read():
    """eval.py:180-210 (save / load of state_dicts).  The reference keeps kernel / w_inner_k / weight_reshaped in plain attributes
    (modules.py:38-41); here a state_dict holds tensors only (the reference's own keys) and the packed tensor's SHAPE says how it
    has to be read: a fresh module or an already packed one (other format, other innerKTiles) takes the checkpoint's shape, and a
    tensor packed for the other operand side is refused."""
    import modules

    import any4_amd

    k, n, g = 256, 64, 64
    mk = lambda **kw: modules.Int4Linear(k, n, bias=False, dtype=torch.bfloat16, group_size=g, **kw)
    src = mk()          # default kernel: weights on the left
    assert src.weight_format is None
    any4_amd.set_auto_relayout(False)       # (this test reads the words back; the repack itself needs the GPU: test_gpu_aside.py)
    try:
        for fmt, shape, inner in (("native", (2 * n // 16, k // 64, 32, 2), 4), ("reference", (n // 16, k // 32, 32, 2), 2)):
            src.weight.data = torch.randint(0, 2 ** 31 - 1, shape, dtype=torch.int32)
            src.weight_reshaped, src.w_inner_k = True, inner
            assert src.weight_format == fmt
            sd = src.state_dict()
            assert set(sd) == {"weight", "scales_and_zeros"} and all(isinstance(v, torch.Tensor) for v in sd.values())
            {key: v.cpu() for key, v in sd.items()}              # tensor-only consumers (safetensors, save_pretrained) work
            dst = mk()
            dst.__dict__["_plan"] = ("stale",)
            dst.load_state_dict(sd)
            assert dst.weight_reshaped and dst.weight_format == fmt and torch.equal(dst.weight, src.weight)
            assert fmt == "native" or dst.w_inner_k == inner
            assert "_plan" not in dst.__dict__
            # a round-5 checkpoint (a dict under _extra_state) still loads, strict
            old = dict(sd, _extra_state={"kernel": src.kernel, "w_inner_k": inner, "weight_reshaped": True})
            dst2 = mk()
            dst2.load_state_dict(old)
            assert dst2.weight_reshaped and dst2.weight_format == fmt and dst2.w_inner_k == inner
            wrong = mk(kernel="linear_y_f16RM_x_f16RM_W_int4TC")
            if fmt == "reference":       # the other operand side's words: refused, not mis-multiplied
                with pytest.raises(RuntimeError, match="packed for kernel"):
                    wrong.load_state_dict(sd)
                with pytest.raises(RuntimeError, match="packed for kernel"):
                    wrong.load_state_dict(old)
            else:                        # a native weights-on-the-left tensor IS the Bint4 tensor of the same rows: either side multiplies it
                wrong.load_state_dict(sd)
                assert wrong.weight_reshaped and wrong.w_inner_k == 4
            # ... INTO AN ALREADY PACKED module of the other format (quantize_model always ends in reshape_weight())
            for fmt2, shape2, inner2 in (("native", (2 * n // 16, k // 64, 32, 2), 4), ("reference", (n // 16, k // 64, 32, 4), 4)):
                packed = mk()
                packed.weight.data = torch.zeros(shape2, dtype=torch.int32)
                packed.weight_reshaped, packed.w_inner_k = True, inner2
                assert packed.weight_format == fmt2
                packed.load_state_dict(sd)
                assert packed.weight_format == fmt and torch.equal(packed.weight, src.weight)
                assert fmt == "native" or packed.w_inner_k == inner
    finally:
        any4_amd.set_auto_relayout(True)
    # a Bint4 checkpoint with innerKTiles 2 into a module packed with innerKTiles 4 (and into a fresh one)
    b2 = mk(kernel="linear_y_f16RM_x_f16RM_W_int4TC")
    b2.weight.data = torch.randint(0, 2 ** 31 - 1, (n // 8, k // 32, 32, 1), dtype=torch.int32)
    b2.weight_reshaped, b2.w_inner_k = True, 2
    b4 = mk(kernel="linear_y_f16RM_x_f16RM_W_int4TC")
    b4.weight.data = torch.zeros((n // 8, k // 64, 32, 2), dtype=torch.int32)
    b4.weight_reshaped, b4.w_inner_k = True, 4
    b4.load_state_dict(b2.state_dict())
    assert b4.w_inner_k == 2 and torch.equal(b4.weight, b2.weight)
    with pytest.raises(RuntimeError, match="packed for kernel"):       # another layer size
        mk(kernel="linear_y_f16RM_x_f16RM_W_int4TC").load_state_dict({"weight": torch.zeros((n // 8 + 1, k // 64, 32, 2), dtype=torch.int32),
                                                                     "scales_and_zeros": b2.scales_and_zeros.data})
    # an unpacked checkpoint into a packed module: back to unpacked
    plain = mk()
    dst.load_state_dict(plain.state_dict())
    assert not dst.weight_reshaped and dst.weight.shape == (n, k)
    # inside a parent module (prefix handling), B side, any4
    parent = torch.nn.Sequential(modules.Any4Linear(k, n, bias=True, dtype=torch.bfloat16, group_size=g))
    parent[0].weight.data = torch.zeros((n // 8, k // 64, 32, 2), dtype=torch.int32)
    parent[0].weight_reshaped = True
    twin = torch.nn.Sequential(modules.Any4Linear(k, n, bias=True, dtype=torch.bfloat16, group_size=g))
    twin.load_state_dict(parent.state_dict())
    assert twin[0].weight_reshaped and twin[0].w_inner_k == 4 and twin[0].weight.shape == (n // 8, k // 64, 32, 2)
    # int8 packed shapes are read the same way
    i8 = modules.Int8Linear(k, n, bias=False, dtype=torch.bfloat16, group_size=g, kernel="linear_y_f16RM_x_f16RM_W_int8TC")
    i8.load_state_dict({"weight": torch.zeros((n // 8, k // 32, 32, 2), dtype=torch.int32), "scales_and_zeros": i8.scales_and_zeros.data})
    assert i8.weight_reshaped and i8.w_inner_k == 2
    a8 = modules.Int8Linear(k, n, bias=False, dtype=torch.bfloat16, group_size=g)
    a8.load_state_dict({"weight": torch.zeros((n // 16, k // 32, 32, 4), dtype=torch.int32), "scales_and_zeros": a8.scales_and_zeros.data})
    assert a8.weight_reshaped and a8.w_inner_k == 2


def test_reference_words_from_a_checkpoint_are_marked_for_one_repack():
    """modules.py:197-205: a CUDA-packed checkpoint for the weights-on-the-left kernels.  Loaded on the CPU, the module remembers that the
    tensor is to be repacked at its first forward on the GPU; with the opt-out (or a 'reference' process default) it is left alone."""
    import modules

    import any4_amd

    k, n, g = 256, 64, 64
    sd = {"weight": torch.randint(0, 2 ** 31 - 1, (n // 16, k // 64, 32, 4), dtype=torch.int32),
          "scales_and_zeros": torch.zeros((k // g, n, 2), dtype=torch.bfloat16)}
    lin = modules.Int4Linear(k, n, bias=False, dtype=torch.bfloat16, group_size=g)
    lin.load_state_dict(sd)
    assert lin.weight_format == "reference" and lin.__dict__["_relayout_pending"] is True and lin.w_inner_k == 4
    for ctx in (any4_amd.weight_format("reference"),):
        with ctx:
            lin2 = modules.Int4Linear(k, n, bias=False, dtype=torch.bfloat16, group_size=g)
            lin2.load_state_dict(sd)
            assert lin2.__dict__["_relayout_pending"] is False
    any4_amd.set_auto_relayout(False)
    try:
        lin3 = modules.Int4Linear(k, n, bias=False, dtype=torch.bfloat16, group_size=g)
        lin3.load_state_dict(sd)
        assert lin3.__dict__["_relayout_pending"] is False and not any4_amd.get_auto_relayout()
    finally:
        any4_amd.set_auto_relayout(True)
    # a B-side module / a native tensor: nothing to repack
    nat = modules.Int4Linear(k, n, bias=False, dtype=torch.bfloat16, group_size=g)
    nat.load_state_dict({"weight": torch.zeros((2 * n // 16, k // 64, 32, 2), dtype=torch.int32), "scales_and_zeros": sd["scales_and_zeros"]})
    assert nat.__dict__["_relayout_pending"] is False


def test_launch_plan_try_run_checks_what_the_plan_pins():
    """LaunchPlan.try_run (the eager hot path of a module): the recorded launch is re-issued only while the activations' shape / dtype /
    contiguity / alignment, the parameters' POINTERS (not versions: the struct points at their storage, and parameters made under
    torch.inference_mode() keep no version counter), the module's kernel attributes and the numerics / weight-format settings are those
    of the recording; anything else returns None and the module records again."""
    import any4_amd
    from any4_amd import _lib, ops

    with torch.inference_mode():
        w, q, lut = torch.zeros(64, dtype=torch.int32), torch.zeros(64, dtype=torch.bfloat16), torch.zeros(16, dtype=torch.bfloat16)
    lp = ops.LaunchPlan.__new__(ops.LaunchPlan)
    lp.args, lp._per_thread = _lib.W4Gemm(wrows=8), {}
    lp.m, lp.n, lp.k, lp.dtype, lp.device, lp.dev_index = 2, 8, 32, torch.bfloat16, torch.device("cpu"), 0
    lp.ptrs, lp.numerics, lp.wformat, lp.attrs = (w.data_ptr(), q.data_ptr(), lut.data_ptr()), ops.get_numerics(), ops.get_weight_format(), ("kern", 128, 4)
    launched = []
    ops.LaunchPlan._launch, saved = (lambda self, xp, y: launched.append((xp, tuple(y.shape))) or y), ops.LaunchPlan._launch
    try:
        x = torch.zeros((2, 32), dtype=torch.bfloat16)
        ok = lambda inp, ww=w, qq=q, ll=lut, attrs=("kern", 128, 4): lp.try_run(inp, ww, qq, ll, attrs)  # noqa: E731
        assert ok(x) is not None and launched[-1] == (x.data_ptr(), (2, 8))
        assert tuple(ok(x.view(1, 2, 32)).shape) == (1, 2, 8)                      # any leading shape with the same rows
        assert ok(torch.zeros((3, 32), dtype=torch.bfloat16)) is None               # another number of rows
        assert ok(x.to(torch.float16)) is None and ok(torch.zeros((2, 64), dtype=torch.bfloat16)[:, :32]) is None   # dtype, contiguity
        assert ok(torch.zeros(2 * 32 + 1, dtype=torch.bfloat16)[1:].view(2, 32)) is None                              # 16-byte alignment
        assert ok(x, ww=torch.zeros(64, dtype=torch.int32)) is None and ok(x, ll=None) is None                        # a re-assigned parameter
        assert ok(x, attrs=("other", 128, 4)) is None
        with torch.inference_mode():
            w.copy_(torch.ones_like(w))                                              # an in-place update: the same storage, the same plan
        assert ok(x) is not None
        with any4_amd.numerics("reference"):
            assert ok(x) is None
        with any4_amd.weight_format("reference"):
            assert ok(x) is None
    finally:
        ops.LaunchPlan._launch = saved


def test_launch_plan_is_per_thread_and_plan_sink_thread_local():
    """LaunchPlan.run fills x / y into a per-thread copy of the recorded struct (the recorded one is never written), and a plan
    being recorded on one thread does not see another thread's launches."""
    import threading

    from any4_amd import _lib, ops

    tmpl = _lib.W4Gemm(x=1, y=2, wrows=8)
    x = torch.empty((1, 32), dtype=torch.bfloat16)
    lp = ops.LaunchPlan.__new__(ops.LaunchPlan)
    lp.args, lp._per_thread = tmpl, {}
    seen, both = {}, threading.Barrier(2)

    def copy_for_thread(i):
        both.wait()                 # both threads alive at once (a finished thread's ident may be reused, harmlessly)
        a, ref = lp.thread_args()
        a.x = 100 + i
        seen[i] = a
        assert lp.thread_args()[0] is a
        both.wait()

    th = [threading.Thread(target=copy_for_thread, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert seen[0] is not seen[1] and (seen[0].x, seen[1].x) == (100, 101) and tmpl.x == 1
    # the sink of record_plan lives in thread-local storage
    got = []

    def other():
        got.append(getattr(ops._tls, "plan_sink", "unset"))

    def fn(_):
        t = threading.Thread(target=other)
        t.start()
        t.join()
        got.append(getattr(ops._tls, "plan_sink", "unset"))
        return torch.empty((1, 8))

    y, plan = ops.record_plan(fn, x, ("key",))
    assert got[0] == "unset" and got[1] == [] and plan is None and getattr(ops._tls, "plan_sink", None) is None
