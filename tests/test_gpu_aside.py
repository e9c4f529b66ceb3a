"""GPU suite (-m gpu): weights on the LEFT (the reference's `..._W_int4TC_x_f16RM` / `weightOnRight=False` ops, Int4Linear's default
kernel, modules.py:21) in the library's default packed format.

SURVEY 8(b): the packed tensor is opaque to every caller of the reference (TinyGemm_int4.cu:322-364 checks its shape only), so
`convert_matrix_to_m16n8k16_Aint4_layout` may return a gfx950-native order as long as the GEMM ops accept what it returns and the
reference's own words stay available as an interchange format with a lossless repack.  Here: the native tensor IS the Bint4 tensor
of the weight rows padded to 16 -- checked bit for bit against the oracle's Bint4 packer; its shape differs from the reference's Aint4
shape for every k, so a GEMM op / module / state_dict reads the format off the tensor (ops.aside_format) -- the repack is checked both ways, and the GEMM through every op flavour is checked against the oracle like the weights-on-the-right
path (tests/test_gpu_gemv.py, test_gpu_fast.py): the same kernels run it.
"""
import numpy as np
import pytest
import torch

from tests.test_gpu_fast import QT
from tests.test_gpu_gemv import check
from tests.test_gpu_parity import DEV, T, assert_gemm_close, oracle_weights, rand_problem  # noqa: F401  (T is a fixture)

pytestmark = pytest.mark.gpu


def native_words(oracle, codes, k):
    """What the native convert must return, from the oracle's Bint4 packer: rows padded to 16, innerKTiles 4 (k % 64 == 0) or 2."""
    m = codes.shape[0]
    pad = -(-m // 16) * 16
    full = np.zeros((pad, k), np.int32)
    full[:m] = codes
    return oracle.pack_Bint4(full, 4 if k % 64 == 0 else 2)


@pytest.mark.parametrize("inner", [1, 2, 4])
@pytest.mark.parametrize("m,k", [(16, 64), (40, 256), (72, 96), (4096, 4096), (8, 32), (129, 1024)])
def test_native_convert_bit_exact_and_repack(T, oracle, inner, m, k):
    import any4_amd
    from any4_amd import ops

    if k % (16 * inner):
        pytest.skip("k must be a multiple of innerKTiles * 16 for a GEMM-able tensor")
    codes = torch.randint(0, 16, (m, k), dtype=torch.int32, generator=torch.Generator().manual_seed(m + k))
    assert any4_amd.get_weight_format() == "native"
    nat = T.convert_matrix_to_m16n8k16_Aint4_layout(codes.to(DEV), inner)
    j = 4 if k % 64 == 0 else 2
    assert tuple(nat.shape) == (2 * -(-m // 16), k // (16 * j), 32, j // 2)  # the Bint4 tensor of the 16-row-padded codes
    assert ops.aside_format(nat, k) == "native"
    assert np.array_equal(nat.cpu().numpy().reshape(-1), native_words(oracle, codes.numpy(), k).reshape(-1))
    with any4_amd.weight_format("reference"):
        ref = T.convert_matrix_to_m16n8k16_Aint4_layout(codes.to(DEV), inner)
    assert np.array_equal(ref.cpu().numpy(), oracle.pack_Aint4(codes.numpy(), inner))  # the interchange format: the reference's words
    assert tuple(ref.shape) == (-(-m // 16), k // (16 * inner), 32, inner) and ops.aside_format(ref, k) == "reference"  # TinyGemm_int4.cu:340-364
    # unpack (both orders) and the lossless repack, both ways
    rows = ref.shape[0] * 16
    want = np.zeros((rows, k), np.int32)
    want[:m] = codes.numpy()
    assert np.array_equal(ops.unpack_int4(nat, rows, k, "B").cpu().numpy(), want)
    assert np.array_equal(ops.unpack_int4(ref, rows, k, "A").cpu().numpy(), want)
    assert torch.equal(ops.relayout_Aint4(ref, k, "native"), nat)
    assert torch.equal(ops.relayout_Aint4(nat, k, "reference", inner), ref)
    assert ops.relayout_Aint4(nat, k, "native") is nat and ops.relayout_Aint4(ref, k, "reference") is ref
    wb = T.convert_matrix_to_m16n8k16_Bint4_layout(codes.to(DEV), 4 if k % 64 == 0 else 2)
    assert np.array_equal(ops.unpack_int4(wb, m, k, "B").cpu().numpy(), codes.numpy())


def run_left(T, codes, x, qinfo, lut, g, qtype, inner):
    d = lambda t: None if t is None else t.to(DEV)
    w2 = T.convert_matrix_to_m16n8k16_Aint4_layout(d(codes), inner)
    if qtype == "mx4":
        return T.tinygemm_y_f16RM_x_f16RM_w_mx4TC(w2, d(x), g, d(qinfo), False)
    if qtype == "int4":
        return T.tinygemm_y_f16RM_x_f16RM_w_int4TC(w2, d(x), g, d(qinfo), False)
    return T.tinygemm_y_f16RM_x_f16RM_w_any4TC(w2, d(x), g, d(qinfo), d(lut), False)


def pad_rows(codes, qinfo, lut, qtype):
    """quantisation info / LUT rows for the 16-row tile padding of the A side (the reference checks them against 16 * size(0))."""
    n = codes.shape[0]
    pad = -(-n // 16) * 16 - n
    if not pad:
        return qinfo, lut
    if qtype == "mx4":
        qinfo = torch.cat([qinfo, torch.full((pad, qinfo.shape[1]), 127, dtype=qinfo.dtype)])
    else:
        qinfo = torch.cat([qinfo, torch.zeros(qinfo.shape[0], pad, 2, dtype=qinfo.dtype)], dim=1)
    if lut is not None and lut.dim() == 2:
        lut = torch.cat([lut, torch.zeros(pad, 16, dtype=lut.dtype)])
    return qinfo, lut


@pytest.mark.parametrize("case", [
    # (n, k, m, g, qtype, inner)
    (4096, 4096, 1, 128, "any4_rowwise", 4),   # Int4Linear / Any4Linear with the weights on the left at batch 1: w4_gemv_kernel
    (4096, 4096, 1, 128, "int4", 4), (4096, 4096, 3, 64, "any4_global", 2), (6144, 4096, 1, 256, "int4", 1),
    (200, 512, 2, 32, "any4_rowwise", 4),      # ragged 16-row tile
    (1024, 4096, 8, 128, "any4_rowwise", 4), (1024, 4096, 16, 128, "int4", 4),   # m > 4: w4_gemm_pair16_kernel
    (2048, 2048, 7, 32, "mx4", 4), (512, 14336, 1, 128, "any4_rowwise", 4), (264, 96, 5, 32, "int4", 2),  # k % 64 != 0: innerKTiles 2 words
])
def test_left_side_gemm_vs_oracle(T, oracle, case):
    from any4_amd import ops

    n, k, m, g, qtype, inner = case
    codes, x, qinfo, lut = rand_problem(n, k, g, m, qtype, seed=n + k + m)
    qp, lp = pad_rows(codes, qinfo, lut, qtype)
    y = run_left(T, codes, x, qp, lp, g, qtype, inner)
    plan = ops.gemm_w4_plan(m, -(-n // 16) * 16, k, g, QT[qtype], False, inner)
    assert plan in ("gemv", "pair"), plan  # a group-scaled kernel of the B side, never the Aint4 fallbacks
    if qtype == "mx4":
        assert_gemm_close(y[:, :n], x, oracle_weights(oracle, codes, g, qtype, qinfo, lut))
    else:
        rows = None if n <= 2048 else np.unique(np.concatenate([np.arange(0, 96), np.arange(n // 2 - 48, n // 2 + 48), np.arange(n - 96, n)]))
        check(oracle, y, codes, x, qinfo, lut, g, qtype, rows=rows)


def test_left_side_equals_right_side_bits(T):
    """Same codes, same kernels: the weights-on-the-left op returns the bits of the weights-on-the-right op."""
    n, k, g = 4096, 4096, 128
    for m in (1, 4, 8):
        codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_rowwise", seed=m)
        d = lambda t: t.to(DEV)
        wl = T.convert_matrix_to_m16n8k16_Aint4_layout(d(codes), 4)
        wr = T.convert_matrix_to_m16n8k16_Bint4_layout(d(codes), 4)
        assert torch.equal(wl.view(-1), wr.view(-1))
        yl = T.tinygemm_y_f16RM_x_f16RM_w_any4TC(wl, d(x), g, d(qinfo), d(lut), False)
        yr = T.tinygemm_y_f16RM_x_f16RM_w_any4TC(d(x), wr, g, d(qinfo), d(lut), True)
        assert torch.equal(yl.view(torch.int16), yr.view(torch.int16))


def test_reference_words_still_served_and_modules_remember_their_format(T, oracle):
    """A tensor in the reference's Aint4 words (a checkpoint packed by the CUDA implementation) runs whatever the process default
    is: the GEMM ops read the format off the tensor; relayout() converts a module's tensor once."""
    import any4_amd
    import modules

    n, k, g = 512, 1024, 128
    codes, x, qinfo, lut = rand_problem(n, k, g, 2, "int4", seed=9)
    with any4_amd.weight_format("reference"):
        w_ref = T.convert_matrix_to_m16n8k16_Aint4_layout(codes.to(DEV), 4)
    # consumed OUTSIDE the context, under the 'native' process default: the tensor says what it is
    y_ref = T.tinygemm_y_f16RM_x_f16RM_w_int4TC(w_ref, x.to(DEV), g, qinfo.to(DEV), False)
    y_nat = run_left(T, codes, x, qinfo, lut, g, "int4", 4)
    with any4_amd.weight_format("reference"):   # ... and a native tensor under the 'reference' default
        w_nat = T.convert_matrix_to_m16n8k16_Bint4_layout(codes.to(DEV), 4)
        assert torch.equal(T.tinygemm_y_f16RM_x_f16RM_w_int4TC(w_nat, x.to(DEV), g, qinfo.to(DEV), False), y_nat)
    # two kernel families (the Aint4 words run a reference-numerics kernel at this size, the native ones the group-scaled gemv):
    # one result within the reference's own weight rounding
    tol = 0.02 * y_nat.float().abs().max()
    assert (y_ref.float() - y_nat.float()).abs().max() <= tol
    lin = modules.Int4Linear(k, n, bias=False, device=DEV, dtype=torch.bfloat16, group_size=g)  # the default kernel: weights on the left
    assert lin.kernel == "linear_y_f16RM_W_int4TC_x_f16RM"
    lin.weight.data = codes.to(DEV)
    lin.scales_and_zeros.data = qinfo.to(DEV)
    with any4_amd.weight_format("reference"):
        lin.reshape_weight(4)
    assert lin.weight_format == "reference" and np.array_equal(lin.weight.cpu().numpy(), oracle.pack_Aint4(codes.numpy(), 4))
    y1 = lin(x.to(DEV))                       # whatever the process default is
    lin.relayout("native")
    assert lin.weight_format == "native" and np.array_equal(lin.weight.cpu().numpy().reshape(-1), native_words(oracle, codes.numpy(), k).reshape(-1))
    y2 = lin(x.to(DEV))
    assert torch.equal(y2, y_nat) and (y1.float() - y2.float()).abs().max() <= tol


def test_left_side_tensor_core_layout_ops(T, oracle):
    """`tinygemm_y_f16TC_x_f16TC_w_int4TC` with the weights on the left (activations / output in B-fragment order) on native words."""
    n, k, g, m = 256, 512, 64, 5
    codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_rowwise", seed=4)
    d = lambda t: t.to(DEV)
    w2 = T.convert_matrix_to_m16n8k16_Aint4_layout(d(codes), 2)
    xb = T.convert_matrix_to_m16n8k16_B_layout(d(x), 1)
    yb = T.tinygemm_y_f16TC_x_f16TC_w_any4TC(w2, xb, g, d(qinfo), d(lut), False)
    y = T.convert_matrix_from_m16n8k16_B_layout(yb, m, n)
    check(oracle, y, codes, x, qinfo, lut, g, "any4_rowwise")


@pytest.mark.parametrize("fmt", ["native", "reference"])
@pytest.mark.parametrize("cls", ["Int4Linear", "Any4Linear"])
def test_state_dict_round_trip_into_a_fresh_module(T, fmt, cls):
    """eval.py:180-210 saves / loads state_dicts of quantised models.  A packed weights-on-the-left module -> state_dict -> a FRESH
    module (unpacked shape, no attributes set) under the OTHER process default gives the same bits: weight_reshaped / w_inner_k
    travel in the extra state, the packed format in the tensor's shape."""
    import io

    import any4_amd
    import modules

    n, k, g = 512, 1024, 128
    codes, x, qinfo, lut = rand_problem(n, k, g, 2, "any4_rowwise" if cls == "Any4Linear" else "int4", seed=11)
    kernel = "linear_y_f16RM_W_any4TC_x_f16RM" if cls == "Any4Linear" else "linear_y_f16RM_W_int4TC_x_f16RM"

    def fresh():
        return getattr(modules, cls)(k, n, bias=False, device=DEV, dtype=torch.bfloat16, group_size=g, kernel=kernel)

    src = fresh()
    src.weight.data, src.scales_and_zeros.data = codes.to(DEV), qinfo.to(DEV)
    if cls == "Any4Linear":
        src.lut.data = lut.to(DEV)
    with any4_amd.weight_format(fmt):
        src.reshape_weight(2 if fmt == "reference" else 4)
    assert src.weight_format == fmt
    want = src(x.to(DEV))
    buf = io.BytesIO()
    torch.save(src.state_dict(), buf)
    buf.seek(0)
    sd = torch.load(buf)
    assert set(sd) == {"weight", "scales_and_zeros"} | ({"lut"} if cls == "Any4Linear" else set())    # tensors only: the reference's keys
    other = "reference" if fmt == "native" else "native"
    with any4_amd.weight_format(other):
        dst = fresh()
        assert not dst.weight_reshaped
        dst.load_state_dict(sd)                      # strict
        # native words stay; the reference's words under the 'native' default are repacked once at load (modules.py:197-205 checkpoints)
        assert dst.weight_reshaped and dst.weight_format == "native"
        assert fmt == "native" or dst.w_inner_k == src.w_inner_k
        if fmt == "native":
            assert torch.equal(dst(x.to(DEV)), want)
        else:
            assert torch.equal(dst.weight, ops_relayout(src))
            y = dst(x.to(DEV))
            assert (y.float() - want.float()).abs().max() <= 0.02 * want.float().abs().max()   # (two kernel families, one result within the reference's own weight rounding)
    # the same checkpoint INTO AN ALREADY PACKED module of the other format (quantize_model ends in reshape_weight())
    dst3 = fresh()
    dst3.weight.data, dst3.scales_and_zeros.data = codes.to(DEV), qinfo.to(DEV)
    with any4_amd.weight_format(other):
        dst3.reshape_weight(4 if other == "native" else 2)
    assert dst3.weight_format == other
    any4_amd.set_auto_relayout(False)                # opt-out: the words are kept as they came
    try:
        dst3.load_state_dict(sd)
        assert dst3.weight_format == fmt and torch.equal(dst3.weight, src.weight) and torch.equal(dst3(x.to(DEV)), want)
        # loaded on the CPU, moved to the GPU: (with the repack enabled) done at the first forward
        cpu = getattr(modules, cls)(k, n, bias=False, dtype=torch.bfloat16, group_size=g, kernel=kernel)
        any4_amd.set_auto_relayout(True)
        cpu.load_state_dict({k_: v.cpu() for k_, v in sd.items()})
        assert cpu.weight_format == fmt
        cpu = cpu.to(DEV)
        y = cpu(x.to(DEV))
        assert cpu.weight_format == "native" and (y.float() - want.float()).abs().max() <= 0.02 * want.float().abs().max()
        assert torch.equal(cpu(x.to(DEV)), y)
    finally:
        any4_amd.set_auto_relayout(True)
    # a tensor packed for the other operand side is refused, not mis-multiplied (a native tensor IS a Bint4 tensor: either side takes it)
    wrong = getattr(modules, cls)(k, n, bias=False, device=DEV, dtype=torch.bfloat16, group_size=g,
                                  kernel="linear_y_f16RM_x_f16RM_W_any4TC" if cls == "Any4Linear" else "linear_y_f16RM_x_f16RM_W_int4TC")
    if fmt == "reference":
        with pytest.raises(RuntimeError, match="packed for kernel"):
            wrong.load_state_dict(sd)
    else:
        wrong.load_state_dict(sd)
        assert (wrong(x.to(DEV)).float() - want.float()).abs().max() <= 0.02 * want.float().abs().max()


def ops_relayout(mod):
    from any4_amd import ops
    return ops.relayout_Aint4(mod.weight.data, mod.in_features, "native")


def test_same_module_from_two_threads_on_two_streams(T):
    """The reference's host functions are stateless and re-entrant (TinyGemm_int4.cu:41-42).  Two host threads, each on its own
    stream, run the SAME module (one recorded launch plan) on different activations many times: every result is its own."""
    import threading

    import modules

    n, k, g = 4096, 4096, 128
    codes, x, qinfo, lut = rand_problem(n, k, g, 1, "any4_rowwise", seed=3)
    lin = modules.Any4Linear(k, n, bias=False, device=DEV, dtype=torch.bfloat16, group_size=g)
    lin.weight.data, lin.scales_and_zeros.data, lin.lut.data = codes.to(DEV), qinfo.to(DEV), lut.to(DEV)
    lin.reshape_weight()
    xs = [(x * s).to(DEV) for s in (1.0, -2.0)]
    want = [lin(xi).clone() for xi in xs]           # (also records the plan)
    assert lin.__dict__["_plan"] is not None
    torch.cuda.synchronize()
    bad, go = [], threading.Barrier(2)

    def work(i):
        st = torch.cuda.Stream()
        go.wait()
        with torch.cuda.stream(st):
            for _ in range(2000):
                y = lin(xs[i])
                if _ % 100 == 0 and not torch.equal(y, want[i]):
                    bad.append(i)
            st.synchronize()
            if not torch.equal(lin(xs[i]), want[i]):
                bad.append(i)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not bad
    assert len(lin.__dict__["_plan"]._per_thread) >= 2   # every thread filled in its own copy of the argument struct


def test_launch_plan_with_a_workspace_two_threads(T, oracle):
    """A module at 17 ... ~256 rows runs the tile GEMM as a split-K launch: the recorded launch plan carries the workspace SIZE and every
    replay takes a fresh scratch from the (stream-ordered) caching allocator.  Two threads on two streams, the same module: every result is
    its own, equal to the first (the split sum is deterministic), and right against the oracle."""
    import threading

    import modules
    from any4_amd import ops

    n, k, g, m = 1024, 4096, 128, 96
    codes, x, qinfo, lut = rand_problem(n, k, g, m, "any4_rowwise", seed=11)
    lin = modules.Any4Linear(k, n, bias=False, device=DEV, dtype=torch.bfloat16, group_size=g)
    lin.weight.data, lin.scales_and_zeros.data, lin.lut.data = codes.to(DEV), qinfo.to(DEV), lut.to(DEV)
    lin.reshape_weight()
    assert ops.gemm_w4_plan(m, n, k, g, 2, True, 4) == "tile"
    xs = [(x * s).to(DEV) for s in (1.0, -0.5)]
    want = [lin(xi).clone() for xi in xs]
    plan = lin.__dict__["_plan"]
    assert plan is not None and plan.ws_bytes >= 2 * m * n * 4
    assert_gemm_close(want[0].cpu(), x, oracle_weights(oracle, codes, g, "any4_rowwise", qinfo, lut, torch.bfloat16), torch.bfloat16)
    torch.cuda.synchronize()
    bad, go = [], threading.Barrier(2)

    def work(i):
        st = torch.cuda.Stream()
        go.wait()
        with torch.cuda.stream(st):
            for it in range(400):
                y = lin(xs[i])
                if it % 50 == 0 and not torch.equal(y, want[i]):
                    bad.append(i)
            st.synchronize()
            if not torch.equal(lin(xs[i]), want[i]):
                bad.append(i)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not bad
