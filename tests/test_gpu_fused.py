"""GPU suite: stages of a decoder layer fused into the 4-bit GEMM launch (tg_w4_gemm ABI 5, any4_amd.ops.w4_linear_fused):
RMSNorm of the activations in the kernel's staging, the residual add in its output store, SwiGLU of gate / up row pairs in its
output store -- each against the separate launch it replaces (include/decode_glue_hip.h), and the five-launch decoder layer
against the eight-launch one."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _layer(n, k, g, seed, dtype=torch.bfloat16, inner=4):
    import tinygemm  # noqa: F401

    gen = torch.Generator().manual_seed(seed)
    codes = torch.randint(0, 16, (n, k), dtype=torch.int32, generator=gen)
    lut = torch.randn(n, 16, generator=gen).to(dtype)
    std = 1.0 / math.sqrt(k)
    sz = torch.stack([(torch.rand(k // g, n, generator=gen) * 0.4 + 0.8) * std, torch.randn(k // g, n, generator=gen) * 0.05 * std], dim=2).to(dtype).contiguous()
    w = torch.ops.tinygemm.convert_matrix_to_m16n8k16_Bint4_layout(codes.to(DEV), inner)
    return w, sz.to(DEV), lut.to(DEV)


@pytest.mark.parametrize("m,n,k", [(1, 4096, 4096), (3, 512, 2048), (8, 256, 4096), (1, 1024, 14336)])
@pytest.mark.parametrize("numerics", ["fast", "reference"])
def test_residual_add_in_the_output_store(m, n, k, numerics):
    """y = RNE16(RNE16(acc) + residual[a][row]) in every kernel family (pair16, stream / split-K): the bits of the separate add."""
    import any4_amd
    from any4_amd import ops

    w, sz, lut = _layer(n, k, 128, seed=n + k)
    gen = torch.Generator().manual_seed(m)
    x = torch.randn(m, k, generator=gen).bfloat16().to(DEV)
    res = torch.randn(m, n, generator=gen).bfloat16().to(DEV)
    with any4_amd.numerics(numerics):
        plain = ops.w4_linear_fused(x, w, 128, sz, lut)
        fused = ops.w4_linear_fused(x, w, 128, sz, lut, residual=res)
        inplace = res.clone()
        assert ops.w4_linear_fused(x, w, 128, sz, lut, residual=inplace, out=inplace) is inplace
    want = plain + res  # bf16 + bf16 -> one rounding
    assert torch.equal(fused.view(torch.int16), want.view(torch.int16))
    assert torch.equal(inplace.view(torch.int16), want.view(torch.int16))


def test_residual_add_stacked_launch_and_a_side():
    """The same through the C ABI for the persistent pair kernel (stacked launch, B side and A side) with a strided residual."""
    import ctypes

    from any4_amd import _lib

    L = _lib.load()
    layers, m, n, k, g = 24, 2, 1024, 2048, 128
    for on_right in (True, False):
        import bench

        w, x, q, lut, y = bench.make_batch(layers, m, n, k, g, 4, torch.device(DEV), 3, "any4_rowwise", on_right)
        res = torch.randn(layers, m, n + 64, device=DEV).bfloat16()  # rows wider than n: a view into a larger residual stream
        y2 = torch.empty_like(y)
        for out, bias in ((y, None), (y2, res)):
            aa = bench.make_args(_lib, w, x, q, lut, out, m, n, k, g, "any4_rowwise", on_right, 4, layers)
            if bias is not None:
                aa.bias, aa.stride_bias, aa.bias_row_stride = bias.data_ptr(), bias.stride(0) * 2, bias.stride(1)
            ws = bench.attach_workspace(L, aa, torch.device(DEV))  # noqa: F841
            assert _lib.check(L.tg_gemm_w4(ctypes.byref(aa), 0, torch.cuda.current_stream().cuda_stream), "tg_gemm_w4") is None
        torch.cuda.synchronize()
        want = y + res[:, :, :n]
        assert torch.equal(y2.view(torch.int16), want.view(torch.int16)), on_right


def _interleave8(t):
    """[gate rows; up rows] -> blocks of 8 gate + 8 up rows."""
    half = t.shape[0] // 2
    return torch.stack([t[:half].reshape(-1, 8, *t.shape[1:]), t[half:].reshape(-1, 8, *t.shape[1:])], dim=1).reshape(t.shape)


@pytest.mark.parametrize("m,il,k,copies", [(1, 2048, 4096, 1), (4, 1792, 2048, 1), (8, 512, 4096, 1), (1, 14336, 4096, 1),
                                          (16, 14336, 4096, 1), (9, 4096, 4096, 1), (13, 2560, 4096, 1)])   # (9 ... 16 rows, more tiles than CUs: w4_gemm_pair16_loop_kernel)
def test_swiglu_in_the_output_store(m, il, k, copies):
    """gate_up GEMM with the SwiGLU epilogue (weight rows in blocks of 8 gate + 8 up) == dg_swiglu of the plain GEMM's output:
    same sums, same rounding points."""
    from any4_amd import decode_ops as G
    from any4_amd import ops

    n, g = 2 * il, 128
    w, sz, lut = _layer(n, k, g, seed=il)
    x = torch.randn(m, k, generator=torch.Generator().manual_seed(7)).bfloat16().to(DEV)
    gu = ops.w4_linear_fused(x, w, g, sz, lut)                       # rows in the weight's (interleaved) order
    act = ops.w4_linear_fused(x, w, g, sz, lut, swiglu=True)
    assert act is not None and act.shape == (m, il)
    split = gu.view(m, -1, 2, 8).transpose(1, 2).reshape(m, -1).contiguous()  # [gate | up]
    want = G.swiglu(split)
    assert torch.equal(act.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("m,n,k", [(1, 6144, 4096), (1, 28672, 4096), (2, 512, 2048), (8, 1024, 4096), (1, 4096, 14336),
                                   (16, 6144, 4096), (9, 28672, 4096), (13, 8192, 4096)])   # (9 ... 16 rows, more tiles than CUs: w4_gemm_pair16_loop_kernel)
def test_rmsnorm_in_the_activation_staging(oracle, m, n, k):
    """GEMM with norm_weight against the CPU oracle's group-scaled contraction of dg_add_rmsnorm's output (the separate launch it
    replaces).  The fused kernels add the squares in another order, so 1 / rms can differ in its last bit and a few normalised
    activations by one bf16 step: the tolerance is the fast kernels' own plus 2^-9 sum|x w| for that."""
    import numpy as np

    from any4_amd import decode_ops as G
    from any4_amd import ops
    from tests.conftest import bits16

    g, rows = 128, 192
    w, sz, lut = _layer(n, k, g, seed=n)
    gen = torch.Generator().manual_seed(11)
    x = (torch.randn(m, k, generator=gen) * 3).bfloat16().to(DEV)
    nw = (1 + 0.1 * torch.randn(k, generator=gen)).bfloat16().to(DEV)
    fused = ops.w4_linear_fused(x, w, g, sz, lut, norm_weight=nw, norm_eps=1e-5)
    assert fused is not None
    xn = G.add_rmsnorm(x.clone(), None, nw, 1e-5)[1].cpu()
    codes = oracle.unpack_Bint4(w.cpu().numpy(), n, k)[:rows]
    qi, lb = bits16(sz.cpu()[:, :rows].contiguous()), bits16(lut.cpu()[:rows])
    _, y_gs = oracle.linear_group_scaled(bits16(xn), codes, g, oracle.Q_ANY4_ROWWISE, qi, lb)
    wq = oracle.bf16_to_f32(oracle.dequant(codes, g, oracle.Q_ANY4_ROWWISE, qi, lb)).astype(np.float64)
    S = np.abs(xn.double().numpy()) @ np.abs(wq).T
    got = fused[:, :rows].double().cpu().numpy()
    ulp = np.exp2(np.floor(np.log2(np.maximum(np.abs(y_gs), 1e-30))) - 7)
    err = np.abs(got - y_gs)
    assert (err <= 0.5 * ulp * (1 + 2.0 ** -7) + (4e-6 + 2.0 ** -9) * S).all(), err.max()
    # ... and a shifted norm weight must show: the stage is really applied, with the right weights at the right k
    nw2 = nw.clone()
    nw2[k // 2:] *= 2
    fused2 = ops.w4_linear_fused(x, w, g, sz, lut, norm_weight=nw2, norm_eps=1e-5)
    assert (fused2.float() - fused.float()).abs().max() > 0.05


@pytest.mark.parametrize("m,il", [(16, 14336), (11, 4096)])
def test_rmsnorm_and_swiglu_together_at_9_to_16_rows(m, il):
    """Both stages in one launch of the loop kernel (a gate_up projection of a 9 ... 16-sequence decode step): the bits of dg_swiglu applied
    to the norm-fused launch's own output."""
    from any4_amd import decode_ops as G
    from any4_amd import ops

    n, k, g = 2 * il, 4096, 128
    w, sz, lut = _layer(n, k, g, seed=il + m)
    gen = torch.Generator().manual_seed(17)
    x = (torch.randn(m, k, generator=gen) * 2).bfloat16().to(DEV)
    nw = (1 + 0.1 * torch.randn(k, generator=gen)).bfloat16().to(DEV)
    gu = ops.w4_linear_fused(x, w, g, sz, lut, norm_weight=nw, norm_eps=1e-5)
    act = ops.w4_linear_fused(x, w, g, sz, lut, norm_weight=nw, norm_eps=1e-5, swiglu=True)
    assert gu is not None and act is not None and act.shape == (m, il)
    want = G.swiglu(gu.view(m, -1, 2, 8).transpose(1, 2).reshape(m, -1).contiguous())
    assert torch.equal(act.view(torch.int16), want.view(torch.int16))


def test_fusion_not_available_is_reported_not_faked():
    import any4_amd
    from any4_amd import ops

    w, sz, lut = _layer(256, 1024, 128, seed=1)
    x = torch.randn(1, 1024).bfloat16().to(DEV)
    nw = torch.ones(1024).bfloat16().to(DEV)
    assert ops.w4_linear_fused(x, w, 128, sz, lut, norm_weight=nw) is None          # k % 2048 != 0
    w2, sz2, lut2 = _layer(256, 2048, 128, seed=2)
    x2 = torch.randn(1, 2048).bfloat16().to(DEV)
    nw2 = torch.ones(2048).bfloat16().to(DEV)
    assert ops.w4_linear_fused(x2, w2, 128, sz2, lut2, norm_weight=nw2) is not None
    with any4_amd.numerics("reference"):                                            # the reference-numerics kernels have no fused stages
        assert ops.w4_linear_fused(x2, w2, 128, sz2, lut2, norm_weight=nw2) is None
        assert ops.w4_linear_fused(x2, w2, 128, sz2, lut2, swiglu=True) is None


@pytest.mark.parametrize("bs", [1, 4])
def test_five_launch_layer_matches_eight_launch_layer(bs):
    """A Llama-shaped stack (hidden 2048) with the stages fused into the GEMM launches against the same stack (same weights) on
    the separate glue kernels, eager and from a hipGraph."""
    from any4_amd.decode import Any4Factory, DecodeConfig, DecodeStack

    cfg = DecodeConfig(hidden=2048, inter=4096, layers=2, heads=16, kv_heads=4, head_dim=128, vocab=512, max_seq=64, gate_up_interleave=8)
    a = DecodeStack(cfg, Any4Factory(cfg, DEV, seed=3), DEV, bs=bs, seed=9)
    b = DecodeStack(cfg, Any4Factory(cfg, DEV, seed=3), DEV, bs=bs, seed=9, fuse_gemm_stages=False)
    toks = torch.randint(0, cfg.vocab, (6, bs), generator=torch.Generator().manual_seed(1)).to(DEV)
    for i, t in enumerate(toks[:3]):
        ya, yb = a.decode(t, i).float(), b.decode(t, i).float()
        assert torch.isfinite(ya).all()
        assert (ya - yb).abs().max() <= 0.02 * yb.abs().max() + 1e-3, (i, (ya - yb).abs().max(), yb.abs().max())
    if bs == 1:
        assert a.layers[0]._fuse == {"norm1": True, "norm2": None, "mlp": True}, a.layers[0]._fuse  # every stage found its fused kernel
    a.capture()
    for i, t in enumerate(toks[3:], start=3):
        ya, yb = a.decode(t, i).float().clone(), b.decode(t, i).float()
        assert (ya - yb).abs().max() <= 0.02 * yb.abs().max() + 1e-3, (i, (ya - yb).abs().max(), yb.abs().max())
