"""GPU suite: the decode harness (any4_amd/decode.py) on the HIP linears.

A small Llama-shaped stack whose linears are `Any4Linear` modules built from known codes / LUTs / scales is
compared with the SAME stack on 16-bit `nn.Linear`s holding the oracle's dequantised weights; eager, hipGraph
replay and a k-means-free random `Any4Factory` stack are exercised."""
import math

import numpy as np
import pytest
import torch

from tests.conftest import bits16, from_bits16
from tests.test_gpu_parity import T  # noqa: F401  (fixture: torch.ops.tinygemm)

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("reference_numerics")]
DEV = "cuda:0"
CFG = dict(hidden=256, inter=512, layers=2, heads=4, kv_heads=2, head_dim=64, vocab=128, max_seq=32, group_size=64)


class _PairedFactories:
    """any4(name, layer, k, rows) -> Any4Linear from explicit tensors; dense(...) -> nn.Linear with the oracle's
    dequantisation of the same tensors."""

    def __init__(self, oracle, cfg, kernel):
        self.oracle, self.cfg, self.kernel, self.w = oracle, cfg, kernel, {}

    def any4(self, name, layer, k, rows):
        import modules

        g = self.cfg.group_size
        gen = torch.Generator().manual_seed(100 * layer + len(name))
        codes = torch.randint(0, 16, (rows, k), dtype=torch.int32, generator=gen)
        lut = torch.randn(rows, 16, generator=gen).to(torch.bfloat16)
        std = 1.0 / math.sqrt(k)
        scales = ((torch.rand(k // g, rows, generator=gen) * 0.4 + 0.8) * std).to(torch.bfloat16)
        zeros = (torch.randn(k // g, rows, generator=gen) * 0.05 * std).to(torch.bfloat16)
        sz = torch.stack([scales, zeros], dim=2).contiguous()
        wb = self.oracle.dequant(codes.numpy(), g, self.oracle.Q_ANY4_ROWWISE, bits16(sz), bits16(lut), self.oracle.BF16)
        self.w[(name, layer)] = from_bits16(wb, torch.bfloat16)
        mod = modules.Any4Linear(k, rows, bias=False, device=DEV, dtype=torch.bfloat16, group_size=g, kernel=self.kernel)
        mod.weight.data = codes.to(DEV)
        mod.lut.data = lut.to(DEV)
        mod.scales_and_zeros.data = sz.to(DEV)
        mod.reshape_weight(4)
        return mod

    def dense(self, name, layer, k, rows):
        lin = torch.nn.Linear(k, rows, bias=False, device=DEV, dtype=torch.bfloat16)
        lin.weight.data = self.w[(name, layer)].to(DEV)
        return lin


@pytest.mark.parametrize("kernel", ["linear_y_f16RM_x_f16RM_W_any4TC", "linear_y_f16RM_W_any4TC_x_f16RM"])
@pytest.mark.parametrize("bs", [1, 3])
@pytest.mark.parametrize("fused", [False, True])
def test_decode_any4_vs_dense_dequantised(oracle, kernel, bs, fused):
    from any4_amd.decode import DecodeConfig, DecodeStack

    cfg = DecodeConfig(**CFG)
    fac = _PairedFactories(oracle, cfg, kernel)
    q = DecodeStack(cfg, fac.any4, DEV, torch.bfloat16, bs=bs, seed=5, fused=fused)
    d = DecodeStack(cfg, fac.dense, DEV, torch.bfloat16, bs=bs, seed=5, fused=False)  # plain torch ops throughout
    toks = torch.randint(0, cfg.vocab, (6, bs), generator=torch.Generator().manual_seed(1)).to(DEV)
    for i, t in enumerate(toks):
        a, b = q.decode(t, i).float(), d.decode(t, i).float()
        assert torch.isfinite(a).all()
        # same weights, same 16-bit pipeline; only the fp32 summation order inside the GEMMs differs
        assert (a - b).abs().max() <= 0.03 * b.abs().max() + 1e-3, (i, (a - b).abs().max(), b.abs().max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_glue_kernels_vs_torch(dtype):
    """Each HIP glue kernel against the torch formulation it replaces (any4_amd/decode.py), same rounding points."""
    from any4_amd import decode_ops as G
    from any4_amd.decode import RMSNorm, _rope, _rope_tables, DecodeConfig

    gen = torch.Generator(device=DEV).manual_seed(0)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11

    def close(a, b, k=2.0):
        a, b = a.float(), b.float()
        return bool(((a - b).abs() <= k * ulp * b.abs().clamp_min(1e-2)).all())

    # residual add + rmsnorm
    bs, dim = 3, 1024
    h = torch.randn(bs, dim, device=DEV, generator=gen).to(dtype)
    dl = torch.randn(bs, dim, device=DEV, generator=gen).to(dtype)
    norm = RMSNorm(dim, 1e-5, DEV, dtype)
    norm.weight.data = (torch.rand(dim, device=DEV, generator=gen) + 0.5).to(dtype)
    want_h = h + dl
    want_y = norm(want_h)
    got_h, got_y = G.add_rmsnorm(h.clone(), dl, norm.weight, 1e-5)
    assert torch.equal(got_h, want_h) and close(got_y, want_y)
    _, y0 = G.add_rmsnorm(h.clone(), None, norm.weight, 1e-5)
    assert close(y0, norm(h))

    # rope + cache write, attention
    cfg = DecodeConfig(hidden=256, heads=4, kv_heads=2, head_dim=64, max_seq=48)
    hl, kvl, d, S = 4, 2, 64, 48
    cos, sin = _rope_tables(cfg, DEV)
    kc = torch.randn(bs, kvl, S, d, device=DEV, generator=gen).to(dtype)
    vc = torch.randn(bs, kvl, S, d, device=DEV, generator=gen).to(dtype)
    for p in (0, 1, 17, 47):
        pos = torch.tensor([p], device=DEV)
        qkv = torch.randn(bs, (hl + 2 * kvl) * d, device=DEV, generator=gen).to(dtype)
        c, s_ = cos[p].view(1, 1, -1), sin[p].view(1, 1, -1)
        want_q = _rope(qkv[:, : hl * d].reshape(bs, hl, d), c, s_)
        want_k = _rope(qkv[:, hl * d: (hl + kvl) * d].reshape(bs, kvl, d), c, s_)
        want_v = qkv[:, (hl + kvl) * d:].reshape(bs, kvl, d)
        kc2, vc2 = kc.clone(), vc.clone()
        got_q = G.rope_kv(qkv, cos, sin, pos, kc2, vc2, hl, kvl, d)
        assert torch.equal(got_q, want_q) and torch.equal(kc2[:, :, p], want_k) and torch.equal(vc2[:, :, p], want_v)
        keep = torch.arange(S, device=DEV) != p
        assert torch.equal(kc2[:, :, keep], kc[:, :, keep]) and torch.equal(vc2[:, :, keep], vc[:, :, keep])
        # attention over positions 0..p
        scale = 1.0 / math.sqrt(d)
        qg = got_q.reshape(bs, kvl, hl // kvl, d)
        att = torch.matmul(qg, kc2.transpose(2, 3)).float() * scale
        att = att.masked_fill((torch.arange(S, device=DEV) > p).view(1, 1, 1, -1), float("-inf")).softmax(-1).to(dtype)
        want = torch.matmul(att, vc2).reshape(bs, hl * d)
        got = G.decode_attn(got_q, kc2, vc2, pos, scale)
        assert (got.float() - want.float()).abs().max() <= 4 * ulp * want.float().abs().max(), p
        # the fused launch (rope + cache write + attention) is bit-identical to the two separate ones
        kc3, vc3 = kc.clone(), vc.clone()
        fused = G.rope_attn(qkv, cos, sin, pos, kc3, vc3, hl, kvl, d, scale)
        assert torch.equal(fused, got) and torch.equal(kc3, kc2) and torch.equal(vc3, vc2), p
        # the latency-built launch (one barrier, flash-decoding style statistics): same caches bit for bit, output within 16-bit rounding
        kc5, vc5 = kc.clone(), vc.clone()
        on = G.rope_attn_online(qkv, cos, sin, pos, kc5, vc5, hl, kvl, d, scale)
        assert torch.equal(kc5, kc2) and torch.equal(vc5, vc2), p
        assert (on.float() - want.float()).abs().max() <= 4 * ulp * want.float().abs().max(), p
        # split-sequence variant: same caches, output within 16-bit rounding; replayable (counters self-reset)
        for ns in (2, 3, 8):
            scr = G.rope_attn_split_scratch(bs, hl, d, ns, DEV)
            for _ in range(2):
                kc4, vc4 = kc.clone(), vc.clone()
                sp = G.rope_attn_split(qkv, cos, sin, pos, kc4, vc4, hl, kvl, d, scale, scr, ns)
                assert torch.equal(kc4, kc2) and torch.equal(vc4, vc2), (p, ns)
                assert (sp.float() - want.float()).abs().max() <= 4 * ulp * want.float().abs().max(), (p, ns)

    # swiglu
    gu = torch.randn(bs, 2 * 512, device=DEV, generator=gen).to(dtype) * 3
    want = torch.nn.functional.silu(gu[:, :512]) * gu[:, 512:]
    assert close(G.swiglu(gu), want)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        G.swiglu(gu.cpu())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_rope_attn_online_llama_heads(dtype):
    """dg_rope_attn_online at Llama-3-8B's head geometry (32 / 8 heads of 128) against dg_rope_attn: positions inside the first
    chunk of 256, on its edge, and several chunks in; an UNINITIALISED (NaN-filled) cache beyond the written positions."""
    from any4_amd import decode_ops as G
    from any4_amd.decode import DecodeConfig, _rope, _rope_tables

    gen = torch.Generator(device=DEV).manual_seed(1)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    bs, hl, kvl, d, S = 2, 32, 8, 128, 1024
    cfg = DecodeConfig(max_seq=S)
    cos, sin = _rope_tables(cfg, DEV)
    scale = 1.0 / math.sqrt(d)
    for p in (0, 1, 135, 255, 256, 257, 700, 1023):
        kc = torch.full((bs, kvl, S, d), float("nan"), device=DEV, dtype=dtype)
        vc = torch.full((bs, kvl, S, d), float("nan"), device=DEV, dtype=dtype)
        kc[:, :, :p] = torch.randn(bs, kvl, p, d, device=DEV, generator=gen).to(dtype)
        vc[:, :, :p] = torch.randn(bs, kvl, p, d, device=DEV, generator=gen).to(dtype)
        pos = torch.tensor([p], device=DEV)
        qkv = torch.randn(bs, (hl + 2 * kvl) * d, device=DEV, generator=gen).to(dtype)
        k1, v1, k2, v2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
        want = G.rope_attn(qkv, cos, sin, pos, k1, v1, hl, kvl, d, scale)
        got = G.rope_attn_online(qkv, cos, sin, pos, k2, v2, hl, kvl, d, scale)
        # the rows written: the torch formulation's bits (two rounded products, a rounded sum, one rounding to 16 bits)
        c, s_ = cos[p].view(1, 1, -1), sin[p].view(1, 1, -1)
        want_k = _rope(qkv[:, hl * d: (hl + kvl) * d].reshape(bs, kvl, d), c, s_)
        want_v = qkv[:, (hl + kvl) * d:].reshape(bs, kvl, d)
        assert torch.equal(k2[:, :, p], want_k) and torch.equal(v2[:, :, p], want_v), p
        assert torch.equal(k2[:, :, :p], kc[:, :, :p]) and torch.equal(v2[:, :, :p], vc[:, :, :p]), p
        assert torch.isnan(k2[:, :, p + 1:].float()).all() and torch.isfinite(got.float()).all(), p
        assert (got.float() - want.float()).abs().max() <= 4 * ulp * want.float().abs().max(), (p, (got.float() - want.float()).abs().max())
        # the same kernel split over the sequence (blocks take every NS-th 32-row iteration; the last block of a head to arrive combines):
        # blocks without rows, the new token's row in any block, twice through the same scratch (the counters reset themselves)
        for ns in (2, 5, 8):
            scr = G.rope_attn_split_scratch(bs, hl, d, ns, DEV)
            for _ in range(2):
                k3, v3 = kc.clone(), vc.clone()
                sp = G.rope_attn_split(qkv, cos, sin, pos, k3, v3, hl, kvl, d, scale, scr, ns)
                assert torch.equal(k3[:, :, :p + 1], k2[:, :, :p + 1]) and torch.equal(v3[:, :, :p + 1], v2[:, :, :p + 1]), (p, ns)
                assert torch.isnan(k3[:, :, p + 1:].float()).all() and torch.isfinite(sp.float()).all(), (p, ns)
                assert (sp.float() - want.float()).abs().max() <= 4 * ulp * want.float().abs().max(), (p, ns)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear16_gemv_vs_torch(dtype):
    """dg_linear16 (the decode step's un-quantised LM head: row-major 16-bit weights, 1 ... 4 rows) against an f64 product: one 16-bit
    rounding (half a spacing = 2^-8 |y| for bf16, 2^-11 |y| for fp16, at most) of a sum with f32 accumulation; ragged row counts (not a
    multiple of the waves), every k."""
    from any4_amd import decode_ops as G

    gen = torch.Generator(device=DEV).manual_seed(5)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for (m, n, k) in [(1, 128256, 4096), (3, 1000, 2048), (4, 32000, 4096), (2, 5003, 8192), (1, 7, 4096)]:
        x = torch.randn(m, k, device=DEV, generator=gen).to(dtype)
        w = (torch.randn(n, k, device=DEV, generator=gen) * 0.02).to(dtype)
        y = G.linear16(x, w)
        assert y is not None and y.shape == (m, n)
        want = x.double() @ w.double().t()
        assert ((y.double() - want).abs() <= ulp * want.abs() * 1.01 + 8e-6 * (x.double().abs() @ w.double().abs().t())).all(), (m, n, k)
    assert G.linear16(torch.randn(2, 1024, device=DEV).to(dtype), torch.randn(8, 1024, device=DEV).to(dtype)) is None   # no instantiation: the caller's GEMM


@pytest.mark.parametrize("ns", [4, 8])
def test_rope_attn_split_many_back_to_back_launches(ns):
    """The split launch's cross-block (cross-XCD) hand-over uses no device-scope fence: everything that crosses blocks is written
    AND read with agent-scope (sc1: write-through / L1-bypassing) accesses, every writing wave drains its stores (vmcnt(0)) before
    the workgroup barrier in front of the counter's atomic increment, and the counter resets itself -- the `sc1 stores and loads on
    both sides` form of the hardware guide's inter-workgroup rules.  Pinned the way that guide asks: many launches back to back
    with fresh inputs (a stale partial or an early combine would show as a result of another launch), nsplit 4 and 8 (blocks of
    one head on different XCDs), under UNEVEN load (a second stream streams 1 GiB copies meanwhile), every output checked."""
    from any4_amd import decode_ops as G
    from any4_amd.decode import DecodeConfig, _rope_tables

    gen = torch.Generator(device=DEV).manual_seed(3 + ns)
    bs, hl, kvl, d, S, p = 1, 32, 8, 128, 2048, 1500
    cos, sin = _rope_tables(DecodeConfig(max_seq=S), DEV)
    scale = 1.0 / math.sqrt(d)
    kc = torch.randn(bs, kvl, S, d, device=DEV, generator=gen).bfloat16()
    vc = torch.randn(bs, kvl, S, d, device=DEV, generator=gen).bfloat16()
    pos = torch.tensor([p], device=DEV)
    scr = G.rope_attn_split_scratch(bs, hl, d, ns, DEV)
    n_launch = 600
    qs = [torch.randn(bs, (hl + 2 * kvl) * d, device=DEV, generator=gen).bfloat16() for _ in range(n_launch)]
    load_a, load_b = torch.empty(1 << 28, dtype=torch.float32, device=DEV), torch.empty(1 << 28, dtype=torch.float32, device=DEV)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(12):
            load_b.copy_(load_a)
    outs = [G.rope_attn_split(q, cos, sin, pos, kc, vc, hl, kvl, d, scale, scr, ns) for q in qs]   # no sync in between
    torch.cuda.synchronize()
    for q, o in zip(qs, outs):
        want = G.rope_attn_online(q, cos, sin, pos, kc, vc, hl, kvl, d, scale)
        assert (o.float() - want.float()).abs().max() <= 2.0 ** -6 * want.float().abs().max()


@pytest.mark.parametrize("max_seq", [32, 4096])  # 4096: the split-sequence attention (8 blocks per head + combine)
def test_decode_graph_replay_equals_eager(max_seq):
    from any4_amd.decode import Any4Factory, DecodeConfig, DecodeStack

    cfg = DecodeConfig(**{**CFG, "max_seq": max_seq})
    eager = DecodeStack(cfg, Any4Factory(cfg, DEV, seed=3), DEV, bs=2, seed=9)
    graph = DecodeStack(cfg, Any4Factory(cfg, DEV, seed=3), DEV, bs=2, seed=9)
    graph.capture()
    assert graph._graph is not None
    toks = torch.randint(0, cfg.vocab, (5, 2), generator=torch.Generator().manual_seed(2)).to(DEV)
    # capture's warm-up steps wrote position 0 of the cache with token 0; decoding from position 0 overwrites it
    assert (eager._attn_split > 1) == (max_seq > 1024)
    for i, t in enumerate(toks):
        a, b = eager.decode(t, i), graph.decode(t, i).clone()
        if max_seq <= 2048:
            assert torch.equal(a, b), i
        else:  # the order in which the chunks of a head reach the combine step is not fixed: equal up to fp32 summation order
            assert (a.float() - b.float()).abs().max() <= 2e-2 * b.float().abs().max(), i
    if max_seq > 2048:  # and the split path agrees with the plain-torch formulation
        plain = DecodeStack(cfg, Any4Factory(cfg, DEV, seed=3), DEV, bs=2, seed=9, fused=False)
        for i, t in enumerate(toks):
            a, b = eager.decode(t, i).float(), plain.decode(t, i).float()
            assert (a - b).abs().max() <= 0.03 * b.abs().max() + 1e-3, i


# ---------------------------------------------------------------- quantizer -> modules on the HIP kernels (N1)

def test_anyq_and_intq_layer_real_kernels(oracle):
    from any4_amd import quantize as Q

    torch.manual_seed(4)
    lin = torch.nn.Linear(512, 256, bias=True, device=DEV, dtype=torch.bfloat16)
    x = torch.randn(5, 512, device=DEV).to(torch.bfloat16)
    y_dense = lin(x).float()
    for layer_fn, kw in ((Q.anyq_layer, dict(group_size=128)), (Q.anyq_layer, dict(group_size=64, per_row=False)),
                         (Q.intq_layer, dict(group_size=128))):
        import copy

        src = copy.deepcopy(lin)
        q = layer_fn(src, name="lin", **kw)
        assert type(q).__name__ in ("Any4Linear", "Int4Linear") and q.weight.dim() == 4 and q.weight_reshaped
        # the fake-quantized twin: same quantizer, weights reconstructed in place of the dense ones
        twin = layer_fn(copy.deepcopy(lin), name="lin", pseudo=True, **kw)
        y_q, y_t = q(x).float(), twin(x).float()
        assert (y_q - y_t).abs().max() <= 0.02 * y_t.abs().max() + 1e-2, (layer_fn.__name__, kw)
        assert (y_q - y_dense).abs().max() <= 0.25 * y_dense.abs().max()       # 4-bit, still the same layer
    # any4 is the better 4-bit grid on the same groups
    e_any = (Q.anyq_layer(copy.deepcopy(lin), pseudo=True, group_size=128).weight - lin.weight).float().pow(2).mean()
    e_int = (Q.intq_layer(copy.deepcopy(lin), pseudo=True, group_size=128, unsigned=True).weight - lin.weight).float().pow(2).mean()
    assert e_any < e_int


def test_kmeans_on_gpu_matches_cpu():
    from any4_amd import quantize as Q

    torch.manual_seed(6)
    w = torch.randn(64, 2048)
    wg = Q.group_q(w, 4, 128)[0]
    a_c, c_c = Q.kmeans_rows(wg, 16)
    a_g, c_g = Q.kmeans_rows(wg.to(DEV), 16)
    sse = lambda a, c: ((c.gather(1, a.long()) - wg.to(c.device)) ** 2).sum(1).cpu()
    assert torch.allclose(sse(a_g, c_g), sse(a_c, c_c), rtol=2e-3)
    codes, lut, sz = Q.anyq_quantize_tensor(w.to(torch.bfloat16), device=DEV)   # CPU checkpoint, clustered on the GPU
    assert codes.device.type == "cpu" and lut.device.type == "cpu" and lut.dtype == torch.bfloat16


def test_quantizer_on_the_gpu_against_the_reference_fixture(T, oracle):
    """N1 pinned on the GPU to the REFERENCE's output, not to itself: the W of the captured fixture (tests/golden/make_golden.py:
    torch.manual_seed(1234), randn(1024, 1024) * 0.02 -> bf16, quantized there by the imported reference's
    quantize.anyq_quantize_tensor, quantize.py:523-610) is quantized here by any4_amd.quantize.anyq_quantize_tensor ON cuda:0.
    Scales / zeros bit-equal to the reference's; reconstruction error within 1.02 x of the reference's codes + LUT (sklearn parity
    is statistical); the result, packed and multiplied by the HIP kernel, within north_star's 1e-2 of the reference's
    own captured y (max|y| = 2.2).  And the reference's exact-recovery property (test_anyq.py:31-49) on the GPU."""
    from any4_amd import quantize as Q
    from tests.conftest import bits16, from_bits16, load_golden

    d = load_golden("any4_n1024_k1024_g128_seed1234.npz")
    torch.manual_seed(1234)
    W = (torch.randn(1024, 1024) * 0.02).to(torch.bfloat16)
    c8 = d["codes_nib"]
    ref_codes = np.empty((1024, 1024), np.int32)
    ref_codes[:, 0::2], ref_codes[:, 1::2] = c8 & 15, c8 >> 4
    ref_lut, ref_sz = from_bits16(d["lut_bits"], torch.bfloat16), from_bits16(d["sz_bits"], torch.bfloat16)
    ref_deq = Q.anyq_dequantize_tensor(torch.from_numpy(ref_codes), ref_lut, ref_sz, n_bit=4, q_group_size=128, per_row=True)

    codes, lut, sz = Q.anyq_quantize_tensor(W.to(DEV), n_bit=4, q_group_size=128, per_row=True)
    assert codes.is_cuda and lut.is_cuda and sz.is_cuda and codes.dtype == torch.int32 and lut.dtype == torch.bfloat16
    assert int(codes.min()) >= 0 and int(codes.max()) <= 15 and lut.shape == (1024, 16) and sz.shape == (8, 1024, 2)
    nbad = int((bits16(sz.cpu()) != d["sz_bits"]).sum())
    assert nbad == 0, f"{nbad} of {sz.numel()} scale / zero values differ from the reference's bits"   # the reference's grouping
    mine = Q.anyq_dequantize_tensor(codes, lut, sz, n_bit=4, q_group_size=128, per_row=True).cpu()
    mse = lambda a: ((a.float() - W.float()) ** 2).mean().item()
    assert mse(mine) <= 1.02 * mse(ref_deq), (mse(mine), mse(ref_deq))
    # through the product path: pack, multiply on the HIP kernel (lut - 8 is what the module receives, quantize.py:893), compare
    # with the y the reference itself computed from ITS quantization of the same W
    x = from_bits16(d["x_bits"], torch.bfloat16)
    w2 = T.convert_matrix_to_m16n8k16_Bint4_layout(codes, 4)
    y = T.tinygemm_y_f16RM_x_f16RM_w_any4TC(x.to(DEV), w2, 128, sz, (lut.float() - 8).to(torch.bfloat16), True)
    y_ref = from_bits16(d["y_bits"], torch.bfloat16).float()
    assert float(y_ref.abs().max()) < 2.3
    # two different (equally good) codebooks of one W: the outputs differ by the quantization noise of either, which at this k
    # and |x| ~ 1 is a few 1e-2 -- bounded here by the noise the reference's own quantization has against the dense product
    dense = (x.float() @ W.float().t())
    noise_ref = (y_ref - dense).abs().max().item()
    assert (y.float().cpu() - dense).abs().max().item() <= 1.5 * noise_ref
    # the kernel itself on the GPU-made tensors: within 1e-2 of the CPU dequant-matmul of the same tensors
    y_cpu = (x.float() @ mine.float().t())
    assert (y.float().cpu() - y_cpu).abs().max().item() <= 1e-2 * max(1.0, float(y_cpu.abs().max()) / 2.2)

    # exact recovery of <= 16 distinct values per row (test_anyq.py:31-49), clustered on the GPU
    for per_row, g in ((True, 32), (False, 64)):
        torch.manual_seed(g)
        vals = torch.linspace(-8, 7, 16)
        w16 = vals[torch.stack([torch.randperm(16) for _ in range(64 * 64 // 16)]).view(64, 64)].to(DEV)
        c, l, s_ = Q.anyq_quantize_tensor(w16, n_bit=4, q_group_size=g, per_row=per_row)
        assert c.is_cuda and torch.equal(Q.anyq_dequantize_tensor(c, l, s_, q_group_size=g, per_row=per_row), w16)


def test_quantize_model_decode_stack():
    """quantize_model over a dense decode stack: every nn.Linear but the LM head becomes an Any4Linear and the
    logits stay close to the dense model's."""
    from any4_amd import quantize as Q
    from any4_amd.decode import DecodeConfig, DecodeStack, DenseFactory

    cfg = DecodeConfig(**CFG)
    stack = DecodeStack(cfg, DenseFactory(cfg, DEV, seed=2), DEV, bs=2, seed=4)
    dense = DecodeStack(cfg, DenseFactory(cfg, DEV, seed=2), DEV, bs=2, seed=4)
    Q.quantize_model(stack, layer_to=Q.anyq_layer, group_size=cfg.group_size)
    kinds = {type(m).__name__ for m in stack.modules()}
    assert "Any4Linear" in kinds and type(stack.lm_head).__name__ == "Linear"
    assert sum(type(m).__name__ == "Any4Linear" for m in stack.modules()) == 4 * cfg.layers
    toks = torch.randint(0, cfg.vocab, (4, 2), generator=torch.Generator().manual_seed(8)).to(DEV)
    for i, t in enumerate(toks):
        a, b = stack.decode(t, i).float(), dense.decode(t, i).float()
        assert torch.isfinite(a).all() and (a - b).abs().max() <= 0.35 * b.abs().max(), i


def test_hf_llama_quantize_model_real_vs_pseudo():
    """The reference's model-level path (benchmark.py / eval.py): a HuggingFace Llama, every nn.Linear but the LM
    head swapped by quantize_model.  Real kernels (Any4Linear on HIP) and fake quantization (reconstructed weights in
    nn.Linear) of the SAME deterministic quantizer must give the same logits up to 16-bit summation order."""
    import copy

    transformers = pytest.importorskip("transformers")
    from any4_amd import quantize as Q

    torch.manual_seed(0)
    cfg = transformers.LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=2, vocab_size=512, max_position_embeddings=64)
    base = transformers.AutoModelForCausalLM.from_config(cfg, dtype=torch.bfloat16).to(DEV).eval()
    real, fake = copy.deepcopy(base), copy.deepcopy(base)
    Q.quantize_model(real, layer_to=Q.anyq_layer, pseudo=False, group_size=64)
    Q.quantize_model(fake, layer_to=Q.anyq_layer, pseudo=True, group_size=64)
    assert sum(type(m).__name__ == "Any4Linear" for m in real.modules()) == 7 * cfg.num_hidden_layers
    assert type(real.lm_head).__name__ == "Linear"
    ids = torch.randint(0, cfg.vocab_size, (2, 5), generator=torch.Generator().manual_seed(1)).to(DEV)
    with torch.no_grad():
        yb = base(input_ids=ids, use_cache=False).logits.float()
        yr = real(input_ids=ids, use_cache=False).logits.float()
        yf = fake(input_ids=ids, use_cache=False).logits.float()
    assert torch.isfinite(yr).all()
    assert (yr - yf).abs().max() <= 0.03 * yf.abs().max() + 1e-3, ((yr - yf).abs().max(), yf.abs().max())
    assert (yf - yb).abs().max() > 0      # it is quantized...
    assert (yr - yb).abs().max() <= 0.5 * yb.abs().max()  # ...and still the same model


def test_nf4_and_mx4_modules():
    """NF4Linear / MX4Linear (the reference's modules.py:10 TODO) on the any4-global-LUT and mx4 kernels: real kernels
    vs the fake-quantized weights of the same quantizer."""
    import copy

    from any4_amd import quantize as Q

    torch.manual_seed(11)
    lin = torch.nn.Linear(512, 192, bias=True, device=DEV, dtype=torch.bfloat16)
    x = torch.randn(6, 512, device=DEV).to(torch.bfloat16)
    for layer_fn, kw, cls in ((Q.nf4_layer, dict(group_size=64), "NF4Linear"), (Q.mx4_layer, dict(group_size=32), "MX4Linear"),
                              (Q.mx4_layer, dict(group_size=32, kernel="linear_y_f16RM_W_mx4TC_x_f16RM"), "MX4Linear")):
        q = layer_fn(copy.deepcopy(lin), **kw)
        twin = layer_fn(copy.deepcopy(lin), pseudo=True, **kw)
        assert type(q).__name__ == cls and q.weight.dim() == 4
        yq, yt = q(x).float(), twin(x).float()
        assert (yq - yt).abs().max() <= 0.02 * yt.abs().max() + 1e-2, (cls, kw)
        assert (yq - lin(x).float()).abs().max() <= 0.3 * yt.abs().max()
    # NF4 codes are the nearest code-book entries of w / absmax(group)
    from any4_amd.modules import NF4_VALUES

    codes, book, sz = Q.nf4_quantize_tensor(lin.weight, 64)
    g = lin.weight.float().reshape(-1, 64)
    scaled = g / g.abs().amax(1, keepdim=True)
    ref = (scaled[..., None] - torch.tensor(NF4_VALUES, device=DEV)).abs().argmin(-1).reshape(lin.weight.shape)
    assert (codes.long() != ref).float().mean() < 1e-4   # exact ties aside
    assert book.dtype == torch.bfloat16 and sz.shape == (512 // 64, 192, 2) and bool((sz[..., 1] == 0).all())


def test_accuracy_loop_with_real_kernels():
    """SURVEY 8f N4 on the GPU: calibration -> any4 quantization (Any4Linear on the HIP library, activation-aware) -> perplexity;
    the quantized model's perplexity on the synthetic stream stays close to the 16-bit model's, and the hook profiler sees every
    attention / MLP block."""
    import math

    from transformers import AutoModelForCausalLM, LlamaConfig

    from any4_amd import accuracy as A
    from any4_amd import quantize as Q

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      vocab_size=512, max_position_embeddings=128)
    model = AutoModelForCausalLM.from_config(cfg, dtype=torch.bfloat16).to(DEV).eval()
    toks = A.synthetic_corpus(512, 64 * 24, seed=7)
    ppl0 = A.perplexity(model, toks, seqlen=64)
    sw = A.calibrate(model, A.windows(toks, 64, 4))
    Q.quantize_model(model, layer_from=torch.nn.Linear, layer_to=Q.anyq_layer, skip_modules=["lm_head"], pseudo=False, group_size=64,
                     sample_weight=sw)
    assert sum(type(m).__name__ == "Any4Linear" for m in model.modules()) == 2 * 7
    ppl1 = A.perplexity(model, toks, seqlen=64)
    assert math.isfinite(ppl1) and abs(math.log(ppl1 / ppl0)) < 0.25, (ppl0, ppl1)
    ids = torch.randint(0, 512, (1, 1), device=DEV)
    prof = A.HookProfiler("cuda")
    prof.run_profiling(model, lambda m: m(input_ids=ids, use_cache=False), warmup=1, iters=2)
    s = prof.summarize()
    assert len(prof.timings) == 4 and s["attention_time"] > 0 and s["mlp_time"] > 0

