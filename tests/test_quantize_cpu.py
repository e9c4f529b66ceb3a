"""CPU suite, part 4: the quantizer (any4_amd/quantize.py) against the reference's own outputs
captured in tests/golden/ (group_q scales bit-exact, anyq_dequantize_tensor bit-exact, k-means by reconstruction
error) and the reference's exact-recovery property (test_anyq.py:31-49)."""
import hashlib

import numpy as np
import pytest
import torch

from any4_amd import quantize as Q
from tests.conftest import bits16, from_bits16, load_golden


@pytest.fixture(scope="module")
def fixture_any4():
    d = load_golden("any4_n1024_k1024_g128_seed1234.npz")
    torch.manual_seed(1234)  # tests/golden/make_golden.py: the W the reference quantized
    W = (torch.randn(1024, 1024) * 0.02).to(torch.bfloat16)
    c8 = d["codes_nib"]
    codes = np.empty((1024, 1024), np.int32)
    codes[:, 0::2], codes[:, 1::2] = c8 & 15, c8 >> 4
    return dict(W=W, codes=torch.from_numpy(codes), lut=from_bits16(d["lut_bits"], torch.bfloat16),
                sz=from_bits16(d["sz_bits"], torch.bfloat16), d=d)


def test_group_q_scales_bit_exact_vs_reference(fixture_any4):
    f = fixture_any4
    wg, wz, sz = Q.group_q(f["W"], 4, q_group_size=128)
    assert np.array_equal(bits16(sz.to(torch.bfloat16)), f["d"]["sz_bits"])
    assert wg.min() >= 0 and wg.max() <= 15 + 1e-4 and wg.shape == f["W"].shape
    s, z = Q.extract_scales_and_zeros(sz, wg.shape, 128)
    assert torch.allclose(Q.degroup_q(wg, scales=s, zeros=z), f["W"].float(), atol=1e-6)


def test_anyq_dequantize_bit_exact_vs_reference(fixture_any4):
    f = fixture_any4
    wdeq = Q.anyq_dequantize_tensor(f["codes"], f["lut"], f["sz"], n_bit=4, q_group_size=128, per_row=True)
    assert wdeq.dtype == torch.bfloat16
    assert np.array_equal(bits16(wdeq[:8]), f["d"]["wdeq_rows0_8_bits"])
    assert hashlib.sha256(bits16(wdeq).tobytes()).digest() == f["d"]["wdeq_sha256"].tobytes()


def test_anyq_quantize_matches_sklearn_reconstruction_error(fixture_any4):
    f = fixture_any4
    W = f["W"]
    ref = Q.anyq_dequantize_tensor(f["codes"], f["lut"], f["sz"])
    codes, lut, sz = Q.anyq_quantize_tensor(W, n_bit=4, q_group_size=128, per_row=True)
    assert codes.dtype == torch.int32 and codes.shape == W.shape and int(codes.min()) >= 0 and int(codes.max()) <= 15
    assert lut.dtype == W.dtype and lut.shape == (1024, 16) and sz.dtype == W.dtype and sz.shape == (8, 1024, 2)
    assert np.array_equal(bits16(sz), f["d"]["sz_bits"])           # same grouping as the reference
    mine = Q.anyq_dequantize_tensor(codes, lut, sz)
    mse = lambda a: ((a.float() - W.float()) ** 2).mean().item()
    assert mse(mine) <= 1.02 * mse(ref), (mse(mine), mse(ref))     # sklearn parity is statistical
    # and clearly better than the uniform int4 grid on the same groups
    assert mse(mine) < 0.8 * mse(Q.intq_reconstruct_tensor(W, unsigned=True, dtype=torch.float32))


@pytest.mark.parametrize("per_row", [True, False])
@pytest.mark.parametrize("g", [32, 64])
def test_anyq_exact_recovery_of_16_values(per_row, g):
    """test_anyq.py:31-49: a tensor whose rows are permutations of 16 values survives quantize -> dequantize."""
    torch.manual_seed(g)
    vals = torch.linspace(-8, 7, 16)
    w = vals[torch.stack([torch.randperm(16) for _ in range(64 * 64 // 16)]).view(64, 64)]
    codes, lut, sz = Q.anyq_quantize_tensor(w, n_bit=4, q_group_size=g, per_row=per_row)
    assert lut.shape == ((64, 16) if per_row else (16,))
    assert torch.equal(Q.anyq_dequantize_tensor(codes, lut, sz, q_group_size=g, per_row=per_row), w)


def test_kmeans_rows_properties():
    torch.manual_seed(0)
    x = torch.randn(6, 300)
    a, c = Q.kmeans_rows(x, 16)
    a0, c0 = a, c
    assert a.shape == x.shape and c.shape == (6, 16) and bool((c[:, 1:] >= c[:, :-1]).all())
    # every point sits in its nearest cluster; every centre is the mean of its cluster (Lloyd fixed point)
    assert torch.equal(a.long(), (x[:, :, None] - c[:, None, :]).abs().argmin(-1))
    for r in range(6):
        for j in range(16):
            sel = x[r][a[r] == j]
            assert sel.numel() > 0 and abs(sel.mean().item() - c[r, j].item()) < 1e-4
    # fewer distinct values than clusters: reproduced exactly
    y = torch.randn(4, 5).repeat(1, 40)
    a, c = Q.kmeans_rows(y, 16)
    assert torch.allclose(c.gather(1, a.long()), y, atol=1e-6)
    # sample weights pull the centres towards the heavy points
    z = torch.cat([torch.zeros(1, 50), torch.ones(1, 50)], 1)
    w = torch.cat([torch.full((50,), 9.0), torch.ones(50)])
    _, c1 = Q.kmeans_rows(z, 1, sample_weight=w)
    assert abs(c1.item() - 0.1) < 1e-6
    with pytest.raises(ValueError):
        Q.kmeans_rows(x, 16, init="nope")
    # the reference's defaults (init=None -> scikit-learn's "k-means++", quantize.py:413) select the built-in seedings
    for init in (None, "k-means++", "random"):
        a2, c2 = Q.kmeans_rows(x, 16, init=init)
        assert torch.equal(a2, a0) and torch.equal(c2, c0)


def test_intq_matches_tinygemm_grid():
    from tinygemm_lib.utils import group_quantize_tensor

    torch.manual_seed(1)
    w = torch.randn(32, 256).to(torch.bfloat16)
    codes, _, sz = Q.intq_quantize_tensor(w, q_group_size=64, new_grouping="tinygemm")
    c2, sz2 = group_quantize_tensor(w, 4, 64)
    assert torch.equal(codes, c2) and torch.equal(sz, sz2)
    rec = Q.intq_reconstruct_tensor(w, q_group_size=64, unsigned=True, dtype=torch.float32)
    s, _ = Q.extract_scales_and_zeros(Q.group_q(w, 4, 64)[2], w.shape, 64)
    # half a grid step, plus the bf16 rounding of scale and zero (they are returned in w's dtype)
    assert ((rec - w.float()).abs() <= s * 0.5 + 2.0 ** -8 * (15 * s + w.float().abs().max())).all()


def test_quantize_model_pseudo_swaps_weights_and_skips_lm_head():
    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = torch.nn.Sequential(torch.nn.Linear(128, 64), torch.nn.ReLU(), torch.nn.Linear(64, 128, bias=False))
            self.lm_head = torch.nn.Linear(128, 10, bias=False)

        def forward(self, x):
            return self.lm_head(self.body(x))

    torch.manual_seed(2)
    m = Tiny()
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    x = torch.randn(3, 128)
    y0 = m(x)
    Q.quantize_model(m, layer_to=Q.anyq_layer, pseudo=True, group_size=64)
    after = dict(m.named_parameters())
    assert torch.equal(after["lm_head.weight"], before["lm_head.weight"])          # skipped by default
    assert torch.equal(after["body.0.bias"], before["body.0.bias"])
    for n in ("body.0.weight", "body.2.weight"):
        err = (after[n] - before[n]).abs().max().item()
        assert 0 < err < 0.05, (n, err)                                               # fake-quantized, close
        assert all(torch.unique(r).numel() <= 16 * (r.numel() // 64) for r in after[n])  # <= 16 levels per group
    assert (m(x) - y0).abs().max() < 0.1
    # skip_modules as a comma-separated string of names (quantize.py:37-38); uniform int4 fake quantization
    m2 = Tiny()
    w0, w2 = m2.body[0].weight.detach().clone(), m2.body[2].weight.detach().clone()
    Q.quantize_model(m2, layer_to=Q.intq_layer, pseudo=True, group_size=64, skip_modules="body.0,lm_head", unsigned=True)
    assert torch.equal(m2.body[0].weight, w0) and not torch.equal(m2.body[2].weight, w2)
    assert (m2.body[2].weight - w2).abs().max() < 0.05


def test_kmeans_rows_sklearn_seedings_are_aliases_and_say_so():
    """None / 'k-means++' / 'random' (what the reference forwards to scikit-learn, quantize.py:413) run the
    deterministic seedings here -- same result as the default, announced once; an unknown name is rejected."""
    import warnings

    Q._WARNED_INIT.discard("random")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 128, generator=g)
    a0, c0 = Q.kmeans_rows(x, 16)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        a1, c1 = Q.kmeans_rows(x, 16, init="random")
        a2, c2 = Q.kmeans_rows(x, 16, init="random")
    assert sum("scikit-learn" in str(w.message) for w in rec) == 1
    assert torch.equal(a0, a1) and torch.equal(c0, c1) and torch.equal(a1, a2)
    with pytest.raises(ValueError):
        Q.kmeans_rows(x, 16, init="no-such-seeding")
