"""GPU suite (-m gpu), default numerics: the model-level rows of SURVEY 8(f) at full size -- the accuracy loop's reference fixture
through the HIP path (N4) and BASELINE config 5's whole 32-layer decode stack against its dequantised dense twin (N2)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_accuracy_fixture_through_the_hip_path():
    """SURVEY 8f row N4 on the GPU: the tiny Llama of tests/golden/accuracy_tiny_llama.npz (values captured from the reference's
    data_gptq.llama_eval / calibrate.calibrate by tests/golden/make_golden_accuracy.py) through any4_amd.accuracy on cuda:0 --
    the fp32 model reproduces the reference's perplexity and calibration means there too; then its linears are quantized to the
    real kernels (any4, no pseudo path) and the perplexity of the quantized model stays within a stated factor of the reference's
    number and equals the fake-quantized twin's within kernel rounding."""
    import copy

    import numpy as np

    from any4_amd import accuracy as A
    from any4_amd import quantize as Q
    from tests.conftest import load_golden
    from tests.golden.make_golden_accuracy import build_model

    f = load_golden("accuracy_tiny_llama.npz")
    state = {k[len("state/"):]: np.asarray(f[k]) for k in f.files if k.startswith("state/")}
    model = build_model(state).to(DEV)
    toks, seqlen = torch.from_numpy(f["tokens"]), int(f["seqlen"])
    ppl = A.perplexity(model, toks.to(DEV), seqlen=seqlen)
    assert abs(ppl - float(f["ppl"])) <= 2e-3 * float(f["ppl"]), (ppl, float(f["ppl"]))  # (GPU fp32 matmuls: another summation order)
    calib = torch.from_numpy(f["calib_tokens"]).to(DEV)
    got = A.calibrate(model, [calib], abs=True)
    want = {k[len("mean/abs/"):]: np.asarray(f[k]) for k in f.files if k.startswith("mean/abs/")}
    assert set(got) == set(want)
    for name, w in want.items():
        assert np.allclose(got[name].cpu().numpy(), w, rtol=2e-3, atol=1e-6), name
    # the quantized model on the real kernels
    m16 = copy.deepcopy(model).to(torch.bfloat16)
    real, fake = copy.deepcopy(m16), copy.deepcopy(m16)
    Q.quantize_model(real, layer_to=Q.anyq_layer, pseudo=False, group_size=64)
    Q.quantize_model(fake, layer_to=Q.anyq_layer, pseudo=True, group_size=64)
    assert any(type(mod).__name__ == "Any4Linear" for mod in real.modules())
    p_real = A.perplexity(real, toks.to(DEV), seqlen=seqlen)
    p_fake = A.perplexity(fake, toks.to(DEV), seqlen=seqlen)
    p_16 = A.perplexity(m16, toks.to(DEV), seqlen=seqlen)
    assert abs(math.log(p_real / p_fake)) < 0.02, (p_real, p_fake)        # real kernels == the fake-quantized weights
    assert abs(math.log(p_16 / float(f["ppl"]))) < 0.05, (p_16, float(f["ppl"]))
    assert abs(math.log(p_real / float(f["ppl"]))) < 0.25, (p_real, float(f["ppl"]))  # 4-bit weights of a random tiny model


def test_full_llama3_8b_stack_logits_against_the_dequantised_dense_stack():
    """BASELINE config 5 at its FULL shape: the 32-layer Llama-3-8B-shaped any4 stack that bench.py times (five launches per layer,
    one hipGraph) against the same stack with every linear replaced by a bf16 nn.Linear holding the dequantised weights and the
    plain-torch formulation of everything else -- several tokens, logits compared, the graph replay bit-equal to eager."""
    from any4_amd import ops
    from any4_amd.decode import Any4Factory, DecodeConfig, DecodeStack

    cfg = DecodeConfig.llama3_8b(max_seq=64, gate_up_interleave=8)
    cfg.vocab = 4096  # (the LM head is a plain 16-bit linear in both stacks: its 1 GB adds nothing to this comparison)
    fac = Any4Factory(cfg, DEV, torch.bfloat16, seed=3)
    made = {}

    def any4(name, layer, k, rows):
        made[(name, layer)] = fac(name, layer, k, rows)
        return made[(name, layer)]

    def dense(name, layer, k, rows):
        mod = made[(name, layer)]
        codes = ops.unpack_int4(mod.weight.data, rows, k, "B")
        lut = mod.lut.data.float()
        sz = mod.scales_and_zeros.data.float()  # [k / g][rows][2]
        g = cfg.group_size
        w = torch.gather(lut, 1, codes.long())  # [rows][k]
        w = w.view(rows, k // g, g) * sz[:, :, 0].t().unsqueeze(-1) + sz[:, :, 1].t().unsqueeze(-1)
        lin = torch.nn.Linear(k, rows, bias=False, device=DEV, dtype=torch.bfloat16)
        lin.weight.data = w.view(rows, k).to(torch.bfloat16)
        return lin

    q = DecodeStack(cfg, any4, DEV, torch.bfloat16, bs=1, seed=5)
    d = DecodeStack(cfg, dense, DEV, torch.bfloat16, bs=1, seed=5, fused=False)
    toks = torch.randint(0, cfg.vocab, (4, 1), generator=torch.Generator().manual_seed(2)).to(DEV)
    eager = []
    for i, t in enumerate(toks):
        a, b = q.decode(t, i).float(), d.decode(t, i).float()
        assert torch.isfinite(a).all()
        # 32 layers of bf16 activations, 4-bit weights rounded once more to bf16 in the dense twin: a few percent of the logit scale
        assert (a - b).abs().max() <= 0.06 * b.abs().max() + 1e-3, (i, (a - b).abs().max(), b.abs().max())
        eager.append(a.clone())
    assert q.layers[0].launches() == 5
    q2 = DecodeStack(cfg, lambda n, l, k, r: made[(n, l)], DEV, torch.bfloat16, bs=1, seed=5)
    q2.capture()
    assert q2.kernels_per_layer == 5 and q2.graph_nodes == 32 * 5 + 3
    for i, t in enumerate(toks):
        assert torch.equal(q2.decode(t, i).float(), eager[i]), i
