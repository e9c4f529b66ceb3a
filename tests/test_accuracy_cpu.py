"""CPU suite: the accuracy loop (any4_amd/accuracy.py: calibration hooks, GPTQ-style perplexity, attention/MLP profiler)
against closed-form answers and a tiny HF Llama.  Mirrors what calibrate.py:41-73, data_gptq.py:196-220 and
benchmark.py:37-111 of the reference compute."""
import math

import numpy as np
import pytest
import torch

from any4_amd import accuracy as A


class UniformLM(torch.nn.Module):
    """Predicts the same logits for every position."""

    def __init__(self, logits):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor(logits, dtype=torch.float32))

    def forward(self, ids):
        return self.w[None, None, :].expand(ids.shape[0], ids.shape[1], -1)


def test_perplexity_of_uniform_model_is_vocab_size():
    V, seqlen = 37, 16
    toks = A.synthetic_corpus(V, 5 * seqlen + 3, seed=1)
    ppl = A.perplexity(UniformLM([0.0] * V), toks, seqlen=seqlen)
    assert ppl == pytest.approx(V, rel=1e-6)


def test_perplexity_matches_direct_formula():
    V, seqlen = 11, 8
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(V, generator=g).tolist()
    toks = A.synthetic_corpus(V, 4 * seqlen, seed=2)
    ppl = A.perplexity(UniformLM(logits), toks, seqlen=seqlen)
    logp = torch.log_softmax(torch.tensor(logits, dtype=torch.float64), 0)
    # per window: mean NLL of its seqlen - 1 predicted tokens, weighted by seqlen (data_gptq.py:211-216)
    total = 0.0
    for w in A.windows(toks, seqlen):
        total += float(-logp[w[0, 1:]].mean()) * seqlen
    assert ppl == pytest.approx(math.exp(total / (4 * seqlen)), rel=1e-6)


def test_perplexity_rejects_short_stream():
    with pytest.raises(ValueError):
        A.perplexity(UniformLM([0.0, 0.0]), torch.zeros(1, 3, dtype=torch.int64), seqlen=8)


def test_synthetic_corpus_is_deterministic_and_structured():
    a, b = A.synthetic_corpus(100, 2000, seed=3), A.synthetic_corpus(100, 2000, seed=3)
    assert torch.equal(a, b) and a.shape == (1, 2000) and int(a.max()) < 100 and int(a.min()) >= 0
    # first-order structure: far fewer distinct bigrams than an i.i.d. stream of the same unigram law would give
    bigrams = len({(int(x), int(y)) for x, y in zip(a[0, :-1], a[0, 1:])})
    perm = a[0][torch.randperm(2000, generator=torch.Generator().manual_seed(0))]
    bigrams_iid = len({(int(x), int(y)) for x, y in zip(perm[:-1], perm[1:])})
    assert bigrams < 0.8 * bigrams_iid


def test_activation_stats_mean_over_all_but_last_dim():
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    xs = [torch.randn(2, 7, 6), torch.randn(3, 7, 6)]
    stats = A.ActivationStats(abs=True, keep_activations=True).register(model)
    for x in xs:
        model(x)
    stats.remove()
    m = stats.mean()
    assert set(m) == {"0", "2"}
    want0 = torch.cat([x.reshape(-1, 6) for x in xs]).double().abs().mean(0)
    assert torch.allclose(m["0"], want0, atol=1e-12) and m["0"].dtype == torch.float64
    h = torch.cat([torch.relu(model[0](x)).reshape(-1, 5) for x in xs]).double()
    assert torch.allclose(m["2"], h.abs().mean(0), atol=1e-6)
    assert len(stats.lists["0"]) == 2
    # hooks are gone
    model(xs[0])
    assert stats.counts["0"] == 5 * 7


def test_activation_stats_signed_and_filter():
    model = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 2))
    x = torch.randn(9, 4)
    stats = A.ActivationStats(abs=False, layer_filter=["0"]).register(model)
    model(x)
    stats.remove()
    assert set(stats.mean()) == {"0"}
    assert torch.allclose(stats.mean()["0"], x.double().mean(0), atol=1e-12)


def _tiny_llama():
    from transformers import AutoModelForCausalLM, LlamaConfig

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      vocab_size=97, max_position_embeddings=64)
    return AutoModelForCausalLM.from_config(cfg).eval()


def test_calibrate_then_quantize_then_perplexity_on_a_tiny_llama():
    """The whole loop on CPU with the pseudo (fake-quant) path: calibration feeds the any4 quantizer's sample weights, and
    perplexity moves by a bounded amount."""
    from any4_amd import quantize as Q

    model = _tiny_llama()
    toks = A.synthetic_corpus(97, 16 * 32, seed=5)
    batches = A.windows(toks, 32, 4)
    sw = A.calibrate(model, batches)
    lin = [n for n, m in model.named_modules() if isinstance(m, torch.nn.Linear)]
    assert set(sw) == set(lin)
    for n, m in model.named_modules():
        if isinstance(m, torch.nn.Linear):
            assert sw[n].shape == (m.in_features,) and bool((sw[n] >= 0).all())
    ppl0 = A.perplexity(model, toks, seqlen=32, nsamples=8)
    assert 1.0 < ppl0 < 200.0
    Q.quantize_model(model, layer_from=torch.nn.Linear, layer_to=Q.anyq_layer, skip_modules=["lm_head"], pseudo=True, group_size=32,
                     sample_weight=sw)
    ppl1 = A.perplexity(model, toks, seqlen=32, nsamples=8)
    assert math.isfinite(ppl1) and abs(math.log(ppl1) - math.log(ppl0)) < 0.5


def test_hook_profiler_splits_attention_and_mlp():
    model = _tiny_llama()
    ids = torch.randint(0, 97, (1, 8))
    prof = A.HookProfiler("cpu")
    prof.run_profiling(model, lambda m: m(input_ids=ids), warmup=1, iters=3)
    assert set(prof.timings) == {"attention_layer_0", "attention_layer_1", "mlp_layer_0", "mlp_layer_1"}
    assert all(len(v) == 3 for v in prof.timings.values())
    s = prof.summarize()
    assert s["attention_time"] > 0 and s["mlp_time"] > 0 and s["ratio"] == pytest.approx(s["attention_time"] / s["mlp_time"])
    assert not prof._handles
    with pytest.raises(ValueError):
        A.HookProfiler("gpu")


# ---- pinned to the reference: data_gptq.llama_eval and calibrate.calibrate run by tests/golden/make_golden_accuracy.py ----

def _golden_tiny_llama():
    import numpy as np

    from tests.conftest import load_golden
    from tests.golden.make_golden_accuracy import build_model

    f = load_golden("accuracy_tiny_llama.npz")
    state = {k[len("state/"):]: np.asarray(f[k]) for k in f.files if k.startswith("state/")}
    return f, build_model(state)


def test_perplexity_matches_the_reference_llama_eval():
    """any4_amd.accuracy.perplexity on the fixture's model and token stream (ragged tail included) == the value the
    reference's data_gptq.llama_eval (data_gptq.py:196-220) returned for them."""
    import torch

    from any4_amd.accuracy import perplexity

    f, model = _golden_tiny_llama()
    ppl = perplexity(model, torch.from_numpy(f["tokens"]), seqlen=int(f["seqlen"]))
    assert abs(ppl - float(f["ppl"])) <= 1e-4 * float(f["ppl"]), (ppl, float(f["ppl"]))


def test_calibration_means_match_the_reference_hooks():
    """any4_amd.accuracy.calibrate on the fixture's model and calibration tokens == the per-layer mean input activations the
    reference's calibrate.calibrate (calibrate.py:41-73, 74-183) collected, for abs = False and abs = True, layer for layer."""
    import numpy as np
    import torch

    from any4_amd.accuracy import calibrate

    f, model = _golden_tiny_llama()
    calib = torch.from_numpy(f["calib_tokens"])
    for tag, use_abs in (("raw/", False), ("abs/", True)):
        got = calibrate(model, [calib], abs=use_abs)
        want = {k[len("mean/" + tag):]: np.asarray(f[k]) for k in f.files if k.startswith("mean/" + tag)}
        assert set(got) == set(want) and len(want) == 15  # 7 linears x 2 layers + lm_head
        for name, w in want.items():
            g = got[name].numpy()
            assert g.dtype == np.float64 and g.shape == w.shape
            assert np.allclose(g, w, rtol=1e-9, atol=1e-12), name
