"""Accuracy loop around the quantizer: activation calibration -> (activation-aware) quantization ->
perplexity, plus the attention / MLP hook profiler of the model benchmark (row H-B).

What it mirrors (behaviour, not code):
  * calibrate.py:41-73   forward hooks on every `layer_type` module that accumulate the mean (optionally of |x|) of the module's
                         INPUT over all dimensions but the last, in float64 on the host; the result is the per-layer
                         `sample_weight` dict the any4 quantizer takes (quantize.py:483-489 -> any4_amd.quantize.anyq_layer).
  * data_gptq.py:196-220 GPTQ-style perplexity: the token stream is cut into nsamples = numel // seqlen windows, the model runs
                         one window at a time, loss = CE(logits[:, :-1], tokens[:, 1:]) * seqlen, ppl = exp(sum / (n * seqlen)).
  * benchmark.py:37-111  HookBasedProfiler: pre/post forward hooks on each decoder layer's attention and MLP blocks, wall-clock
                         ("cpu") or event ("cuda") time per block, summarised as attention_time / mlp_time / ratio.

There is no network here, so no wikitext / c4: `synthetic_corpus` draws a Zipf-distributed token stream with local
structure (a first-order Markov chain) so that a model's perplexity on it is well below the vocabulary size and moves when the
weights are perturbed; `load_tokens` reads a local .npy / .pt / whitespace-separated token file for real data.
"""
from __future__ import annotations

import time
from collections import defaultdict
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------------------------
# calibration (calibrate.py:41-73)
# ------------------------------------------------------------------------------------------------------------------

class ActivationStats:
    """Mean input activation per layer.  `register(model)` hooks every `layer_type` module; run the calibration batches;
    `mean()` returns {module name: float64 [in_features]} and `remove()` drops the hooks."""

    def __init__(self, layer_type=torch.nn.Linear, abs: bool = True, layer_filter: Optional[Iterable[str]] = None,
                 keep_activations: bool = False):
        self.layer_type = layer_type
        self.abs = abs
        self.layer_filter = None if layer_filter is None else set(layer_filter)
        self.keep = keep_activations
        self.sums: Dict[str, torch.Tensor] = {}
        self.counts: Dict[str, int] = {}
        self.lists: Dict[str, List[torch.Tensor]] = {}
        self._handles = []

    def _hook(self, name):
        def hook(module, inputs, output):
            if self.layer_filter is not None and name not in self.layer_filter:
                return
            x = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
            x = x.detach().to("cpu", torch.float64)  # double on the host: sums over long calibration sets do not overflow
            if self.abs:
                x = x.abs()
            lead = list(range(x.dim() - 1))
            s = x.sum(dim=lead) if lead else x
            n = int(np.prod(x.shape[:-1])) if lead else 1
            if name in self.sums:
                self.sums[name] += s
                self.counts[name] += n
            else:
                self.sums[name] = s
                self.counts[name] = n
            if self.keep:
                self.lists.setdefault(name, []).append(x)
        return hook

    def register(self, model: torch.nn.Module) -> "ActivationStats":
        for name, module in model.named_modules():
            if isinstance(module, self.layer_type):
                self._handles.append(module.register_forward_hook(self._hook(name)))
        return self

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []

    def mean(self) -> Dict[str, torch.Tensor]:
        return {k: self.sums[k] / self.counts[k] for k in self.sums}


@torch.no_grad()
def calibrate(model: torch.nn.Module, batches: Iterable[torch.Tensor], layer_type=torch.nn.Linear, abs: bool = True,
              layer_filter: Optional[Iterable[str]] = None, return_activations: bool = False):
    """Runs `batches` (token id tensors [b][t]) through `model` with ActivationStats hooks and returns the per-layer mean
    activations (and the raw activation lists when asked): the `sample_weight` argument of quantize_model / anyq_layer."""
    stats = ActivationStats(layer_type, abs, layer_filter, return_activations).register(model)
    was_training = model.training
    model.eval()
    try:
        dev = next(model.parameters()).device
        for ids in batches:
            model(ids.to(dev))
    finally:
        stats.remove()
        model.train(was_training)
    return (stats.mean(), stats.lists) if return_activations else stats.mean()


# ------------------------------------------------------------------------------------------------------------------
# data
# ------------------------------------------------------------------------------------------------------------------

def synthetic_corpus(vocab_size: int, n_tokens: int, seed: int = 0, branching: int = 8) -> torch.Tensor:
    """A [1][n_tokens] token stream: every token has `branching` likely successors (drawn once per token from a Zipf law over
    the vocabulary) taken with probability 0.9, otherwise a fresh Zipf draw.  Deterministic in `seed`."""
    rng = np.random.default_rng(seed)
    ranks = np.arange(1, vocab_size + 1, dtype=np.float64)
    zipf = 1.0 / ranks
    zipf /= zipf.sum()
    succ = rng.choice(vocab_size, size=(vocab_size, branching), p=zipf)
    fresh = rng.choice(vocab_size, size=n_tokens, p=zipf)
    pick = rng.integers(0, branching, size=n_tokens)
    follow = rng.random(n_tokens) < 0.9
    out = np.empty(n_tokens, dtype=np.int64)
    out[0] = fresh[0]
    for i in range(1, n_tokens):
        out[i] = succ[out[i - 1], pick[i]] if follow[i] else fresh[i]
    return torch.from_numpy(out)[None, :]


def load_tokens(path: str) -> torch.Tensor:
    """Token ids from a local file: .npy, .pt (a 1-D / [1][n] integer tensor) or text with whitespace-separated integers."""
    if path.endswith(".npy"):
        t = torch.from_numpy(np.load(path).astype(np.int64))
    elif path.endswith(".pt"):
        t = torch.load(path).to(torch.int64)
    else:
        with open(path) as f:
            t = torch.tensor([int(v) for v in f.read().split()], dtype=torch.int64)
    return t.reshape(1, -1)


def windows(tokens: torch.Tensor, seqlen: int, nsamples: Optional[int] = None) -> List[torch.Tensor]:
    n = tokens.numel() // seqlen
    if nsamples is not None:
        n = min(n, nsamples)
    return [tokens[:, i * seqlen:(i + 1) * seqlen] for i in range(n)]


# ------------------------------------------------------------------------------------------------------------------
# perplexity (data_gptq.py:196-220)
# ------------------------------------------------------------------------------------------------------------------

@torch.no_grad()
def perplexity(model: torch.nn.Module, tokens: torch.Tensor, seqlen: int = 2048, nsamples: Optional[int] = None) -> float:
    """exp(mean token NLL) over the non-overlapping `seqlen` windows of `tokens` ([1][n])."""
    dev = next(model.parameters()).device
    was_training = model.training
    model.eval()
    nlls = []
    wins = windows(tokens, seqlen, nsamples)
    if not wins:
        raise ValueError("token stream shorter than one window")
    try:
        for w in wins:
            w = w.to(dev)
            out = model(w)
            logits = out.logits if hasattr(out, "logits") else out
            shift_logits = logits[:, :-1, :].float()
            shift_labels = w[:, 1:]
            loss = torch.nn.functional.cross_entropy(shift_logits.reshape(-1, shift_logits.size(-1)), shift_labels.reshape(-1))
            nlls.append(loss.double() * seqlen)  # (the reference weighs the mean of seqlen - 1 losses by seqlen)
    finally:
        model.train(was_training)
    return float(torch.exp(torch.stack(nlls).sum() / (len(wins) * seqlen)))


# ------------------------------------------------------------------------------------------------------------------
# attention / MLP profiler (benchmark.py:37-111)
# ------------------------------------------------------------------------------------------------------------------

def decoder_layers(model: torch.nn.Module) -> Sequence[torch.nn.Module]:
    """The list of decoder blocks of a HF-style causal LM (model.model.layers, transformer.h, ...)."""
    for path in ("model.layers", "model.decoder.layers", "transformer.h", "gpt_neox.layers", "layers"):
        obj = model
        ok = True
        for part in path.split("."):
            if not hasattr(obj, part):
                ok = False
                break
            obj = getattr(obj, part)
        if ok and isinstance(obj, (torch.nn.ModuleList, list, tuple)) and len(obj):
            return obj
    raise ValueError("could not find the decoder layers of this model")


def _children_matching(layer: torch.nn.Module, words) -> List[torch.nn.Module]:
    # (post_attention_layernorm and friends are not blocks of their own)
    return [m for n, m in layer.named_children() if any(w in n.lower() for w in words) and "norm" not in n.lower() and "ln" not in n.lower().split("_")]


class HookProfiler:
    """Per-block forward time of every decoder layer's attention and MLP modules.

    mode "cpu": perf_counter around the block (host time, includes launch overhead);
    mode "cuda": a HIP event pair around the block on the current stream, read after a synchronize."""

    def __init__(self, mode: str = "cpu"):
        if mode not in ("cpu", "cuda"):
            raise ValueError("mode must be 'cpu' or 'cuda'")
        self.mode = mode
        self.timings: Dict[str, List[float]] = defaultdict(list)
        self._handles = []

    def register_hooks(self, model: torch.nn.Module) -> None:
        def pre(module, inputs):
            if self.mode == "cpu":
                module._prof_t0 = time.perf_counter()
            else:
                module._prof_e0 = torch.cuda.Event(enable_timing=True)
                module._prof_e0.record()

        def post(name):
            def hook(module, inputs, output):
                if self.mode == "cpu":
                    self.timings[name].append((time.perf_counter() - module._prof_t0) * 1e3)
                else:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    torch.cuda.synchronize()
                    self.timings[name].append(module._prof_e0.elapsed_time(e1))
            return hook

        for i, layer in enumerate(decoder_layers(model)):
            for kind, words in (("attention", ("attn", "attention")), ("mlp", ("mlp", "feed_forward", "ffn"))):
                mods = _children_matching(layer, words)
                if not mods:
                    raise ValueError(f"decoder layer {i} has no {kind} block")
                for j, mod in enumerate(mods):
                    name = f"{kind}_layer_{i}" if len(mods) == 1 else f"{kind}_layer_{i}_{j}"
                    self._handles.append(mod.register_forward_pre_hook(pre))
                    self._handles.append(mod.register_forward_hook(post(name)))

    def clear_hooks(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []

    @torch.no_grad()
    def run_profiling(self, model: torch.nn.Module, fwd, warmup: int = 5, iters: int = 10) -> None:
        """`fwd(model)` runs one forward pass.  Warm-up passes run without hooks."""
        model.eval()
        for _ in range(warmup):
            fwd(model)
        self.register_hooks(model)
        try:
            for _ in range(iters):
                fwd(model)
        finally:
            self.clear_hooks()

    def summarize(self) -> Dict[str, float]:
        attn = sum(float(np.mean(v)) for k, v in self.timings.items() if k.startswith("attention"))
        mlp = sum(float(np.mean(v)) for k, v in self.timings.items() if k.startswith("mlp"))
        return {"attention_time": attn, "mlp_time": mlp, "ratio": attn / mlp if mlp > 0 else 0.0}
