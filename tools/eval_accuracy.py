#!/usr/bin/env python3
"""Accuracy loop on the box (the reference's calibrate.py + data_gptq.py llama_eval driven from its eval scripts):
16-bit baseline perplexity -> activation calibration -> any4 / int4 / nf4 / mx4 quantization with the REAL kernels
(pseudo=False: Any4Linear & co. on the HIP library) -> perplexity again.

No hub access here: the model is a LlamaConfig with random weights (or --model-path <local checkpoint>), the text a
synthetic Markov/Zipf token stream (or --tokens <local .npy/.pt/.txt of token ids>).  With random weights the absolute
perplexity means nothing; the loop, the calibration plumbing and the deltas between quantizers are what this exercises.

    python tools/eval_accuracy.py --arch tiny --quantize anyq --calibrate
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from hf_benchmark import ARCH  # noqa: E402


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="tiny", choices=sorted(ARCH))
    ap.add_argument("--model-path", default=None)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--tokens", default=None, help="local token-id file (.npy / .pt / text); default: synthetic stream")
    ap.add_argument("--n-tokens", type=int, default=1 << 15)
    ap.add_argument("--seqlen", type=int, default=256)
    ap.add_argument("--nsamples", type=int, default=None)
    ap.add_argument("--calib-samples", type=int, default=8)
    ap.add_argument("--quantize", default="anyq", choices=["anyq", "intq", "nf4", "mx4"])
    ap.add_argument("--group-size", type=int, default=128)
    ap.add_argument("--calibrate", action="store_true", help="activation-aware any4 (sample_weight = mean |input| per layer)")
    ap.add_argument("--pseudo", action="store_true", help="fake-quant weights in 16-bit modules instead of the HIP kernels")
    a = ap.parse_args()
    if not torch.cuda.is_available() and not a.pseudo:
        raise SystemExit("needs a GPU unless --pseudo (no CPU fallback for the kernels)")
    from transformers import AutoModelForCausalLM, LlamaConfig

    from any4_amd import accuracy as A
    from any4_amd import quantize as Q

    dev = "cuda" if torch.cuda.is_available() else "cpu"
    torch.manual_seed(0)
    if a.model_path:
        model = AutoModelForCausalLM.from_pretrained(a.model_path, dtype=torch.bfloat16, local_files_only=True)
    else:
        cfg = dict(ARCH[a.arch])
        if a.layers is not None:
            cfg["num_hidden_layers"] = a.layers
        model = AutoModelForCausalLM.from_config(LlamaConfig(**cfg), dtype=torch.bfloat16)
    model = model.to(dev).eval()
    toks = A.load_tokens(a.tokens) if a.tokens else A.synthetic_corpus(model.config.vocab_size, a.n_tokens, seed=0)
    t0 = time.perf_counter()
    ppl0 = A.perplexity(model, toks, a.seqlen, a.nsamples)
    t_eval = time.perf_counter() - t0
    sw = None
    if a.calibrate:
        sw = A.calibrate(model, A.windows(toks, a.seqlen, a.calib_samples))
    layer_to = {"anyq": Q.anyq_layer, "intq": Q.intq_layer, "nf4": Q.nf4_layer, "mx4": Q.mx4_layer}[a.quantize]
    kw = dict(group_size=32 if a.quantize == "mx4" else a.group_size, pseudo=a.pseudo)
    if sw is not None and a.quantize == "anyq":
        kw["sample_weight"] = sw
    t0 = time.perf_counter()
    Q.quantize_model(model, layer_from=torch.nn.Linear, layer_to=layer_to, skip_modules=["lm_head"], **kw)
    t_q = time.perf_counter() - t0
    ppl1 = A.perplexity(model, toks, a.seqlen, a.nsamples)
    kinds = sorted({type(m).__name__ for m in model.modules() if type(m).__name__.endswith("Linear")})
    print(json.dumps({"model": a.model_path or a.arch, "layers": model.config.num_hidden_layers, "quantize": a.quantize,
                      "calibrated": bool(sw), "pseudo": a.pseudo, "group_size": kw["group_size"], "seqlen": a.seqlen,
                      "windows": len(A.windows(toks, a.seqlen, a.nsamples)), "ppl_16bit": round(ppl0, 4), "ppl_quantized": round(ppl1, 4),
                      "ppl_ratio": round(ppl1 / ppl0, 5), "linear_kinds": kinds, "eval_s": round(t_eval, 2), "quantize_s": round(t_q, 2),
                      "data": "synthetic Zipf/Markov stream" if not a.tokens else a.tokens}))


if __name__ == "__main__":
    main()
