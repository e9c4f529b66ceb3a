#!/usr/bin/env python3
"""Golden fixture for the accuracy loop, made by importing the REFERENCE's Python in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_accuracy.py      -> tests/golden/accuracy_tiny_llama.npz

A tiny HuggingFace LlamaForCausalLM (config only, seeded random weights, float32 on the CPU) and a seeded token stream go
through
  * data_gptq.llama_eval (data_gptq.py:196-220)  -> the reference's perplexity;
  * calibrate.calibrate  (calibrate.py:74-183, prompt branch; hooks of 41-73) -> the reference's mean input activation of every
    nn.Linear, with abs=False and abs=True.
The fixture holds DATA only: the model's config numbers and weights, the tokens, and those outputs.  `lm_eval` (imported at
calibrate.py's module scope, absent here) and `bitsandbytes` are EMPTY stub packages on PYTHONPATH: nothing from them is called
on this path; the tokenizer is a stand-in whose encode() returns the fixture's calibration tokens.
"""
import os
import sys
import tempfile
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = os.environ.get("ANY4_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

CFG = dict(vocab_size=96, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
           max_position_embeddings=64, rms_norm_eps=1e-5, rope_theta=10000.0, tie_word_embeddings=False)
SEQLEN, WINDOWS = 32, 5


def build_model(state=None):
    import torch
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(1234)
    model = LlamaForCausalLM(LlamaConfig(**CFG, attn_implementation="eager")).float().eval()
    if state is not None:
        model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    return model


def main():
    import numpy as np
    import torch

    stub = tempfile.mkdtemp(prefix="any4_stub_")
    for pkg in ("bitsandbytes", "lm_eval"):
        os.makedirs(os.path.join(stub, pkg))
        open(os.path.join(stub, pkg, "__init__.py"), "w").close()
    with open(os.path.join(stub, "lm_eval", "utils.py"), "w") as f:
        f.write("def simple_parse_args_string(s):\n    raise NotImplementedError('stub')\n")
    sys.path[:0] = [stub, REF]
    import calibrate as ref_calibrate  # noqa: E402  (the reference's)
    import data_gptq as ref_gptq      # noqa: E402

    assert ref_calibrate.__file__.startswith(REF) and ref_gptq.__file__.startswith(REF)

    model = build_model()
    gen = torch.Generator().manual_seed(77)
    tokens = torch.randint(0, CFG["vocab_size"], (1, SEQLEN * WINDOWS + 7), generator=gen)   # a ragged tail the eval must ignore
    ppl = ref_gptq.llama_eval(model, types.SimpleNamespace(input_ids=tokens), "cpu", seqlen=SEQLEN)

    calib = torch.randint(0, CFG["vocab_size"], (1, 48), generator=gen)

    class Tok:  # calibrate()'s prompt branch only calls encode(prompt, return_tensors="pt")
        pad_token = eos_token = None

        def encode(self, prompt, return_tensors=None):
            return calib.clone()

    means = {}
    for use_abs in (False, True):
        m = ref_calibrate.calibrate(model, Tok(), prompt="a prompt that is not a file", abs=use_abs)
        for name, v in m.items():
            means[("abs/" if use_abs else "raw/") + name] = v.double().numpy()
    out = {"ppl": np.float64(ppl), "tokens": tokens.numpy(), "calib_tokens": calib.numpy(), "seqlen": np.int64(SEQLEN)}
    out.update({"state/" + k: v.detach().numpy() for k, v in model.state_dict().items()})
    out.update({"mean/" + k: v for k, v in means.items()})
    np.savez_compressed(os.path.join(HERE, "accuracy_tiny_llama.npz"), **out)
    print(f"ppl = {ppl:.6f}; {len(means)} activation means; wrote accuracy_tiny_llama.npz "
          f"({os.path.getsize(os.path.join(HERE, 'accuracy_tiny_llama.npz')) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
