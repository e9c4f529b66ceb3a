#!/usr/bin/env python3
"""HuggingFace causal LM forward: 16-bit baseline vs `quantize_model(..., pseudo=False)` (counterpart of the reference's
benchmark.py:113-215, same protocol: random input_ids[bs, seqlen], warm-up / iterations, wall-clock and device-only
time, model size, peak memory).  There is no hub access here, so the model is built from a LlamaConfig with random
weights (`--arch llama3_8b|llama2_7b|tiny`, `--layers N` to shorten); with a local checkpoint directory use
`--model-path`.

    python tools/hf_benchmark.py --arch llama3_8b --layers 8 --quantize anyq
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

ARCH = {
    "llama3_8b": dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                      num_key_value_heads=8, vocab_size=128256, rope_theta=500000.0, max_position_embeddings=8192),
    "llama2_7b": dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                      num_key_value_heads=32, vocab_size=32000, max_position_embeddings=4096),
    "tiny": dict(hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=8,
                 num_key_value_heads=4, vocab_size=1024, max_position_embeddings=512),
}


def model_size_bytes(model) -> int:
    return sum(p.numel() * p.element_size() for p in model.parameters()) + sum(b.numel() * b.element_size() for b in model.buffers())


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="llama3_8b", choices=sorted(ARCH))
    ap.add_argument("--model-path", default=None, help="local HF checkpoint directory (optional)")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=1)
    ap.add_argument("--seqlen", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--quantize", default="anyq", choices=["anyq", "intq"])
    ap.add_argument("--group-size", type=int, default=128)
    ap.add_argument("--graph", action="store_true",
                    help="also time both models with the forward captured in ONE hipGraph (torch.cuda.CUDAGraph) and replayed: what the GPU "
                         "needs per forward, without HuggingFace's ~0.3 ms of Python per decoder layer (the eager numbers are host-bound at seqlen 1)")
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU (no CPU fallback)")
    from transformers import AutoModelForCausalLM, LlamaConfig

    from any4_amd import quantize as Q
    from any4_amd.bench_utils import benchmark_cuda_only_in_ms, benchmark_in_ms, memory_allocated_mb

    torch.manual_seed(0)
    if a.model_path:
        model = AutoModelForCausalLM.from_pretrained(a.model_path, dtype=torch.bfloat16, local_files_only=True)
    else:
        cfg = dict(ARCH[a.arch])
        if a.layers is not None:
            cfg["num_hidden_layers"] = a.layers
        model = AutoModelForCausalLM.from_config(LlamaConfig(**cfg), dtype=torch.bfloat16)
    model = model.to("cuda").eval()
    ids = torch.randint(0, model.config.vocab_size, (a.batch_size, a.seqlen), device="cuda")
    mask = torch.ones_like(ids)
    f = lambda m: m(input_ids=ids, attention_mask=mask, use_cache=False)

    def graphed_ms(m):
        """(ms per replay on the device, ms per replay wall, logits of the captured forward) or an error string."""
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    f(m)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                logits = f(m).logits
            for _ in range(max(3, a.warmup // 4)):
                gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            w0 = time.perf_counter()
            e0.record()
            for _ in range(a.iters):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / a.iters, (time.perf_counter() - w0) * 1e3 / a.iters, logits.float().clone()
        except Exception as e:  # noqa: BLE001  (a model whose forward cannot be captured: report, keep the eager numbers)
            torch.cuda.synchronize()
            return f"{type(e).__name__}: {e}"

    from any4_amd.accuracy import HookProfiler

    def split(m):  # attention / MLP time per forward (benchmark.py:37-111): host wall-clock and device events
        out = {}
        for mode in ("cpu", "cuda"):
            prof = HookProfiler(mode)
            prof.run_profiling(m, f, warmup=3, iters=max(3, a.iters // 5))
            out[mode] = prof.summarize()
        return out

    torch.cuda.reset_peak_memory_stats()
    t, tc = benchmark_in_ms(f, a.warmup, a.iters, model), benchmark_cuda_only_in_ms(f, a.warmup, a.iters, model)
    size0, peak0 = model_size_bytes(model), memory_allocated_mb()
    split0 = split(model)
    ref = f(model).logits.float()
    g0 = graphed_ms(model) if a.graph else None

    t0 = time.perf_counter()
    layer_to = Q.anyq_layer if a.quantize == "anyq" else Q.intq_layer
    Q.quantize_model(model, layer_from=torch.nn.Linear, layer_to=layer_to, pseudo=False, group_size=a.group_size)
    torch.cuda.synchronize()
    tq_quant = time.perf_counter() - t0
    torch.cuda.reset_peak_memory_stats()
    qt, qtc = benchmark_in_ms(f, a.warmup, a.iters, model), benchmark_cuda_only_in_ms(f, a.warmup, a.iters, model)
    out = f(model).logits.float()
    g1 = graphed_ms(model) if a.graph else None
    split1 = split(model)
    n_q = sum(type(m).__name__ in ("Any4Linear", "Int4Linear") for m in model.modules())

    print(f"Model: {a.model_path or a.arch}  layers={model.config.num_hidden_layers}  bs={a.batch_size} seqlen={a.seqlen}")
    print("Baseline:")
    print(f"\tModel Size:\t{size0 / 2**30:.2f} GB\tPeak: {peak0:.0f} MB")
    print(f"\tModel:\tTotal {t:.3f} ms\tCUDA {tc:.3f} ms")
    print(f"Quantized ({a.quantize}, {n_q} linears swapped in {tq_quant:.1f} s):")
    print(f"\tModel Size:\t{model_size_bytes(model) / 2**30:.2f} GB\tPeak: {memory_allocated_mb():.0f} MB")
    print(f"\tModel:\tTotal {qt:.3f} ms\tCUDA {qtc:.3f} ms")
    print(f"Speedup:\tTotal {t / qt:.2f}x\tCUDA {tc / qtc:.2f}x")
    if a.graph:
        if isinstance(g0, str) or isinstance(g1, str):
            print(f"hipGraph:\tcapture failed: baseline {g0 if isinstance(g0, str) else 'ok'}; quantized {g1 if isinstance(g1, str) else 'ok'}")
        else:
            print(f"hipGraph (one captured forward, {a.iters} replays):\tbaseline {g0[0]:.3f} ms device / {g0[1]:.3f} ms wall\tquantized {g1[0]:.3f} / {g1[1]:.3f} ms"
                  f"\tSpeedup {g0[0] / g1[0]:.2f}x device / {g0[1] / g1[1]:.2f}x wall")
            print(f"\tcaptured logits vs eager: baseline max |d| = {(g0[2] - ref).abs().max():.3g}, quantized max |d| = {(g1[2] - out).abs().max():.3g}")
    for kind in ("attention", "mlp"):
        b_c, b_g = split0["cpu"][f"{kind}_time"], split0["cuda"][f"{kind}_time"]
        q_c, q_g = split1["cpu"][f"{kind}_time"], split1["cuda"][f"{kind}_time"]
        print(f"\t{kind:<10} baseline {b_c:.3f} / {b_g:.3f} ms ({b_c / t * 100:.0f} % / {b_g / tc * 100:.0f} % of the model)"
              f"\tquantized {q_c:.3f} / {q_g:.3f} ms\tspeedup {b_c / q_c:.2f}x / {b_g / q_g:.2f}x   (total / CUDA)")
    print(f"\tattention : MLP ratio\tbaseline {split0['cuda']['ratio']:.2f}\tquantized {split1['cuda']['ratio']:.2f}")
    print(f"logits: max |quantized - baseline| = {(out - ref).abs().max():.3f} at max |baseline| = {ref.abs().max():.3f}")


if __name__ == "__main__":
    main()
