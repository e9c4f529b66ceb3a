#!/usr/bin/env python3
"""Model-level decode benchmark (BASELINE config 5; counterpart of the reference's benchmark.py:113-215).

    python tools/llama_decode_bench.py --config llama3_8b --baseline
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        tools/llama_decode_bench.py --config llama3_8b                   # TP=8, rows sharded, RCCL all-gathers

Random-initialised weights of the named architecture (no checkpoints here), random token ids, one new token per
sequence per step over a static KV cache.  Prints one JSON line on rank 0: ms per token, tokens/s, and the rate at which
the 4-bit weights stream (algorithmic bytes of the quantized linears / step time).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def time_steps(stack, steps, warmup, start_pos):
    tok = torch.randint(0, stack.cfg.vocab, (stack.bs,), device=stack.tokens.device)
    for i in range(warmup):
        stack.decode(tok, start_pos + i)
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        stack.decode(tok, start_pos + warmup + i)
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="llama3_8b", choices=["llama3_8b", "llama2_7b", "tiny"])
    ap.add_argument("--layers", type=int, default=None, help="override the number of decoder layers")
    ap.add_argument("--bs", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--max-seq", type=int, default=1024)
    ap.add_argument("--start-pos", type=int, default=128, help="sequence position of the first timed token")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--interleave", action="store_true", help="gate_up rows in blocks of 8 gate + 8 up (lets the GEMM fuse SwiGLU)")
    ap.add_argument("--no-fuse", action="store_true", help="separate glue kernels around the GEMMs (8 launches per layer instead of 5)")
    ap.add_argument("--baseline", action="store_true", help="also time the same stack with 16-bit nn.Linear")
    ap.add_argument("--kernel", default="linear_y_f16RM_x_f16RM_W_any4TC")
    ap.add_argument("--emulate-tp", type=int, default=0,
                    help="single process, timing only: build rank 0 of a TP=N model and replace the all-gathers by local "
                         "copies -> per-GPU compute time of a TP=N step without the interconnect")
    ap.add_argument("--gather", default="rccl", choices=["rccl", "peer"],
                    help="TP > 1: RCCL all_gather_into_tensor per exchange, or the one-shot peer-write gather (include/peer_gather_hip.h)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo needs --gather peer (no CUDA collectives: only the IPC handles travel through it)")
    ap.add_argument("--same-device", action="store_true",
                    help="functional check on a one-GPU box: every rank uses GPU 0 (needs --backend gloo --gather peer); the time "
                         "it prints is two processes sharing one GPU, not a TP measurement")
    a = ap.parse_args()
    if a.backend == "gloo" and a.gather != "peer":
        raise SystemExit("--backend gloo moves no CUDA tensors: use it with --gather peer")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU (the HIP path has no CPU fallback)")
    if a.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend=a.backend)

    from any4_amd.decode import Any4Factory, DecodeConfig, DecodeStack, DenseFactory, memory_allocated_mb

    if a.config == "tiny":
        cfg = DecodeConfig(hidden=512, inter=1024, layers=2, heads=8, kv_heads=8, head_dim=64, vocab=1024, max_seq=a.max_seq)
    else:
        cfg = getattr(DecodeConfig, a.config)(max_seq=a.max_seq)
    if a.layers is not None:
        cfg.layers = a.layers
    if a.interleave:
        cfg.gate_up_interleave = 8

    def run(factory_cls, label):
        torch.cuda.reset_peak_memory_stats(device)
        fac = factory_cls(cfg, device, torch.bfloat16, seed=1 + rank) if factory_cls is DenseFactory else \
            factory_cls(cfg, device, torch.bfloat16, seed=1 + rank, kernel=a.kernel)
        if a.emulate_tp > 1:
            stack = DecodeStack(cfg, fac, device, torch.bfloat16, bs=a.bs, rank=0, world=a.emulate_tp, emulate_gather=True)
        else:
            stack = DecodeStack(cfg, fac, device, torch.bfloat16, bs=a.bs, rank=rank, world=world, gather=a.gather, fuse_gemm_stages=not a.no_fuse)
        graph = False
        if not a.no_graph:
            try:
                stack.capture()
                graph = True
            except Exception as e:  # noqa: BLE001  (symmetric across ranks: same code, same shapes)
                if rank == 0:
                    print(f"[{label}] graph capture failed ({type(e).__name__}: {e}); timing eager", file=sys.stderr)
                stack._graph = None
        dt = time_steps(stack, a.steps, a.warmup, a.start_pos)
        out = {"linears": label, "ms_per_token": round(dt * 1e3, 4), "tokens_per_s": round(a.bs / dt, 1),
               "hipgraph": graph, "peak_mem_mib": round(memory_allocated_mb(device), 1)}
        if label == "any4":
            out["weight_stream_GBps_all_ranks"] = round(cfg.weight_bytes_4bit() / dt / 1e9, 1)
        for pg in stack._peer.values():
            pg.check()
            pg.close()
        del stack, fac
        torch.cuda.empty_cache()
        return out

    res = {"config": a.config, "layers": cfg.layers, "bs": a.bs, "tp": world, "gather": a.gather if world > 1 else None,
           "ranks_share_one_gpu": bool(a.same_device), "emulated_tp_compute_only": a.emulate_tp or None, "max_seq": cfg.max_seq,
           "steps": a.steps, "warmup": a.warmup, "data": "synthetic (random weights, random tokens)",
           "algorithmic_4bit_bytes_per_token": cfg.weight_bytes_4bit()}
    res["any4"] = run(Any4Factory, "any4")
    if a.baseline:
        res["bf16"] = run(DenseFactory, "bf16")
        res["speedup_vs_bf16"] = round(res["bf16"]["ms_per_token"] / res["any4"]["ms_per_token"], 3)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
