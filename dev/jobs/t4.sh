#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_model.py tests/test_gpu_peer.py -q -m gpu 2>&1 | tail -4 > gpurun_out/t4.txt
for ms in 1024 2048 8192; do timeout 300 python tools/llama_decode_bench.py --steps 40 --warmup 10 --max-seq $ms --start-pos $((ms-150)) --interleave 2>&1 | tail -1 | cut -c330-400; done >> gpurun_out/t4.txt
