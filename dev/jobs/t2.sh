#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_fast.py tests/test_gpu_gemv.py tests/test_gpu_fused.py tests/test_gpu_decode.py tests/test_gpu_aside.py -q -m gpu 2>&1 | tail -8 > gpurun_out/t2.txt
for bs in 4 8; do timeout 300 python tools/llama_decode_bench.py --steps 30 --warmup 5 --max-seq 1024 --start-pos 136 --interleave --bs $bs 2>&1 | tail -1 | cut -c330-420; done >> gpurun_out/t2.txt
timeout 200 python tools/quick_bench.py --configs "3,4096,4096,1;4,4096,4096,1;5,4096,4096,1;8,4096,4096,1;4,28672,4096,1;8,28672,4096,1;8,6144,4096,1" --L 12 2>&1 | grep -E "^m=|graph" | paste - - | awk '{print $1,$2,$3,$(NF-7),$(NF-6)}' >> gpurun_out/t2.txt
