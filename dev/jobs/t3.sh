#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_gemv.py tests/test_gpu_fused.py tests/test_gpu_decode.py tests/test_gpu_model.py tests/test_gpu_aside.py -q -m gpu 2>&1 | tail -5 > gpurun_out/t3.txt
timeout 300 python tools/llama_decode_bench.py --steps 50 --warmup 10 --max-seq 1024 --start-pos 136 --interleave 2>&1 | tail -1 | cut -c330-420 >> gpurun_out/t3.txt
timeout 300 python tools/llama_decode_bench.py --steps 50 --warmup 10 --max-seq 1024 --start-pos 136 --interleave --bs 2 2>&1 | tail -1 | cut -c330-420 >> gpurun_out/t3.txt
timeout 200 python tools/quick_bench.py --configs "1,4096,4096,1;1,4096,14336,1;2,4096,14336,1;1,28672,4096,1" --L 12 2>&1 | grep -E "^m=|graph" | paste - - | awk '{print $1,$2,$3,$(NF-7),$(NF-6)}' >> gpurun_out/t3.txt
