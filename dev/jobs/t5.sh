#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemv.py tests/test_gpu_fused.py -q -m gpu 2>&1 | tail -4 > gpurun_out/t5.txt
timeout 300 python tools/quick_bench.py --configs "4,8192,8192,1;3,8192,8192,1;1,8192,8192,1;2,8192,8192,1" --L 12 2>&1 | grep -E "^m=|graph" | paste - - | awk '{print $1,$2,$3,$(NF-7),$(NF-6)}' >> gpurun_out/t5.txt
