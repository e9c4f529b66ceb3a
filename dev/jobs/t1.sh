#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_gemv.py tests/test_gpu_decode.py tests/test_gpu_fused.py tests/test_gpu_aside.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/t1.txt
