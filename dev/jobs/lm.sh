#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_model.py -q -m gpu 2>&1 | tail -4 > gpurun_out/lm.txt
