#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fast.py -q -m gpu -k "launch_plan" 2>&1 | tail -3 > gpurun_out/plan.txt
timeout 1200 python -m pytest tests/test_gpu_decode.py tests/test_gpu_peer.py tests/test_gpu_aside.py tests/test_gpu_fused.py -q -m gpu 2>&1 | tail -3 >> gpurun_out/plan.txt
timeout 300 python dev/jobs/plandbg.py 2>&1 | grep "wall per call" >> gpurun_out/plan.txt
for q in anyq intq; do timeout 600 python tools/microbenchmark.py --input-dim 4096 --output-dim 4096 --quantize $q 2>&1 | grep -v "amdgpu.ids\|ROCTracer" | tail -5; done >> gpurun_out/plan.txt
