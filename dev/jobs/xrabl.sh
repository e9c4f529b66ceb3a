#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fast.py tests/test_gpu_fused.py -q -m gpu -k "xr or benchmarked or north or batch_shape or tc_ops or residual or mx4" 2>&1 | tail -4 > gpurun_out/xrpair.txt
bash dev/exp.sh "8,4096,4096,1,any4_rowwise,128 16,4096,4096,1,any4_rowwise,128 4,4096,4096,1,any4_rowwise,128" xrbase xrpair 2>&1 | grep -o "===.*\|[0-9.]* us/layer *[0-9.]*%\|MISMATCH" >> gpurun_out/xrpair.txt
