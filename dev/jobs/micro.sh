#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
(for k in 4096 8192; do
  for q in "anyq" "intq" "anyq --quantize-args per_row=False"; do
    echo "##### K=$k --quantize $q"
    timeout 600 python tools/microbenchmark.py --input-dim $k --output-dim $k --quantize $q 2>&1 | grep -v amdgpu.ids | tail -6
  done
  for bs in 4 8; do echo "##### K=$k anyq batch $bs"; timeout 600 python tools/microbenchmark.py --input-dim $k --output-dim $k --quantize anyq --batch-size $bs 2>&1 | grep -v amdgpu.ids | tail -3; done
done) > gpurun_out/microbenchmark.txt 2>&1
