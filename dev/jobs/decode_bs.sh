#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
for bs in 1 2 4 8 16; do
  echo "##### bs=$bs"; timeout 400 python tools/llama_decode_bench.py --steps 30 --warmup 5 --max-seq 1024 --start-pos 136 --interleave --bs $bs $([ $bs -le 8 ] && echo --baseline) 2>&1 | grep -v amdgpu.ids | tail -4
done > gpurun_out/decode_bs.txt 2>&1
