#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decode.py -q -m gpu 2>&1 | tail -4 > gpurun_out/split.txt
for ns in 1 2 4 8; do
  echo "=== ANY4_ATTN_SPLIT=$ns"
  for sp in 136 500 900 1900; do ANY4_ATTN_SPLIT=$ns timeout 300 python tools/llama_decode_bench.py --steps 50 --warmup 10 --max-seq 2048 --start-pos $sp --interleave 2>&1 | tail -1 | cut -c330-400; done
done >> gpurun_out/split.txt 2>&1
