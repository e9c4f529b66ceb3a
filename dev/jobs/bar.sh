#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
for v in bar0 bar1 bar0 bar1; do
  cp variants/$v.so any4_amd/lib/libtinygemm_hip.so
  echo "=== $v"; timeout 300 python tools/llama_decode_bench.py --steps 50 --warmup 10 --max-seq 1024 --start-pos 136 --interleave 2>&1 | tail -1 | cut -c330-420
  timeout 200 python tools/quick_bench.py --configs "1,4096,4096,1;4,4096,4096,1;8,4096,4096,1;1,28672,4096,1;1,6144,4096,1;1,4096,14336,1" --L 12 2>&1 | grep -E "^m=|graph" | paste - - | awk '{print $1,$2,$3,$(NF-7),$(NF-6)}'
done > gpurun_out/bar.txt 2>&1
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
