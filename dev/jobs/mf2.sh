#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
for v in orig mf1 orig mf1; do
  if [ "$v" = orig ]; then cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so; else cp variants/$v.so any4_amd/lib/libtinygemm_hip.so; fi
  echo "=== $v"; timeout 300 python tools/llama_decode_bench.py --steps 40 --warmup 10 --max-seq 1024 --start-pos 136 --interleave 2>&1 | tail -1 | cut -c330-420
  timeout 200 python tools/quick_bench.py --configs "1,4096,4096,1;1,28672,4096,1;1,6144,4096,1;1,14336,4096,1;1,4096,4096,0" --L 12 2>&1 | grep -E "^m=|graph" | paste - - | awk '{print $1,$2,$3,$(NF-7),$(NF-6)}'
done > gpurun_out/mf1.txt 2>&1
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
