#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_model.py -q -m gpu 2>&1 | tail -3 > gpurun_out/ns.txt
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
for v in orig ns8 orig ns8; do
  if [ "$v" = orig ]; then cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so; else cp variants/$v.so any4_amd/lib/libtinygemm_hip.so; fi
  echo "=== $v"
  for sp in 136 900 1900; do timeout 300 python tools/llama_decode_bench.py --steps 50 --warmup 10 --max-seq 2048 --start-pos $sp --interleave 2>&1 | tail -1 | cut -c330-400; done
done >> gpurun_out/ns.txt 2>&1
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
