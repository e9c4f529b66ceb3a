#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
CFG="16,4096,4096,1;8,4096,4096,1"
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
for v in orig p16a2 p16a4 p16a5 p16a6 orig; do
  if [ "$v" = orig ]; then cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so; else cp variants/$v.so any4_amd/lib/libtinygemm_hip.so; fi
  echo "=== $v"; timeout 300 python tools/quick_bench.py --configs "$CFG" --L 32 2>&1 | grep -v amdgpu.ids | grep -E "^m=|graph|rror"
done > gpurun_out/p16abl.txt 2>&1
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
