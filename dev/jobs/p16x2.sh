#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
CFG="8,4096,14336,1;16,4096,14336,1;8,4096,8192,1;16,4096,8192,1;12,8192,8192,1;6,4096,14336,1"
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
for v in orig p16_lds_full orig p16_lds_full; do
  if [ "$v" = orig ]; then cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so; else cp variants/$v.so any4_amd/lib/libtinygemm_hip.so; fi
  echo "=== $v"; timeout 300 python tools/quick_bench.py --configs "$CFG" --L 12 2>&1 | grep -v amdgpu.ids | grep -E "^m=|eager|graph|rror"
done > gpurun_out/p16x_ab2.txt 2>&1
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
