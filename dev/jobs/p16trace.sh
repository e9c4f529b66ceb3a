#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
cp variants/p16trace.so any4_amd/lib/libtinygemm_hip.so
(for m in 16 8; do echo "##### m=$m"; timeout 200 python dev/gemv_trace.py --repeat 6 --m $m 2>&1 | grep -v amdgpu.ids; done) > gpurun_out/p16trace.txt 2>&1
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
