#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
cp variants/trace.so any4_amd/lib/libtinygemm_hip.so
timeout 300 python dev/gemv_trace.py --layers 2 2>&1 | grep -v amdgpu.ids > gpurun_out/decode_trace.txt
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
