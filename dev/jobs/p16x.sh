#!/bin/bash
# same-box A/B of the register-resident-activation path of w4_gemm_pair16_kernel (single launches, m = 5 ... 16)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fast.py -x -q -m gpu -k "pair16 or fused or north" 2>&1 | tail -5 > gpurun_out/p16x_tests.txt
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_aside.py -x -q -m gpu 2>&1 | tail -3 >> gpurun_out/p16x_tests.txt
CFG="5,4096,4096,1;8,4096,4096,1;12,4096,4096,1;16,4096,4096,1;8,6144,4096,1;16,6144,4096,1;8,4096,14336,1;16,4096,14336,1;2,4096,4096,1;8,4096,8192,1"
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
for v in orig p16_lds p16_x1 orig p16_lds; do
  if [ "$v" = orig ]; then cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so; else cp variants/$v.so any4_amd/lib/libtinygemm_hip.so; fi
  echo "=== $v"; timeout 300 python tools/quick_bench.py --configs "$CFG" --L 32 2>&1 | grep -v amdgpu.ids | grep -E "^m=|eager|graph"
done > gpurun_out/p16x_ab.txt 2>&1
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
