#!/bin/bash
# reference numerics: hand-written lookup block (d16 loads, masked X read) against the compiler's loads + merges, same box
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/lk_tests.txt
export ANY4_AB_NUMERICS=reference
bash dev/exp.sh "1,4096,4096,1,any4_rowwise,128 8,4096,4096,1,any4_rowwise,128 1,4096,4096,1,int4,128 16,4096,4096,1,any4_rowwise,128 4,4096,4096,1,any4_rowwise,128" orig lk0 > gpurun_out/lk_ab.txt 2>&1
