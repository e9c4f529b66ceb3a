#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
bash dev/exp.sh "8,4096,4096,1,any4_rowwise,128 16,4096,4096,1,any4_rowwise,128" xr8 xr16 > gpurun_out/xr16_ab.txt 2>&1
