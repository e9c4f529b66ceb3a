#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
bash dev/exp.sh "1,4096,4096,1,any4_rowwise,128 1,4096,4096,1,int4,128" base2 abl5 abl1 > gpurun_out/abl5.txt 2>&1
