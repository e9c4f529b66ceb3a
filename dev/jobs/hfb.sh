#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
(for q in anyq intq; do echo "##### --arch llama3_8b --layers 8 --quantize $q (bs = 1, seqlen = 1)"; timeout 900 python tools/hf_benchmark.py --arch llama3_8b --layers 8 --quantize $q 2>&1 | grep -v "amdgpu.ids\|ROCTracer\|Warning" | tail -14; done) > gpurun_out/hf_benchmark.txt 2>&1
