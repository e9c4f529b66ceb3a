#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for bs in 4 8; do
rm -rf /tmp/dp$bs
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp$bs -o dec -- python $R/tools/llama_decode_bench.py --steps 20 --warmup 5 --max-seq 1024 --start-pos 136 --interleave --bs $bs > /tmp/dec$bs.log 2>&1
python $R/tools/decode_timeline.py "$(find /tmp/dp$bs -name 'dec_kernel_trace.csv' | head -1)" --last 6 > $R/gpurun_out/decode_timeline_bs$bs.txt 2>&1
done
