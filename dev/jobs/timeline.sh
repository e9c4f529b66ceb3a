set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tl
cd /tmp && export TMPDIR=/tmp
for variant in "" "--interleave"; do
tag=$( [ -z "$variant" ] && echo plain || echo il8 )
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o dec -- python $R/tools/llama_decode_bench.py --steps 20 --warmup 5 --max-seq 1024 --start-pos 136 $variant > $R/gpurun_out/tl/bench_$tag.log 2>&1
f=$(find /tmp/tl_$tag -name 'dec_kernel_trace.csv' | head -1)
python $R/tools/decode_timeline.py "$f" --last 10 > $R/gpurun_out/tl/timeline_$tag.txt 2>&1
tail -2 $R/gpurun_out/tl/bench_$tag.log
cat $R/gpurun_out/tl/timeline_$tag.txt
done
