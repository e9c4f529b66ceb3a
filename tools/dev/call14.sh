set -u
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/decode -o dec -- python $R/tools/llama_decode_bench.py --config llama3_8b --steps 60 --warmup 10 --interleave > $R/gpurun_out/prof_decode.log 2>&1
cd $R
f=$(ls gpurun_out/prof/decode/*/dec_kernel_stats.csv gpurun_out/prof/decode/dec_kernel_stats.csv 2>/dev/null | head -1)
cp $f gpurun_out/decode_kernel_stats.csv
tail -2 gpurun_out/prof_decode.log | cut -c1-600
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/decode_kernel_stats.csv')))
for r in rows[:16]:
    print(r['Name'][:150].replace('(anonymous namespace)::',''), r['Calls'], r['AverageNs'], r['Percentage'])
P
