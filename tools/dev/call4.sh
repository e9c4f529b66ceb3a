set -u
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "--no-fuse" "--interleave"; do
  tag=$(echo $v | tr -d '-')
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dec_$tag -o d -- python $R/tools/llama_decode_bench.py --config llama3_8b --steps 40 --warmup 5 $v > $R/gpurun_out/dec_$tag.log 2>&1
  tail -1 $R/gpurun_out/dec_$tag.log | cut -c1-400
  python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/dec_$tag/**/d_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:110].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e3,2), r['Percentage'])
PY
done
