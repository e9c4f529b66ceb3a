mkdir -p gpurun_out/pmc3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "MfmaUtil"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc3/$tag -o q -- python $R/tools/quick_bench.py --configs "8,8192,8192,0;1,4096,4096,1" --iters 2 > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc3/stats -o q -- python $R/tools/quick_bench.py --configs "8,8192,8192,0;1,4096,4096,1" --iters 3 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc3/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "w4_gemm_stream" in r["Kernel_Name"]:
            agg[(r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]; print(k, "n=%d mean=%.1f" % (len(v), sum(v) / len(v)))
PY
grep w4_gemm gpurun_out/pmc3/stats/q_kernel_stats.csv | cut -c1-200
