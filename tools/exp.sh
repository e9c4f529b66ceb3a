mkdir -p gpurun_out/pmc
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
 for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | cut -d' ' -f1)
  TG_STREAM=$mode TG_VARIANT=801 timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc/m${mode}_$tag -o q -- python $R/tools/quick_bench.py --configs "1,4096,4096,1" --iters 2 > /dev/null 2>&1
 done
done
cd $R
python - <<'PY'
import csv, glob, collections
for mode in (0, 1):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc/m{mode}_*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "w4_gemm" in r["Kernel_Name"] and r["Grid_Size"] == "1048576":
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("mode", mode, {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
