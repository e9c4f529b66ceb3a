R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
C="1,4096,4096,1;2,4096,4096,1;4,4096,4096,1;8,4096,4096,1;16,4096,4096,1;8,4096,4096,0;8,8192,8192,0"
timeout 300 python tools/quick_bench.py --configs "$C" --iters 3 2>&1 | grep -E "^m=|steady"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc5 -o q -- python $R/tools/quick_bench.py --configs "8,4096,4096,1;8,8192,8192,0;4,4096,4096,1;1,4096,4096,1" --iters 2 --settle 0 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc5/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "w4_gemm_stream" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("w4_gemm_stream_kernel")[1][:70], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, "n=%d mean=%.3g" % (len(v), sum(v) / len(v)))
PY
