# One GPU-box visit (round 2): smoke, gpu tests, bench line, rocprof kernel stats + PMC traffic + SQ counters.  Writes gpurun_out/.
set -u
RN=${ROUND:-r02}
mkdir -p gpurun_out/prof
R=$GRAFT_REPO_ROOT
cd $R
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gpu_tests.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.err; cp gpurun_out/bench.log gpurun_out/${RN}_bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -o bench -- python $R/bench.py --roofline-only > $R/gpurun_out/prof/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch -o bench -- python $R/bench.py --steps 5 --warmup 2 --settle-s 0.05 --roofline-only > $R/gpurun_out/prof/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_write -o bench -- python $R/bench.py --steps 5 --warmup 2 --settle-s 0.05 --roofline-only > $R/gpurun_out/prof/pmc_write.log 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "MfmaUtil"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/prof/sq$i -o bench -- python $R/bench.py --steps 5 --warmup 2 --settle-s 0.05 --roofline-only > $R/gpurun_out/prof/sq$i.log 2>&1
done
cd $R
cp gpurun_out/prof/stats/bench_kernel_stats.csv gpurun_out/${RN}_bench_kernel_stats.csv 2>/dev/null
head -3 gpurun_out/${RN}_bench_kernel_stats.csv | cut -c1-300
python - <<PY
import csv, glob, collections, json
RN = "${RN}"
line = json.loads(open("gpurun_out/bench.log").read().strip().splitlines()[-1])
L = line["config"]["layers_per_step"]
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof/*/*counter_collection.csv") + glob.glob("gpurun_out/prof/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "w4_gemm_pair" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
rd = 2 * c.get("FETCH_SIZE", 0) * 1024
wr = c.get("WRITE_SIZE", 0) * 1024
alg = line["roofline"]["bytes_per_launch"]
traffic = {
    "command": "rocprofv3 --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- python bench.py --steps 5 --warmup 2 --settle-s 0.05 --roofline-only   (tools/gpu_round.sh)",
    "kernel": line["roofline"]["kernel"], "layers_per_launch": L,
    "raw_mean_KB": {"FETCH_SIZE": c.get("FETCH_SIZE"), "WRITE_SIZE": c.get("WRITE_SIZE")},
    "correction": "gfx950 rocprofv3 FETCH_SIZE tallies the 128-B requests of 16-B/lane streaming reads at 64 B (MI355X_MICROARCH.md, HBM section): read bytes = 2 * FETCH_SIZE[KB] * 1024; WRITE_SIZE[KB] * 1024 as reported",
    "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
    "hbm_bytes_per_layer": (rd + wr) / L, "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": round((rd + wr) / alg, 4),
}
json.dump(traffic, open(f"gpurun_out/{RN}_pmc_traffic.json", "w"), indent=1)
print("traffic/algorithmic", traffic["traffic_over_algorithmic"])
sq = {k: v for k, v in c.items() if k not in ("FETCH_SIZE", "WRITE_SIZE")}
sq["_note"] = f"per launch of {L} layers (512 persistent workgroups x 8 waves); SQ_* summed over the chip, GRBM_* summed over 8 XCDs"
json.dump(sq, open(f"gpurun_out/{RN}_sq_counters_pair.json", "w"), indent=1)
for k, v in sq.items():
    print(k, v)
PY
echo "== quick_bench default dispatch"; timeout 600 python tools/quick_bench.py --configs "1,4096,4096,1;2,4096,4096,1;4,4096,4096,1;8,4096,4096,1;16,4096,4096,1;1,8192,8192,1;8,8192,8192,1;1,14336,4096,1;1,4096,14336,1;8,4096,14336,1;1,4096,4096,0;8,4096,4096,0;1,8192,8192,0;8,8192,8192,0;16,8192,8192,0" --L 256 --iters 3 2>&1 | grep -E "^m=|eager|graph|stacked|steady" | tee gpurun_out/${RN}_quick_bench_default.txt
for q in "int4" "any4_global" "mx4" "int4 --g 32" "any4_rowwise --g 64" "any4_rowwise --g 256"; do timeout 300 python tools/quick_bench.py --configs "1,4096,4096,1;8,4096,4096,1;8,8192,8192,0" --qtype $q --L 256 --iters 3 2>&1 | grep -E "^m=|stacked|steady"; done | tee gpurun_out/${RN}_quick_bench_variants.txt
