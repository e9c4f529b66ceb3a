# One GPU-box visit: smoke, gpu tests, bench line, rocprof kernel stats + PMC traffic.  Writes gpurun_out/.
set -u
mkdir -p gpurun_out/prof
R=$GRAFT_REPO_ROOT
cd $R
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gpu_tests.log
timeout 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.log; tail -3 gpurun_out/bench.err; cp gpurun_out/bench.log gpurun_out/r01_bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -o bench -- python $R/bench.py --roofline-only > $R/gpurun_out/prof/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch -o bench -- python $R/bench.py --steps 5 --warmup 2 --roofline-only > $R/gpurun_out/prof/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_write -o bench -- python $R/bench.py --steps 5 --warmup 2 --roofline-only > $R/gpurun_out/prof/pmc_write.log 2>&1
cd $R
find gpurun_out/prof -name "*.csv" | head -20
head -4 gpurun_out/prof/stats/bench_kernel_stats.csv | cut -c1-260
python - <<'PY'
import csv, glob, collections, json
raw = {}
for d, c in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/prof/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "w4_gemm" in r["Kernel_Name"] and r["Counter_Name"] == c:
                kind = "stream" if "stream" in r["Kernel_Name"] else "splitk"
                agg[f"{kind}:{r['Grid_Size']}"].append(float(r["Counter_Value"]))
    raw[c] = {k: {"n": len(v), "mean_KB": sum(v) / len(v)} for k, v in agg.items()}
    print(c, raw[c])
line = json.loads(open("gpurun_out/bench.log").read().strip().splitlines()[-1])
L = line["config"]["layers_per_step"]
key = max((k for k in raw["FETCH_SIZE"] if k.startswith("stream")), key=lambda k: int(k.split(":")[1]))
rd = 2 * raw["FETCH_SIZE"][key]["mean_KB"] * 1024
wr = raw["WRITE_SIZE"][key]["mean_KB"] * 1024
alg = line["roofline"]["bytes_per_launch"]
out = {
    "command": "rocprofv3 --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- python bench.py --steps 5 --warmup 2 --roofline-only   (tools/gpu_round.sh)",
    "kernel": line["roofline"]["kernel"], "layers_per_launch": L, "raw": raw,
    "correction": "gfx950 rocprofv3 FETCH_SIZE tallies the 128-B requests of 16-B/lane streaming reads at 64 B (MI355X_MICROARCH.md, HBM section): read bytes = 2 * FETCH_SIZE[KB] * 1024; WRITE_SIZE[KB] * 1024 as reported",
    "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
    "hbm_bytes_per_layer": (rd + wr) / L, "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": round((rd + wr) / alg, 4),
}
json.dump(out, open("gpurun_out/r01_pmc_traffic.json", "w"), indent=1)
print("traffic/algorithmic", out["traffic_over_algorithmic"])
PY
echo "== quick_bench default dispatch"; timeout 300 python tools/quick_bench.py --configs "1,4096,4096,1;8,4096,4096,1;16,4096,4096,1;1,4096,4096,0;8,8192,8192,0;1,8192,8192,1" --iters 3 2>&1 | grep -E "^m=|eager|stacked|steady" | tee gpurun_out/qb_default.log
for q in int4 any4_global mx4; do timeout 200 python tools/quick_bench.py --configs "1,4096,4096,1" --qtype $q --iters 3 2>&1 | grep -E "^m=|eager|stacked|steady"; done | tee gpurun_out/qb_variants.log
# MFMA utilisation / HBM traffic at m = 8 and 16 (north_star: "MFMA utilisation at m=8/16"), separate --pmc passes
cd /tmp
for c in MfmaUtil FETCH_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/prof/pmc_m8_$c -o q -- python $R/tools/quick_bench.py --configs "8,4096,4096,1;16,4096,4096,1;8,8192,8192,0" --iters 2 --settle 0 > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof/pmc_m8_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "w4_gemm_stream" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("w4_gemm_stream_kernel")[1][:60], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
out = {f"{k[0]} grid={k[1]} {k[2]}": {"n": len(v), "mean": sum(v) / len(v)} for k, v in sorted(agg.items())}
json.dump(out, open("gpurun_out/r01_pmc_m8_m16.json", "w"), indent=1)
for k, v in out.items():
    print(k, v)
PY
