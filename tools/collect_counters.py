#!/usr/bin/env python3
"""Run one command under several rocprofv3 --pmc passes (and optionally a --kernel-trace --stats pass) and aggregate the
counters of the kernels whose name contains a given substring into one JSON object.

    python tools/collect_counters.py --out gpurun_out/r03_counters_m8.json --match w4_gemm_pair_kernel \
        --label "m=8 Bint4 4096^2" -- python tools/ab.py 8,4096,4096,1,any4_rowwise,128

Counters are collected in their own passes (never together with a trace); each pass holds what fits the gfx950 PMC slots
(MI355X_MICROARCH.md, 'rocprofv3 PMC slots').  Values are the MEAN over the matching dispatches of a pass.
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

PASSES = [
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU",
    "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD",
    "GRBM_GUI_ACTIVE GRBM_COUNT",
    "MfmaUtil",
    "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum",
    "FETCH_SIZE",
    "WRITE_SIZE",
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--match", default="w4_gemm")
    ap.add_argument("--exclude", default="xprep")
    ap.add_argument("--label", default="")
    ap.add_argument("--stats-csv", default=None, help="also run a --kernel-trace --stats pass and copy its kernel_stats.csv here")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    root = os.getcwd()
    cmd = [c if not (c.endswith(".py") and os.path.exists(c)) else os.path.abspath(c) for c in cmd]
    env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    tmp = tempfile.mkdtemp(prefix="cc_", dir="/tmp")
    agg = collections.defaultdict(list)
    names = collections.Counter()
    for i, counters in enumerate(PASSES):
        d = os.path.join(tmp, f"p{i}")
        r = subprocess.run([prof, "--pmc", *counters.split(), "--output-format", "csv", "-d", d, "-o", "p", "--", *cmd],
                           cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        if r.returncode != 0:
            print(f"pass {i} ({counters}) failed rc={r.returncode}: {r.stdout[-400:]}", file=sys.stderr)
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                kn = row["Kernel_Name"]
                if a.match in kn and not (a.exclude and a.exclude in kn):
                    agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
                    if row["Counter_Name"] == counters.split()[0]:
                        names[kn] += 1
    out = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
    if "FETCH_SIZE" in out:
        out["hbm_read_bytes"] = 2.0 * out["FETCH_SIZE"] * 1024.0  # gfx950: 16-B/lane streaming reads are tallied at half
    if "WRITE_SIZE" in out:
        out["hbm_write_bytes"] = out["WRITE_SIZE"] * 1024.0
    if out.get("TCC_HIT_sum") is not None and out.get("TCC_MISS_sum") is not None:
        out["l2_hit_rate"] = out["TCC_HIT_sum"] / max(out["TCC_HIT_sum"] + out["TCC_MISS_sum"], 1.0)
    out["_kernels"] = dict(names)
    out["_label"] = a.label
    out["_command"] = " ".join(a.cmd)
    out["_note"] = ("mean per dispatch of the matching kernel; SQ_* summed over the chip, GRBM_* summed over 8 XCDs; one rocprofv3 --pmc pass per "
                    "counter group (tools/collect_counters.py); MfmaUtil as rocprofv3 derives it (gfx94x formula on gfx950)")
    if a.stats_csv:
        d = os.path.join(tmp, "stats")
        r = subprocess.run([prof, "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "s", "--", *cmd],
                           cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(f, a.stats_csv)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}))


if __name__ == "__main__":
    main()
