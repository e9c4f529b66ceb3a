#!/usr/bin/env python3
"""Per-node timeline of a decode step from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d OUT -o dec -- python tools/llama_decode_bench.py --steps 20 --warmup 5
    python tools/decode_timeline.py OUT/**/dec_kernel_trace.csv [--last 10]

Takes the last `--last` graph replays (a replay = the kernels between two LM-head GEMMs), and prints for every position
of the replay's kernel sequence that repeats per layer: kernel name, average duration (End - Start) and the average gap
to the previous kernel's End.  rocprofv3's Start is when the first wave is dispatched, so `gap` is the dependent-node
boundary and `dur` contains the kernel's own latency chain.
"""
import argparse
import csv
import glob
import sys
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--last", type=int, default=10, help="graph replays to average over")
    ap.add_argument("--period", type=int, default=0, help="kernels per layer (0: find the shortest repeating period)")
    a = ap.parse_args()
    paths = glob.glob(a.trace, recursive=True)
    if not paths:
        sys.exit(f"no file matches {a.trace}")
    rows = []
    with open(paths[0]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    names = [short(r[2]) for r in rows]
    # a replay ends with the LM-head GEMM: the longest kernel by far
    durs = [r[1] - r[0] for r in rows]
    big = max(durs[-2000:]) * 0.6
    ends = [i for i, d in enumerate(durs) if d >= big]
    if len(ends) < a.last + 1:
        sys.exit(f"only {len(ends)} replays found")
    ends = ends[-(a.last + 1):]
    agg = defaultdict(lambda: [0, 0, 0])
    seqs = []
    for e0, e1 in zip(ends[:-1], ends[1:]):
        seqs.append(list(range(e0 + 1, e1 + 1)))
    n = min(len(s) for s in seqs)
    total = 0
    for s in seqs:
        total += rows[s[-1]][1] - rows[s[0]][0]
        for j, i in enumerate(s[:n]):
            g = rows[i][0] - rows[i - 1][1]
            k = agg[j]
            k[0] += durs[i]
            k[1] += g
            k[2] += 1
    print(f"{len(seqs)} replays, {n} kernels per replay, first-start -> last-end {total / len(seqs) / 1e3:.1f} us")
    seq_names = [names[i] for i in seqs[-1][:n]]
    period = a.period or find_period(seq_names)
    print(f"period {period} kernels")
    # average the layers' positions (skip the first and last period: embedding / final norm / LM head)
    per = defaultdict(lambda: [0.0, 0.0, 0])
    body = range(period, n - period - (n % period))
    for j in body:
        p = per[(j % period, seq_names[j])]
        p[0] += agg[j][0] / agg[j][2]
        p[1] += agg[j][1] / agg[j][2]
        p[2] += 1
    tot_d = tot_g = 0.0
    for (pos, nm), (d, g, c) in sorted(per.items()):
        print(f"  [{pos}] {nm:60s} dur {d / c / 1e3:7.2f} us   gap before {g / c / 1e3:6.2f} us   ({c} layers)")
        tot_d += d / c
        tot_g += g / c
    print(f"  per layer: kernels {tot_d / 1e3:.2f} us + gaps {tot_g / 1e3:.2f} us = {(tot_d + tot_g) / 1e3:.2f} us")
    print("  tail of the replay:")
    for j in range(max(0, n - 4), n):
        print(f"      {seq_names[j]:60s} dur {agg[j][0] / agg[j][2] / 1e3:7.2f} us   gap before {agg[j][1] / agg[j][2] / 1e3:6.2f} us")


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) <= 60 else name[:57] + "..."


def find_period(seq):
    body = seq[len(seq) // 4: 3 * len(seq) // 4]
    for p in range(1, 40):
        if all(body[i] == body[i + p] for i in range(len(body) - p)):
            return p
    return 1


if __name__ == "__main__":
    main()
