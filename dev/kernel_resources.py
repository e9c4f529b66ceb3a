"""Registers / scratch of every kernel of one translation unit (developer tool):
    python dev/kernel_resources.py tg_pair16.hip [substring of the demangled name] [-- extra hipcc flags]
Compiles the unit with -Rpass-analysis=kernel-resource-usage (any4_amd/build.py's flags) and prints one line per kernel."""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from any4_amd import build as b  # noqa: E402

args = sys.argv[1:]
extra = []
if "--" in args:
    extra = args[args.index("--") + 1:]
    args = args[:args.index("--")]
unit = args[0]
filt = args[1] if len(args) > 1 else ""
defs = next((list(d) for _, s, d in b.UNITS if s == unit), [])
cmd = [b.hipcc(), "-Rpass-analysis=kernel-resource-usage", *b.FLAGS, *defs, *extra, "-c", os.path.join(b.CSRC, unit), "-o", "/tmp/kernel_resources.o"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in txt.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark: +([A-Za-z ]+?(?: \[[^\]]*\])?): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1)] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, names):
    d = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"\((anonymous namespace)?[^()]*Params\)$", "", d))
    if filt in d:
        print(f"{d[:120]:120s} VGPR {r.get('VGPRs'):4d} AGPR {r.get('AGPRs'):3d} scratch {r.get('ScratchSize [bytes/lane]'):4d} occupancy {r.get('Occupancy [waves/SIMD]')}")
