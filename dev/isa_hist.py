"""Instruction histogram of one kernel of a device assembly file (developer tool):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only any4_amd/csrc/tg_xr.hip -o /tmp/tg_xr.s
    python dev/isa_hist.py /tmp/tg_xr.s 'w4_gemm_xr_kernel<BF16, 4, 16, 4, 4, false, 8>' [--blocks]
Prints, per basic block (label to label) of the kernel: instruction count and the counts of the classes that matter here
(VALU, v_perm, MFMA, permlane swaps, DS reads / writes, VMEM loads / stores, SALU, s_waitcnt, s_barrier, readlane / writelane =
SGPR spill traffic, scratch)."""
import re
import subprocess
import sys

path, want = sys.argv[1], sys.argv[2]
blocks = "--blocks" in sys.argv
txt = open(path).read().split("\n")
names = {}
for i, line in enumerate(txt):
    m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
    if m:
        names[m.group(1)] = i
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
norm = lambda s: s.replace("(anonymous namespace)::", "")
start = None
for mangled, d in zip(names, dem):
    if want in norm(d):
        start = names[mangled]
        print("kernel:", norm(d))
        break
if start is None:
    sys.exit("kernel not found")
CLASSES = [("mfma", r"v_mfma"), ("swap", r"v_permlane"), ("perm", r"v_perm_b32"), ("dot2", r"v_dot2"), ("rdlane", r"v_readlane|v_readfirstlane"), ("wrlane", r"v_writelane"),
           ("valu", r"v_"), ("ds_rd", r"ds_read|ds_bpermute"), ("ds_wr", r"ds_write"), ("vm_ld", r"global_load|buffer_load|flat_load"),
           ("vm_st", r"global_store|buffer_store|flat_store"), ("scratch", r"scratch_"), ("wait", r"s_waitcnt"), ("barrier", r"s_barrier"), ("salu", r"s_")]
tot, cur, label, out = {}, {}, "entry", []
def flush():
    if cur:
        out.append((label, dict(cur)))
for line in txt[start + 1:]:
    s = line.strip()
    if s.startswith("s_endpgm"):
        break
    m = re.match(r"^(\.LBB\w+):", s)
    if m:
        flush()
        cur, label = {}, m.group(1)
        continue
    if not s or s.startswith((";", ".", "//")):
        continue
    op = s.split()[0]
    for name, pat in CLASSES:
        if re.match(pat, op):
            cur[name] = cur.get(name, 0) + 1
            tot[name] = tot.get(name, 0) + 1
            break
    cur["n"] = cur.get("n", 0) + 1
    tot["n"] = tot.get("n", 0) + 1
flush()
keys = ["n"] + [c for c, _ in CLASSES]
if blocks:
    for lab, c in out:
        if c.get("n", 0) >= 20:
            print(f"{lab:14s} " + " ".join(f"{k}={c[k]}" for k in keys if c.get(k)))
print("total          " + " ".join(f"{k}={tot[k]}" for k in keys if tot.get(k)))
