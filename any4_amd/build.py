"""Compile the gfx950 C-ABI library in-tree:  python -m any4_amd.build

hipcc cross-compiles without a GPU.  The resulting any4_amd/lib/libtinygemm_hip.so is
git-ignored but travels with the working tree (it is what the GPU box loads).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "tinygemm_hip.hip")
HDR = os.path.join(os.path.dirname(HERE), "include", "tinygemm_hip.h")
OUT_DIR = os.path.join(HERE, "lib")
OUT = os.path.join(OUT_DIR, "libtinygemm_hip.so")
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def needs_build() -> bool:  # noqa
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    srcs = [HDR] + [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))]
    return any(os.path.getmtime(p) > t for p in srcs)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-Wno-comment", "-Wno-int-to-pointer-cast", SRC, "-o", OUT + ".tmp"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
