"""Compile the gfx950 C-ABI library in-tree:  python -m any4_amd.build  [--force] [-v] [-j N]

hipcc cross-compiles without a GPU.  The library is one translation unit per kernel family (any4_amd/csrc/tg_common.cuh lists
them), compiled in parallel and linked into any4_amd/lib/libtinygemm_hip.so -- git-ignored, but it travels with the working
tree (it is what the GPU box loads).  Objects are cached under any4_amd/lib/obj/ and rebuilt when a source they include is newer.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
OUT_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
OUT = os.path.join(OUT_DIR, "libtinygemm_hip.so")
ARCH = "gfx950"

# (object name, source, extra defines) -- longest first, so that the pool starts them first
UNITS = [
    ("tg_pair_bf16", "tg_pair.hip", []),
    ("tg_pair_f16", "tg_pair.hip", ["-DTG_TU_F16"]),
    ("tg_stream_bf16", "tg_stream.hip", []),
    ("tg_stream_f16", "tg_stream.hip", ["-DTG_TU_F16"]),
    ("tg_xr", "tg_xr.hip", []),
    ("tg_pair16", "tg_pair16.hip", []),
    ("tg_splitk", "tg_splitk.hip", []),
    ("tg_gemv", "tg_gemv.hip", []),
    ("tg_tile", "tg_tile.hip", []),
    ("tinygemm_hip", "tinygemm_hip.hip", []),
]
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-comment", "-Wno-int-to-pointer-cast"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=...)")


_INC_RE = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _deps(path: str, seen=None) -> set:
    """The file and every quoted include below it (resolved next to the including file)."""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path) as f:
        for inc in _INC_RE.findall(f.read()):
            _deps(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def _stale(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _obj(name: str) -> str:
    return os.path.join(OBJ_DIR, name + ".o")


def needs_build() -> bool:
    return any(_stale(_obj(n), _deps(os.path.join(CSRC, s)) | {__file__}) for n, s, _ in UNITS) or \
        _stale(OUT, [_obj(n) for n, _, _ in UNITS])


def build(force: bool = False, verbose: bool = False, jobs: int | None = None, extra_flags=(), out: str | None = None,
          obj_dir: str | None = None) -> str:
    """extra_flags / out / obj_dir: developer variant builds (dev/build_variant.sh): every unit gets the -D flags."""
    out = out or OUT
    odir = obj_dir or OBJ_DIR
    if not force and not extra_flags and not needs_build():
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    os.makedirs(odir, exist_ok=True)
    cc = hipcc()
    todo = []
    for name, src, defs in UNITS:
        obj = os.path.join(odir, name + ".o")
        srcp = os.path.join(CSRC, src)
        if force or extra_flags or _stale(obj, _deps(srcp) | {__file__}):
            cmd = [cc, *FLAGS, *defs, *extra_flags, "-c", srcp, "-o", obj + ".tmp"]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            todo.append((cmd, obj))

    def run(job):
        cmd, obj = job
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(obj + ".tmp", obj)

    jobs = jobs or min(len(todo) or 1, os.cpu_count() or 1, 8)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(run, todo))
    link = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fvisibility=hidden", *[os.path.join(odir, n + ".o") for n, _, _ in UNITS],
            "-o", out + ".tmp"]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    j = int(sys.argv[sys.argv.index("-j") + 1]) if "-j" in sys.argv else None
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, jobs=j))
