"""Developer probe: does the m = 8 steady time depend on where the process's buffers land?  Prints base addresses (mod 2 MiB)
next to the steady time; run several times (one process each)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from any4_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
pad = int(os.environ.get("PROBE_PAD", "0"))
junk = torch.empty(pad, dtype=torch.uint8, device=dev) if pad else None
m, n, k, g, L = int(os.environ.get("PROBE_M", "8")), 4096, 4096, 128, 512
w, x, q, lut, y = bench.make_batch(L, m, n, k, g, 4, dev, 7, "any4_rowwise", True)
aa = bench.make_args(_lib, w, x, q, lut, y, m, n, k, g, "any4_rowwise", True, 4, L, "fast")
ws = bench.attach_workspace(lib, aa, dev)
def launch():
    _lib.check(lib.tg_gemm_w4(ctypes.byref(aa), 0, st.cuda_stream), "tg_gemm_w4")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.7:
    for _ in range(20): launch()
    torch.cuda.synchronize()
best = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(40): launch()
    e1.record(st); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 40 / L * 1e3)
M = 2 << 20
f = lambda t: f"{t.data_ptr():#x}(+{t.data_ptr() % M:#x})"
print(f"m={m} pad={pad} {sorted(best)[2]:.3f} us/layer  w {f(w)} x {f(x)} q {f(q)} lut {f(lut)} y {f(y)} ws {f(ws) if ws is not None else None}", flush=True)
