"""One stacked tg_gemm_w4 launch per shape, checked against the CPU oracle and timed in the steady state: the command the counter
passes of tools/gpu_round.sh profile, and the unit of a same-box A/B (dev/exp.sh).
    python tools/ab.py [m,n,k,on_right,qtype,g[,L]] ...      AB_HOLD=<s>: seconds of continuous load before the samples (the chip
    is power-capped: the first second after idle runs at a higher clock than the sustained state, dev/README.md)"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from any4_amd import _lib, ops

lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
for cfg in sys.argv[1:] or ["1,4096,4096,1,any4_rowwise,128"]:
    f = cfg.split(",")
    m, n, k, on_right, qtype, g = int(f[0]), int(f[1]), int(f[2]), int(f[3]) == 1, f[4], int(f[5])
    L = int(f[6]) if len(f) > 6 else (512 if n * k <= 4096 * 4096 else 128)
    w, x, q, lut, y = bench.make_batch(L, m, n, k, g, 4, dev, 7, qtype, on_right)
    num = os.environ.get("ANY4_AB_NUMERICS", "fast")
    aa = bench.make_args(_lib, w, x, q, lut, y, m, n, k, g, qtype, on_right, 4, L, num)
    ws = bench.attach_workspace(lib, aa, dev)
    # (weights on the left: the tensor holds the reference's Aint4 words here -- the native format runs the B-side kernels, which the
    #  on_right = 1 configurations already cover)
    plan = ops.gemm_w4_plan(m, n, k, g, bench.QT[qtype], on_right, 4, torch.bfloat16, L, num, weight_format="reference")

    def launch():
        _lib.check(lib.tg_gemm_w4(ctypes.byref(aa), 0, st.cuda_stream), "tg_gemm_w4")

    bench.calibrate_x(launch, x, y)  # max|y| into (0.95, 1.9]: where bench.check_layers' raw 1e-2 contract is stated

    y.fill_(float("nan"))
    launch()
    torch.cuda.synchronize()
    try:
        err = bench.check_layers(w, x, q, lut, y, g, qtype, on_right, 4, plan, layers=(0, L // 2, -1), rows=128)
        ok = f"ok max|err| {err['max_abs_err_vs_kernel_formula']:.3e} (vs reference {err['max_abs_err_vs_reference']:.3e})"
    except SystemExit as e:
        ok = f"MISMATCH {e}"
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.7:
        for _ in range(20):
            launch()
        torch.cuda.synchronize()
    hold = float(os.environ.get("AB_HOLD", "0"))  # seconds of continuous load before the samples (power / clock probes)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < hold:
        for _ in range(20):
            launch()
        torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(40):
            launch()
        e1.record(st)
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 40 / L * 1e3)
    us = sorted(best)[2]
    by = bench.alg_bytes(m, n, k, g, qtype)
    print(f"{cfg:40s} plan={plan} {ok}  {us:.3f} us/layer  {by / us / 1e3 / 8000 * 100:.1f}% (min {min(best):.3f} max {max(best):.3f})", flush=True)
