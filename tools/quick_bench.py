"""Developer micro-benchmark (not the judged bench.py): times tg_gemm_w4 through the C ABI with
cold weights (rotating over > 288 MiB of distinct matrices) in three ways:
  eager   : back-to-back launches on one stream, HIP-event time / launches
  graph   : the same launches captured in one hipGraph
  stacked : ONE launch over all matrices (grid.z = L)
"""
import argparse
import ctypes
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from any4_amd import _lib


def alg_bytes(m, n, k, g, qtype):
    lut = {"any4_rowwise": 32 * n, "any4_global": 32, "int4": 0, "mx4": 0, "int8": 0}[qtype]
    q = n * k // 32 if qtype == "mx4" else (k // g) * n * 4
    return (n * k if qtype == "int8" else n * k // 2) + q + lut + m * k * 2 + m * n * 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,4096,4096,1;8,4096,4096,1;8,8192,8192,0;16,4096,4096,1;1,4096,4096,0")
    ap.add_argument("--L", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--settle", type=float, default=0.15, help="seconds of continuous stacked launches before the 'steady' timing")
    ap.add_argument("--qtype", default="any4_rowwise")
    ap.add_argument("--g", type=int, default=128)
    ap.add_argument("--inner", type=int, default=4)
    ap.add_argument("--numerics", default="fast", choices=["fast", "reference"])
    a = ap.parse_args()
    L_ = _lib.load()
    dev = "cuda:0"
    T = None
    import tinygemm  # noqa
    T = torch.ops.tinygemm
    qt = {"int4": 0, "any4_global": 1, "any4_rowwise": 2, "mx4": 3, "int8": 4}[a.qtype]
    gemm = L_.tg_gemm_w8 if a.qtype == "int8" else L_.tg_gemm_w4
    for cfg in a.configs.split(";"):
        m, n, k, on_right = [int(v) for v in cfg.split(",")]
        L = a.L if n * k <= 4096 * 4096 else max(a.L // 4, 12)
        g = 32 if a.qtype == "mx4" else a.g
        gen = torch.Generator(device=dev).manual_seed(0)
        inner = a.inner
        if a.qtype == "int8":
            inner = min(inner, 4 if on_right else 2)
            shape = (L, n // 8, k // (16 * inner), 32, inner) if on_right else (L, n // 16, k // (16 * inner), 32, 2 * inner)
        elif on_right:
            shape = (L, n // 8, k // (16 * inner), 32, inner // 2)
        else:
            shape = (L, n // 16, k // (16 * inner), 32, inner)
        w = torch.randint(-2**31, 2**31 - 1, shape, dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
        x = torch.randn(L, m, k, device=dev, generator=gen).bfloat16()
        if os.environ.get("QB_XZERO"):  # power experiment: only activation row 0 carries data
            x[:, 1:] = 0
        if a.qtype == "mx4":
            q = torch.randint(120, 131, (L, n, k // g), dtype=torch.uint8, device=dev, generator=gen)
            qstride = q.stride(0)
        else:
            q = (torch.rand(L, k // g, n, 2, device=dev, generator=gen) * 0.02).bfloat16()
            qstride = q.stride(0) * 2
        lut = torch.randn(L, n, 16, device=dev, generator=gen).bfloat16() if a.qtype == "any4_rowwise" else torch.randn(L, 16, device=dev, generator=gen).bfloat16()
        y = torch.empty(L, m, n, device=dev, dtype=torch.bfloat16)

        def mk(b, batch):
            return _lib.W4Gemm(x=x[b].data_ptr(), w=w[b].data_ptr(), qinfo=q[b].data_ptr(),
                               lut=(lut[b].data_ptr() if a.qtype.startswith("any4") else None), y=y[b].data_ptr(),
                               m=m, wrows=n, k=k, group=g, qtype=qt, dtype=0, w_on_right=on_right, inner_k_tiles=inner,
                               batch=batch, stride_x=x.stride(0) * 2, stride_w=w.stride(0) * 4, stride_qinfo=qstride,
                               stride_lut=lut.stride(0) * 2, stride_y=y.stride(0) * 2,
                               numerics=(0 if a.numerics == "fast" else 1))

        singles = [mk(b, 1) for b in range(L)]
        stacked = mk(0, L)
        ws = None
        if a.qtype != "int8" and not os.environ.get("QB_NO_WS"):
            need = L_.tg_gemm_w4_workspace_bytes(ctypes.byref(stacked))
            assert need >= 0, need
            if need:
                ws = torch.empty(need, dtype=torch.uint8, device=dev)
                stacked.workspace, stacked.workspace_bytes = ws.data_ptr(), need
        plan = L_.tg_gemm_w4_plan(ctypes.byref(stacked), 0) if a.qtype != "int8" else 0

        def run_eager():
            for s in singles:
                rc = gemm(ctypes.byref(s), 0, torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc

        def run_stacked():
            rc = gemm(ctypes.byref(stacked), 0, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc

        def timeit(fn, iters):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / iters  # us

        t_eager = timeit(run_eager, a.iters) / L
        run_stacked()
        y_st = y.clone()
        run_eager()
        torch.cuda.synchronize()
        same = torch.equal(y_st, y)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            run_eager()
        t_graph = timeit(gr.replay, a.iters) / L
        t_stack = timeit(run_stacked, a.iters) / L
        # steady state: the power controller settles after ~30-60 ms of continuous load
        t0 = __import__("time").perf_counter()
        while __import__("time").perf_counter() - t0 < a.settle:
            for _ in range(20):
                run_stacked()
            torch.cuda.synchronize()
        t_steady = timeit(run_stacked, max(a.iters, 20)) / L
        B = alg_bytes(m, n, k, g, a.qtype)
        print(f"m={m} n={n} k={k} on_right={on_right} {a.qtype} g={g} I={inner} L={L} bytes={B} stacked==eager:{same} plan={plan} ws={0 if ws is None else ws.numel()}")
        for name, t in (("eager", t_eager), ("graph", t_graph), ("stacked", t_stack), ("steady", t_steady)):
            print(f"   {name:8s} {t:8.2f} us/matrix  {B / t / 1e6:8.3f} TB/s  {B / t / 1e6 / 8.0 * 100:5.1f}% of 8 TB/s")


if __name__ == "__main__":
    main()
