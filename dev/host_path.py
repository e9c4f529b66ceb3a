#!/usr/bin/env python3
"""Where the eager forward of a quantized module spends its host time (developer tool): wall time per call, whether the recorded
launch plan is in use, and a cProfile of 2000 forwards.   python dev/host_path.py [--k 4096] [--profile]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    from any4_amd import quantize as Q

    k = a.k
    x = torch.randn(1, k, dtype=torch.bfloat16, device="cuda")
    lin = torch.nn.Linear(k, k, dtype=torch.bfloat16, device="cuda", bias=False)
    mods = {"nn.Linear": lin, "anyq": Q.anyq_layer(torch.nn.Linear(k, k, dtype=torch.bfloat16, device="cuda", bias=False), pseudo=False),
            "anyq per_row=False": Q.anyq_layer(torch.nn.Linear(k, k, dtype=torch.bfloat16, device="cuda", bias=False), pseudo=False, per_row=False),
            "intq": Q.intq_layer(torch.nn.Linear(k, k, dtype=torch.bfloat16, device="cuda", bias=False), pseudo=False)}
    for name, m in mods.items():
        for _ in range(50):
            m(x)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(1000):
                m(x)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 1000 * 1e6)
        # host-only: the same calls while the GPU is kept far behind is the same thing here (launches are asynchronous)
        plan = m.__dict__.get("_plan")
        info = "" if name == "nn.Linear" else f"  plan={'yes' if plan is not None else 'NO'}  params aligned16: " + \
            str({n: (p.data_ptr() % 16 == 0) for n, p in m.named_parameters()})
        print(f"{name:22s} {best:7.2f} us per forward (wall, 1000 calls){info}")
        if a.profile and name != "nn.Linear":
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(2000):
                m(x)
            pr.disable()
            torch.cuda.synchronize()
            st = pstats.Stats(pr)
            st.sort_stats("tottime").print_stats(8)


if __name__ == "__main__":
    main()
