#!/usr/bin/env python3
"""The side formats of the reference's op surface against nn.Linear (bf16), one 4096 x 4096 layer per hipGraph node over six
distinct layers: Int8Linear (the 16-row kernel up to 16 rows, the tile GEMM's int8 flavour from 17) and the 16-bit-weight op
tinygemm_y_f16RM_x_f16RM_w_f16TC.   python tools/side_formats_bench.py [--rows 1,4,16,64,128,512,2048]"""
import argparse
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dev"))
import torch  # noqa: E402


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="1,4,16,64,128,512,2048")
    ap.add_argument("--f16-rows", default="1,8,16,64")
    a = ap.parse_args()
    import tinygemm  # noqa: F401  (registers the ops)
    from any4_amd import quantize as Q
    from many_rows_bench import graph_time

    T = torch.ops.tinygemm
    k = n = 4096
    lins = [torch.nn.Linear(k, n, dtype=torch.bfloat16, device="cuda", bias=False) for _ in range(6)]
    q = Q.intq_layer(torch.nn.Linear(k, n, dtype=torch.bfloat16, device="cuda", bias=False), n_bit=8)
    mods = [copy.deepcopy(q) for _ in range(6)]
    for m in (int(v) for v in a.rows.split(",") if v):
        x = torch.randn(m, k, dtype=torch.bfloat16, device="cuda") * 0.05
        print(f"int8 m={m}: nn.Linear {graph_time(lins, x):.2f} us  Int8Linear {graph_time(mods, x):.2f} us", flush=True)
    w = torch.randn(n, k, dtype=torch.bfloat16, device="cuda")
    for m in (int(v) for v in a.f16_rows.split(",") if v):
        x = torch.randn(m, k, dtype=torch.bfloat16, device="cuda") * 0.05
        for side, inner in ((True, 2), (True, 1), (False, 1)):
            wp = T.convert_matrix_to_m16n8k16_B_layout(w, inner) if side else T.convert_matrix_to_m16n8k16_A_layout(w, 1)
            ws = [wp.clone() for _ in range(6)]
            fns = [(lambda xx, wi=wi: T.tinygemm_y_f16RM_x_f16RM_w_f16TC(xx, wi, True) if side else T.tinygemm_y_f16RM_x_f16RM_w_f16TC(wi, xx, False)) for wi in ws]
            print(f"f16 weights m={m} side={'B' if side else 'A'} I={inner}: nn.Linear {graph_time(lins, x):.2f} us  tinygemm f16TC {graph_time(fns, x):.2f} us", flush=True)


if __name__ == "__main__":
    main()
