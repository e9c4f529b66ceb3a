#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the REFERENCE's Python.

Run in the build container only (it needs /root/reference; the GPU box has no copy):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is captured (inputs + the reference's own outputs; data only, no reference source):

  group_quant.npz   tinygemm_lib/utils.py:27-67 group_quantize_tensor on seeded bf16 weights,
                    g in {32,64,128,256}: codes + scales_and_zeros bit patterns.           (Q1)
  group_quant_int8.npz  the same function at n_bit = 8 (inputs of the int8 kernels, row N3).
  mx4.npz           tinygemm_lib/utils.py:137-232 quantize_mx4 / dequantize_mx4, g=32.       (Q2)
  any4_n1024_k1024_g128_seed1234.npz
                    quantize.py:523-637 anyq_quantize_tensor (sklearn k-means, per-row LUT)
                    on W = randn*0.02 -> bf16, then the reference's own CPU dequant-matmul
                    y = x @ anyq_dequantize_tensor(...)^T in bf16.  BASELINE config 1.      (H-Q)
  anyq_linspace64.npz
                    tests/test_anyq.py:63-108 inputs: 64x64 permutations of linspace(-8,7),
                    per_row=False, g in {32,64}: codes, lut, scales_and_zeros, x, y_ref.    (7)

The reference imports `bitsandbytes` at module scope (quantize.py); an EMPTY stub package
is put on PYTHONPATH for the import to succeed (nothing from it is called on this path).
"""
import hashlib
import os
import sys
import tempfile

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

REF = os.environ.get("ANY4_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def _bits16(t):
    import torch

    return t.contiguous().view(torch.int16).numpy().view("uint16").copy()


def main():
    import numpy as np
    import torch

    stub = tempfile.mkdtemp(prefix="any4_stub_")
    os.makedirs(os.path.join(stub, "bitsandbytes"))
    open(os.path.join(stub, "bitsandbytes", "__init__.py"), "w").close()
    # joblib/loky workers re-import `quantize`, so the stub must be on PYTHONPATH, not just sys.path
    os.environ["PYTHONPATH"] = os.pathsep.join([stub, REF, os.environ.get("PYTHONPATH", "")])
    sys.path[:0] = [stub, REF]

    # mx quantiser imports are gated on torch.version.cuda (tinygemm_lib/utils.py:13)
    if not torch.version.cuda:
        torch.version.cuda = "shim"
    import tinygemm_lib.utils as ref_utils  # noqa: E402  (the reference's)

    assert ref_utils.__file__.startswith(REF), ref_utils.__file__

    # ---------------------------------------------------------------- group_quant.npz (Q1)
    out = {}
    g0 = torch.Generator().manual_seed(20250718)
    w = (torch.randn(40, 512, generator=g0) * 0.05).to(torch.bfloat16)
    out["w_bits"] = _bits16(w)
    for g in (32, 64, 128, 256):
        codes, sz = ref_utils.group_quantize_tensor(w, 4, g)
        out[f"codes_g{g}"] = codes.numpy().astype("uint8")
        out[f"sz_bits_g{g}"] = _bits16(sz)
    # the identity known-answer case of the kernel tests (test_tinygemm_any4.py:14-37)
    eye = torch.eye(256, dtype=torch.bfloat16)
    codes, sz = ref_utils.group_quantize_tensor(eye, 4, 64)
    out["eye256_codes_g64"] = codes.numpy().astype("uint8")
    out["eye256_sz_bits_g64"] = _bits16(sz)
    np.savez_compressed(os.path.join(HERE, "group_quant.npz"), **out)

    # ---------------------------------------------------------------- group_quant_int8.npz (Q1 at n_bit = 8, row N3)
    out8 = {"w_bits": _bits16(w)}
    for g in (32, 128):
        codes, sz = ref_utils.group_quantize_tensor(w, 8, g)
        out8[f"codes_g{g}"] = codes.numpy().astype("uint8")
        out8[f"sz_bits_g{g}"] = _bits16(sz)
    codes, sz = ref_utils.group_quantize_tensor(eye, 8, 64)  # identity case of test_tinygemm_int8.py:23-50
    out8["eye256_codes_g64"] = codes.numpy().astype("uint8")
    out8["eye256_sz_bits_g64"] = _bits16(sz)
    np.savez_compressed(os.path.join(HERE, "group_quant_int8.npz"), **out8)

    # ---------------------------------------------------------------- mx4.npz (Q2)
    out = {}
    g1 = torch.Generator().manual_seed(4)
    w = torch.randn(24, 256, generator=g1) * torch.logspace(-3, 3, 24).unsqueeze(1)
    w[3, :32] = 0.0
    q, e = ref_utils.quantize_mx4(w, 32)
    deq = ref_utils.dequantize_mx4(q, e)
    out["w"] = w.numpy()
    out["q"] = q.numpy().astype("uint8")
    out["e"] = e.numpy()
    out["deq"] = deq.numpy()
    # identity case used by test_tinygemm_mx4.py:14-39
    q, e = ref_utils.quantize_mx4(torch.eye(128), 32)
    out["eye128_q"] = q.numpy().astype("uint8")
    out["eye128_e"] = e.numpy()
    np.savez_compressed(os.path.join(HERE, "mx4.npz"), **out)

    # ---------------------------------------------------------------- H-Q fixture
    import quantize as ref_quantize  # noqa: E402

    assert ref_quantize.__file__.startswith(REF)
    n = k = 1024
    g = 128
    torch.manual_seed(1234)
    W = (torch.randn(n, k) * 0.02).to(torch.bfloat16)
    x = torch.randn(1, k).to(torch.bfloat16)
    codes, lut, sz = ref_quantize.anyq_quantize_tensor(W, n_bit=4, q_group_size=g, per_row=True)
    Wdeq = ref_quantize.anyq_dequantize_tensor(codes, lut, sz, n_bit=4, q_group_size=g, per_row=True)
    assert Wdeq.dtype == torch.bfloat16 and lut.dtype == torch.bfloat16
    y = x @ Wdeq.t()
    c8 = codes.to(torch.uint8).numpy()
    assert c8.max() <= 15
    out = {
        "n": n, "k": k, "g": g,
        "x_bits": _bits16(x),
        "codes_nib": (c8[:, 0::2] | (c8[:, 1::2] << 4)).astype("uint8"),  # low nibble = even k
        "lut_bits": _bits16(lut),            # [n,16], in the [0,15]-scaled domain (module gets lut-8)
        "lut_m8_bits": _bits16(lut - 8),     # what quantize.py:893 hands to Any4Linear
        "sz_bits": _bits16(sz),              # [k/g, n, 2]
        "wdeq_rows0_8_bits": _bits16(Wdeq[:8]),
        "wdeq_sha256": np.frombuffer(hashlib.sha256(_bits16(Wdeq).tobytes()).digest(), dtype="uint8"),
        "y_bits": _bits16(y),
        "y_f32_from_wdeq": (x.float() @ Wdeq.float().t()).numpy(),
    }
    np.savez_compressed(os.path.join(HERE, "any4_n1024_k1024_g128_seed1234.npz"), **out)

    # ---------------------------------------------------------------- test_anyq.py:63-108 inputs
    out = {}
    for dtype, tag in ((torch.bfloat16, "bf16"),):
        for g in (32, 64):
            torch.manual_seed(7 + g)
            w_vals = torch.linspace(start=-8, end=7, steps=16, dtype=dtype)
            idx = torch.stack([torch.randperm(16) for _ in range(64 * 64 // 16)]).view(64, 64)
            w = w_vals[idx]
            x = torch.randn(29, 64, dtype=dtype)
            y_ref = x @ w.t()
            codes, lut, sz = ref_quantize.anyq_quantize_tensor(
                w, n_bit=4, q_group_size=g, new_grouping=False, zero_point=True, per_row=False)
            lut = lut - 8
            out[f"{tag}_g{g}_w_bits"] = _bits16(w)
            out[f"{tag}_g{g}_x_bits"] = _bits16(x)
            out[f"{tag}_g{g}_y_bits"] = _bits16(y_ref)
            out[f"{tag}_g{g}_codes"] = codes.numpy().astype("uint8")
            out[f"{tag}_g{g}_lut_bits"] = _bits16(lut)
            out[f"{tag}_g{g}_sz_bits"] = _bits16(sz)
    np.savez_compressed(os.path.join(HERE, "anyq_linspace64.npz"), **out)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
