"""CPU emulation of the index logic of any4_amd/csrc/w4_gemm_pair.cuh (B side): lane -> packed words -> table bytes ->
MFMA operand slots, activation staging order, per-group scaling, split-K slices.  Checks the mapping against the oracle's
dequant + matmul without a GPU.  Developer tool (imports oracle/, so it is test infrastructure, not product)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc


def bf16_round(a):
    return orc.bf16_to_f32(orc.bf16_bits(a.astype(np.float32)))


def emulate(codes, lut_f, sz_f, x_f, g, I, exact, m):
    n, k = codes.shape
    packed = orc.pack_Bint4(codes, I).reshape(-1)  # [n/8][k/(16I)][32][I/2]
    ksuper = k // (16 * I)
    TILES = 1 if exact else 2
    RW = 32 * TILES
    CPS = I // 2
    gch = g // 32
    nsg = max(1, g // (16 * I))
    spw = ((ksuper // nsg + 7) // 8) * nsg
    nch = k // 32
    # staged X: [a][chunk][q][8]
    xs = np.zeros((m, nch, 4, 8), np.float64)
    for a in range(m):
        for ch in range(nch):
            xx = x_f[a, 32 * ch:32 * ch + 32]
            for q in range(4):
                xs[a, ch, q] = [xx[2 * q], xx[2 * q + 8], xx[2 * q + 16], xx[2 * q + 24], xx[2 * q + 1], xx[2 * q + 9], xx[2 * q + 17], xx[2 * q + 25]]
    xsum = x_f.reshape(m, k // g, g).sum(axis=2)  # [a][group]
    y = np.zeros((m, n), np.float64)
    for row0 in range(0, n, RW):
        part = np.zeros((8, TILES, m, 32), np.float64)
        for wave in range(8):
            s_begin, s_end = wave * spw, min(wave * spw + spw, ksuper)
            for t in range(TILES):
                for c in range(32):
                    row = min(row0 + t * 32 + c, n - 1)
                    acc = np.zeros(m)
                    yacc = np.zeros(m)
                    for s in range(s_begin, s_end):
                        words = {}
                        for h in range(2):
                            off_words = ((row >> 3) * ksuper * 32 + 4 * (row & 7) + 2 * h) * (I // 2) + s * 32 * (I // 2)
                            words[h] = packed[off_words:off_words + I].astype(np.uint32)
                        for jc in range(CPS):
                            chunk = s * CPS + jc
                            grp = (chunk * 32) // g
                            sc, zr = sz_f[grp, row, 0], sz_f[grp, row, 1]
                            for qq in range(2):
                                for h in range(2):
                                    w = int(words[h][qq * CPS + jc])
                                    for j in range(4):
                                        byte = (w >> (8 * j)) & 0xff
                                        lo, hi = lut_f[row, byte & 15], lut_f[row, byte >> 4]
                                        if exact:
                                            lo = float(bf16_round(np.array([np.float32(np.float32(lo) * np.float32(sc) + np.float32(zr))]))[0]) if False else lo
                                        xa = xs[:, chunk, 2 * h + qq, 2 * j:2 * j + 2]  # [m][2]
                                        if exact:
                                            w0 = float(bf16_round(np.array([lo * sc + zr]))[0])
                                            w1 = float(bf16_round(np.array([hi * sc + zr]))[0])
                                            acc += xa[:, 0] * w0 + xa[:, 1] * w1
                                        else:
                                            acc += xa[:, 0] * lo + xa[:, 1] * hi
                            if not exact and (chunk & (gch - 1)) == gch - 1:
                                yacc += sc * acc + zr * xsum[:, grp]
                                acc[:] = 0
                    part[wave, t, :, c] = acc if exact else yacc
        for t in range(TILES):
            for c in range(32):
                row = row0 + t * 32 + c
                if row < n:
                    y[:, row] = part[:, t, :, c].sum(axis=0)
    return y


def main():
    rng = np.random.default_rng(0)
    for (n, k, g, I, m) in [(64, 512, 128, 4, 1), (40, 256, 32, 4, 2), (64, 512, 64, 8, 1), (32, 256, 128, 2, 3), (64, 1024, 256, 4, 1)]:
        codes = rng.integers(0, 16, (n, k), dtype=np.int32)
        lut_b = orc.bf16_bits(rng.standard_normal((n, 16)).astype(np.float32))
        sz_b = orc.bf16_bits((rng.random((k // g, n, 2)) * 0.02 + 0.005).astype(np.float32))
        x_b = orc.bf16_bits(rng.standard_normal((m, k)).astype(np.float32))
        lut_f, sz_f, x_f = [orc.bf16_to_f32(v).astype(np.float64) for v in (lut_b, sz_b, x_b)]
        wq = orc.bf16_to_f32(orc.dequant(codes, g, orc.Q_ANY4_ROWWISE, sz_b, lut_b)).astype(np.float64)
        y_exact_ref = x_f @ wq.T
        s = np.repeat(sz_f[:, :, 0].T, g, axis=1); z = np.repeat(sz_f[:, :, 1].T, g, axis=1)
        w_unrounded = np.take_along_axis(lut_f, codes.astype(np.int64), axis=1) * s + z
        y_gs_ref = x_f @ w_unrounded.T
        for exact in (True, False):
            y = emulate(codes, lut_f, sz_f, x_f, g, I, exact, m)
            ref = y_exact_ref if exact else y_gs_ref
            err = np.abs(y - ref).max()
            print(f"n={n} k={k} g={g} I={I} m={m} exact={exact}: max|emul - ref| = {err:.3e}  (max|y| {np.abs(ref).max():.3f})")
            assert err < 1e-9, "mapping error"


if __name__ == "__main__":
    main()
