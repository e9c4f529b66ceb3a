"""numpy front-end of the CPU oracle (oracle/tinygemm_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of tinygemm_oracle.c.  The product path
(any4_amd/, tinygemm/, tinygemm_lib/, modules.py) never imports this module; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.

16-bit float matrices are passed as ``np.uint16`` bit patterns (``dtype`` says whether
they are bf16 or fp16) so that no rounding happens on the Python side.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtinygemm_oracle.so")

BF16, F16 = 0, 1
Q_INT4, Q_ANY4_GLOBAL, Q_ANY4_ROWWISE, Q_MX4, Q_INT8 = 0, 1, 2, 3, 4


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (seconds).  Idempotent."""
    src = os.path.join(_HERE, "tinygemm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"] if force else ["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        i64, i32, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
        L.tgo_pack_Bint4.argtypes = [vp, i64, i64, i32, vp]
        L.tgo_unpack_Bint4.argtypes = [vp, i64, i64, i32, vp]
        L.tgo_pack_Aint4.argtypes = [vp, i64, i64, i32, vp]
        L.tgo_unpack_Aint4.argtypes = [vp, i64, i64, i32, vp]
        L.tgo_pack_Bint8.argtypes = [vp, i64, i64, i32, vp]
        L.tgo_pack_Aint8.argtypes = [vp, i64, i64, i32, vp]
        L.tgo_to_A16.argtypes = [vp, i64, i64, vp]
        L.tgo_from_A16.argtypes = [vp, i64, i64, vp]
        L.tgo_to_B16.argtypes = [vp, i64, i64, i32, vp]
        L.tgo_from_B16.argtypes = [vp, i64, i64, i32, vp]
        L.tgo_dequant.argtypes = [vp, i64, i64, i32, i32, i32, vp, vp, vp]
        L.tgo_gemm.argtypes = [vp, vp, i64, i64, i64, i32, i64, vp, vp]
        L.tgo_linear.argtypes = [vp, vp, i64, i64, i64, i32, i32, i32, vp, vp, i64, vp, vp]
        L.tgo_linear_group_scaled.argtypes = [vp, vp, i64, i64, i64, i32, i32, i32, vp, vp, i64, vp, vp]
        L.tgo_dequant_int4_debug.argtypes = [vp, i64, vp]
        L.tgo_set_num_threads.argtypes = [i32]
        for f in ("tgo_pack_Bint4", "tgo_unpack_Bint4", "tgo_pack_Aint4", "tgo_unpack_Aint4", "tgo_to_A16",
                  "tgo_from_A16", "tgo_to_B16", "tgo_from_B16", "tgo_dequant", "tgo_gemm", "tgo_linear", "tgo_linear_group_scaled",
                  "tgo_dequant_int4_debug", "tgo_num_threads"):
            getattr(L, f).restype = i32
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


def _check(rc, what):
    if rc != 0:
        raise ValueError(f"oracle {what} rejected its arguments (code {rc})")


def _cdiv(a, b):
    return (a + b - 1) // b


# --- P1 / P2 -------------------------------------------------------------------------------

def pack_Bint4(codes: np.ndarray, inner_k_tiles: int) -> np.ndarray:
    codes = _c(codes, np.int32)
    n, k = codes.shape
    I = inner_k_tiles
    if I not in (2, 4, 8) or k % (16 * I):
        raise ValueError("Bint4: innerKTiles must be 2/4/8 and k % (16*innerKTiles) == 0")
    out = np.empty((_cdiv(n, 8), k // (16 * I), 32, I // 2), np.int32)
    _check(lib().tgo_pack_Bint4(_p(codes), n, k, I, _p(out)), "pack_Bint4")
    return out


def unpack_Bint4(packed: np.ndarray, n: int, k: int) -> np.ndarray:
    packed = _c(packed, np.int32)
    I = packed.shape[3] * 2
    out = np.zeros((n, k), np.int32)
    _check(lib().tgo_unpack_Bint4(_p(packed), n, k, I, _p(out)), "unpack_Bint4")
    return out


def pack_Aint4(codes: np.ndarray, inner_k_tiles: int) -> np.ndarray:
    codes = _c(codes, np.int32)
    m, k = codes.shape
    I = inner_k_tiles
    if I not in (1, 2, 4):
        raise ValueError("Aint4: innerKTiles must be 1/2/4")
    out = np.empty((_cdiv(m, 16), _cdiv(k, 16 * I), 32, I), np.int32)
    _check(lib().tgo_pack_Aint4(_p(codes), m, k, I, _p(out)), "pack_Aint4")
    return out


def pack_Bint8(codes: np.ndarray, inner_k_tiles: int) -> np.ndarray:
    """int32 [n][k] (byte codes 0..255) -> [ceil(n/8)][k/(16 I)][32][I] (TinyGemmConvertB.cu:366-411)."""
    codes = _c(codes, np.int32)
    n, k = codes.shape
    I = inner_k_tiles
    if I not in (1, 2, 4) or k % (16 * I):
        raise ValueError("Bint8: innerKTiles must be 1/2/4 and k a multiple of 16*innerKTiles")
    out = np.empty((_cdiv(n, 8), k // (16 * I), 32, I), np.int32)
    _check(lib().tgo_pack_Bint8(_p(codes), n, k, I, _p(out)), "pack_Bint8")
    return out


def pack_Aint8(codes: np.ndarray, inner_k_tiles: int) -> np.ndarray:
    """int32 [m][k] -> [ceil(m/16)][ceil(ceil(k/16)/I)][32][2 I] (TinyGemmConvertA.cu:337-397)."""
    codes = _c(codes, np.int32)
    m, k = codes.shape
    I = inner_k_tiles
    if I not in (1, 2):
        raise ValueError("Aint8: innerKTiles must be 1/2")
    out = np.empty((_cdiv(m, 16), _cdiv(_cdiv(k, 16), I), 32, 2 * I), np.int32)
    _check(lib().tgo_pack_Aint8(_p(codes), m, k, I, _p(out)), "pack_Aint8")
    return out


def unpack_Aint4(packed: np.ndarray, m: int, k: int) -> np.ndarray:
    packed = _c(packed, np.int32)
    I = packed.shape[3]
    out = np.zeros((m, k), np.int32)
    _check(lib().tgo_unpack_Aint4(_p(packed), m, k, I, _p(out)), "unpack_Aint4")
    return out


# --- P3 ------------------------------------------------------------------------------------

def to_A16(x: np.ndarray) -> np.ndarray:
    x = _c(x, np.uint16)
    m, k = x.shape
    out = np.empty((_cdiv(m, 16), _cdiv(k, 16), 32, 8), np.uint16)
    _check(lib().tgo_to_A16(_p(x), m, k, _p(out)), "to_A16")
    return out


def from_A16(t: np.ndarray, m: int, k: int) -> np.ndarray:
    t = _c(t, np.uint16)
    out = np.zeros((m, k), np.uint16)
    _check(lib().tgo_from_A16(_p(t), m, k, _p(out)), "from_A16")
    return out


def to_B16(x: np.ndarray, inner_k_tiles: int) -> np.ndarray:
    x = _c(x, np.uint16)
    n, k = x.shape
    I = inner_k_tiles
    out = np.empty((_cdiv(n, 8), _cdiv(k, 16 * I), 32, 4 * I), np.uint16)
    _check(lib().tgo_to_B16(_p(x), n, k, I, _p(out)), "to_B16")
    return out


def from_B16(t: np.ndarray, n: int, k: int) -> np.ndarray:
    t = _c(t, np.uint16)
    I = t.shape[3] // 4
    out = np.zeros((n, k), np.uint16)
    _check(lib().tgo_from_B16(_p(t), n, k, I, _p(out)), "from_B16")
    return out


# --- D1-D5, H8 -----------------------------------------------------------------------------

def dequant(codes, group, qtype, qinfo, lut=None, dtype=BF16) -> np.ndarray:
    """codes int32 [rows][k]; qinfo uint16 [k/g][rows][2] (or uint8 [rows][k/g] for mx4);
    lut uint16 [16] / [rows][16].  Returns the 16-bit weight matrix [rows][k] (uint16)."""
    codes = _c(codes, np.int32)
    rows, k = codes.shape
    qinfo = _c(qinfo, np.uint8 if qtype == Q_MX4 else np.uint16)
    lut = None if lut is None else _c(lut, np.uint16)
    out = np.empty((rows, k), np.uint16)
    _check(lib().tgo_dequant(_p(codes), rows, k, group, qtype, dtype, _p(qinfo), _p(lut), _p(out)), "dequant")
    return out


def gemm(x, w, dtype=BF16):
    """x uint16 [m][k], w uint16 [rows][k] -> (y16 uint16 [m][rows], y32 float32 [m][rows])."""
    x = _c(x, np.uint16)
    w = _c(w, np.uint16)
    m, k = x.shape
    rows = w.shape[0]
    y16 = np.empty((m, rows), np.uint16)
    y32 = np.empty((m, rows), np.float32)
    _check(lib().tgo_gemm(_p(x), _p(w), m, rows, k, dtype, rows, _p(y16), _p(y32)), "gemm")
    return y16, y32


def linear(x, codes, group, qtype, qinfo, lut=None, dtype=BF16):
    """Fused dequant + contraction (same arithmetic as dequant() then gemm())."""
    x = _c(x, np.uint16)
    codes = _c(codes, np.int32)
    m, k = x.shape
    rows = codes.shape[0]
    qinfo = _c(qinfo, np.uint8 if qtype == Q_MX4 else np.uint16)
    lut = None if lut is None else _c(lut, np.uint16)
    y16 = np.empty((m, rows), np.uint16)
    y32 = np.empty((m, rows), np.float32)
    _check(lib().tgo_linear(_p(x), _p(codes), m, rows, k, group, qtype, dtype, _p(qinfo), _p(lut), rows,
                            _p(y16), _p(y32)), "linear")
    return y16, y32


def linear_group_scaled(x, codes, group, qtype, qinfo, lut=None, dtype=BF16):
    """The same contraction with scale / zero applied per quantisation group to the exact group sums (the library's
    TG_NUM_FAST numerics; derived, see tgo_linear_group_scaled).  Returns (y16, y32) like linear()."""
    x = _c(x, np.uint16)
    codes = _c(codes, np.int32)
    m, k = x.shape
    rows = codes.shape[0]
    qinfo = _c(qinfo, np.uint8 if qtype == Q_MX4 else np.uint16)
    lut = None if lut is None else _c(lut, np.uint16)
    y16 = np.empty((m, rows), np.uint16)
    y32 = np.empty((m, rows), np.float32)
    _check(lib().tgo_linear_group_scaled(_p(x), _p(codes), m, rows, k, group, qtype, dtype, _p(qinfo), _p(lut), rows,
                                         _p(y16), _p(y32)), "linear_group_scaled")
    return y16, y32


def dequant_int4_debug(words: np.ndarray) -> np.ndarray:
    words = _c(words, np.int32).reshape(-1)
    out = np.empty(words.size * 8, np.uint16)
    _check(lib().tgo_dequant_int4_debug(_p(words), words.size, _p(out)), "dequant_int4_debug")
    return out


def num_threads() -> int:
    return lib().tgo_num_threads()


def set_num_threads(n: int) -> None:
    lib().tgo_set_num_threads(n)


# --- bit-pattern helpers -------------------------------------------------------------------

def bf16_bits(a: np.ndarray) -> np.ndarray:
    """float32 array -> bf16 bit patterns (RNE), via the same C routine the oracle uses."""
    a = np.ascontiguousarray(a, np.float32)
    u = a.view(np.uint32)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + (0x7FFF + ((u >> 16) & 1))) >> 16).astype(np.uint16)
    r[nan] = ((u[nan] >> 16) | 0x40).astype(np.uint16)
    return r


def bf16_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, np.uint16).astype(np.uint32) << 16).view(np.float32)
