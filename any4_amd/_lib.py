"""ctypes binding of include/tinygemm_hip.h.  No fallbacks: if the HIP library is missing the
import fails loudly (build it with `python -m any4_amd.build`)."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtinygemm_hip.so")

TG_BF16, TG_F16 = 0, 1
TG_Q_INT4, TG_Q_ANY4_GLOBAL, TG_Q_ANY4_ROWWISE, TG_Q_MX4, TG_Q_INT8 = 0, 1, 2, 3, 4
TG_NUM_FAST, TG_NUM_REFERENCE, TG_NUM_FAST_MFMA, TG_NUM_FAST_DOT2 = 0, 1, 2, 3
TG_ABI_VERSION = 8
TG_PLAN_SPLITK, TG_PLAN_STREAM, TG_PLAN_PAIR, TG_PLAN_PAIR_XR, TG_PLAN_GEMV, TG_PLAN_TILE = 1, 2, 3, 4, 5, 6
TG_LAYOUT_RM, TG_LAYOUT_TC_A = 0, 1
TG_E_LAYOUT = -12
TG_E_FUSION = -13
TG_E_STRUCT = -14
TG_EPI_NONE, TG_EPI_SWIGLU = 0, 1
TG_WFMT_M16N8K16, TG_WFMT_ROWS = 0, 1

_i32, _i64, _vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p


class W4Gemm(ctypes.Structure):
    """struct tg_w4_gemm (include/tinygemm_hip.h)"""

    _fields_ = [
        ("struct_bytes", ctypes.c_uint32), ("struct_reserved", ctypes.c_uint32),
        ("x", _vp), ("w", _vp), ("qinfo", _vp), ("lut", _vp), ("y", _vp),
        ("m", _i64), ("wrows", _i64), ("k", _i64),
        ("group", _i32), ("qtype", _i32), ("dtype", _i32), ("w_on_right", _i32), ("inner_k_tiles", _i32),
        ("batch", _i32),
        ("stride_x", _i64), ("stride_w", _i64), ("stride_qinfo", _i64), ("stride_lut", _i64), ("stride_y", _i64),
        ("numerics", _i32), ("reserved", _i32), ("bias", _vp), ("stride_bias", _i64),
        ("workspace", _vp), ("workspace_bytes", _i64),
        ("x_layout", _i32), ("y_layout", _i32),
        ("bias_row_stride", _i64), ("norm_weight", _vp), ("norm_eps", ctypes.c_float), ("epilogue", _i32),
        ("w_format", _i32), ("reserved6", _i32),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        if "struct_bytes" not in kw:  # ABI 6: the struct says how long it is
            self.struct_bytes = ctypes.sizeof(W4Gemm)


TG_PEER_MAX_WORLD = 16


class PeerHandle(ctypes.Structure):
    """struct tg_peer_handle (include/peer_gather_hip.h): a hipIpcMemHandle_t"""

    _fields_ = [("bytes", ctypes.c_ubyte * 64)]


class PeerGather(ctypes.Structure):
    """struct tg_peer_gather (include/peer_gather_hip.h)"""

    _fields_ = [
        ("src", _vp), ("dst", _vp * TG_PEER_MAX_WORLD), ("flags", _vp * TG_PEER_MAX_WORLD), ("seq", _vp), ("status", _vp),
        ("world", _i32), ("rank", _i32), ("m", _i64), ("cols_local", _i64), ("timeout_us", _i64),
    ]


# name -> argtypes, exactly the prototypes of include/tinygemm_hip.h
SYMBOLS = {
    "tg_abi_version": [],
    "tg_m1_default_contraction": [],
    "tg_error_string": [ctypes.c_int],
    "tg_convert_to_Bint4": [_vp, _i64, _i64, ctypes.c_int, _vp, ctypes.c_int, _vp],
    "tg_convert_to_Aint4": [_vp, _i64, _i64, ctypes.c_int, _vp, ctypes.c_int, _vp],
    "tg_convert_to_A16": [_vp, _i64, _i64, _vp, ctypes.c_int, _vp],
    "tg_convert_from_A16": [_vp, _i64, _i64, _vp, ctypes.c_int, _vp],
    "tg_convert_to_B16": [_vp, _i64, _i64, ctypes.c_int, _vp, ctypes.c_int, _vp],
    "tg_convert_from_B16": [_vp, _i64, _i64, ctypes.c_int, _vp, ctypes.c_int, _vp],
    "tg_dequant_int4": [_vp, _i64, _vp, ctypes.c_int, _vp],
    "tg_unpack_int4": [_vp, ctypes.c_int, _i64, _i64, ctypes.c_int, _vp, ctypes.c_int, _vp],
    "tg_dequant_w4": [_vp, _vp, _vp, _i64, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp],
    "tg_dequant_w4_panel": [_vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp],
    "tg_gemm_w4": [ctypes.POINTER(W4Gemm), ctypes.c_int, _vp],
    "tg_gemm_w4_plan": [ctypes.POINTER(W4Gemm), ctypes.c_int],
    "tg_gemm_w4_workspace_bytes": [ctypes.POINTER(W4Gemm)],
    "tg_gemm_f16": [_vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp],
    "tg_convert_to_Bint8": [_vp, _i64, _i64, ctypes.c_int, _vp, ctypes.c_int, _vp],
    "tg_convert_to_Aint8": [_vp, _i64, _i64, ctypes.c_int, _vp, ctypes.c_int, _vp],
    "tg_gemm_w8": [ctypes.POINTER(W4Gemm), ctypes.c_int, _vp],
    "tg_gemm_w8_workspace_bytes": [ctypes.POINTER(W4Gemm)],
    # include/peer_gather_hip.h (one-shot peer-write gather of row-sharded outputs)
    "tg_peer_alloc": [ctypes.c_int, _i64, ctypes.POINTER(_vp)],
    "tg_peer_free": [ctypes.c_int, _vp],
    "tg_peer_export": [ctypes.c_int, _vp, ctypes.POINTER(PeerHandle)],
    "tg_peer_open": [ctypes.c_int, ctypes.POINTER(PeerHandle), ctypes.POINTER(_vp)],
    "tg_peer_close": [ctypes.c_int, _vp],
    "tg_peer_gather_launch": [ctypes.POINTER(PeerGather), ctypes.c_int, _vp],
    # include/decode_glue_hip.h (non-GEMM kernels of the decode harness)
    "dg_add_rmsnorm": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_float, ctypes.c_int, ctypes.c_int, _vp],
    "dg_rope_kv": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i64,
                   ctypes.c_int, ctypes.c_int, _vp],
    "dg_decode_attn": [_vp, _vp, _vp, _vp, _vp, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i64, ctypes.c_float,
                       ctypes.c_int, ctypes.c_int, _vp],
    "dg_rope_attn": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i64, ctypes.c_float,
                     ctypes.c_int, ctypes.c_int, _vp],
    "dg_rope_attn_online": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i64, ctypes.c_float,
                     ctypes.c_int, ctypes.c_int, _vp],
    "dg_rope_attn_split_scratch_bytes": [_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int],
    "dg_rope_attn_split": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i64,
                           ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp],
    "dg_swiglu": [_vp, _vp, _i64, _i64, ctypes.c_int, ctypes.c_int, _vp],
    "dg_linear16": [_vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_int, ctypes.c_int, _vp],
}

_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"tinygemm HIP library not built: {LIB_PATH} is missing. "
            "Run `python -m any4_amd.build` (needs hipcc; cross-compiles gfx950 without a GPU). "
            "There is no CPU/PyTorch fallback for the tinygemm ops."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # a stale build: the header declares a symbol the .so does not export
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild with `python -m any4_amd.build`") from e
        fn.argtypes = argtypes
        fn.restype = (ctypes.c_char_p if name == "tg_error_string" else
                      ctypes.c_int64 if name in ("dg_rope_attn_split_scratch_bytes", "tg_gemm_w4_workspace_bytes", "tg_gemm_w8_workspace_bytes") else ctypes.c_int)
    if lib.tg_abi_version() != TG_ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.tg_abi_version()} != {TG_ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().tg_error_string(rc).decode()
        raise RuntimeError(f"tinygemm::{what}: {msg} (code {rc})")
