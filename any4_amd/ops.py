"""`torch.ops.tinygemm.*` -- the reference's op surface (tinygemm_lib/TinyGemm.cpp:17-122), registered
from Python with torch.library and implemented by calls into the C-ABI HIP library
(include/tinygemm_hip.h).  ROCm tensors dispatch on the "CUDA" key; there is deliberately no CPU
implementation, so a CPU tensor (or a missing .so) fails loudly instead of falling back.

Host-side validation mirrors the TORCH_CHECKs of the reference host functions
(TinyGemm_int4.cu:28-548, TinyGemm_bf16.cu, TinyGemmConvert{A,B}.cu) and raises RuntimeError like
c10::Error does.  Outputs are allocated here with torch.empty (caching allocator, current stream)
and handed to the library, which never allocates.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading

import torch

from . import _lib
from ._lib import TG_BF16, TG_F16, TG_Q_ANY4_GLOBAL, TG_Q_ANY4_ROWWISE, TG_Q_INT4, TG_Q_MX4, W4Gemm

_L = _lib.load()  # ImportError if the HIP library has not been built

NAMESPACE = "tinygemm"

# schema strings: verbatim argument lists of TinyGemm.cpp:17-122
SCHEMAS = {
    "convert_matrix_to_m16n8k16_A_layout": "(Tensor t, int innerKTiles) -> Tensor",
    "convert_matrix_to_m16n8k16_Aint4_layout": "(Tensor t, int innerKTiles) -> Tensor",
    "convert_matrix_to_m16n8k16_Aint8_layout": "(Tensor t, int innerKTiles) -> Tensor",
    "convert_matrix_from_m16n8k16_A_layout": "(Tensor t, int m, int k) -> Tensor",
    "convert_matrix_to_m16n8k16_B_layout": "(Tensor t, int innerKTiles) -> Tensor",
    "convert_matrix_to_m16n8k16_Bint4_layout": "(Tensor t, int innerKTiles) -> Tensor",
    "convert_matrix_to_m16n8k16_Bint8_layout": "(Tensor t, int innerKTiles) -> Tensor",
    "convert_matrix_from_m16n8k16_B_layout": "(Tensor t, int n, int k) -> Tensor",
    "tinygemm_y_f16TC_x_f16TC_w_int4TC": "(Tensor A, Tensor B, int qGroupSize, Tensor qScaleAndZeros, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16RM_x_f16RM_w_int4TC": "(Tensor A, Tensor B, int qGroupSize, Tensor qScaleAndZeros, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16TC_x_f16TC_w_any4TC": "(Tensor A, Tensor B, int qGroupSize, Tensor qScaleAndZeros, Tensor int4DequantValues, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16RM_x_f16RM_w_any4TC": "(Tensor A, Tensor B, int qGroupSize, Tensor qScaleAndZeros, Tensor int4DequantValues, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16TC_x_f16TC_w_mx4TC": "(Tensor A, Tensor B, int qGroupSize, Tensor mx4Exponents, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16RM_x_f16RM_w_mx4TC": "(Tensor A, Tensor B, int qGroupSize, Tensor mx4Exponents, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16TC_x_f16TC_w_int8TC": "(Tensor A, Tensor B, int qGroupSize, Tensor qScaleAndZeros, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16RM_x_f16RM_w_int8TC": "(Tensor A, Tensor B, int qGroupSize, Tensor qScaleAndZeros, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16TC_x_f16TC_w_f16TC": "(Tensor A, Tensor B, bool weightOnRight) -> Tensor",
    "tinygemm_y_f16RM_x_f16RM_w_f16TC": "(Tensor A, Tensor B, bool weightOnRight) -> Tensor",
    "tinygemm_dequant_int4": "(Tensor t) -> Tensor",
}

_F16_TYPES = (torch.bfloat16, torch.float16)

# ---------------------------------------------------------------------------------------------
# call-side state the reference's op schemas have no argument for: GEMM numerics and a fused bias
# ---------------------------------------------------------------------------------------------
_NUMERICS = {"fast": _lib.TG_NUM_FAST, "reference": _lib.TG_NUM_REFERENCE, "fast_mfma": _lib.TG_NUM_FAST_MFMA, "fast_dot2": _lib.TG_NUM_FAST_DOT2}
_numerics = os.environ.get("ANY4_NUMERICS", "fast")
if _numerics not in _NUMERICS:
    raise ImportError(f"ANY4_NUMERICS must be one of {sorted(_NUMERICS)}, got {_numerics!r}")
_tls = threading.local()


def get_numerics() -> str:
    """'fast' (default): the 4-bit GEMMs may apply scale / zero per quantisation group to the f32 accumulator instead of to
    every weight (include/tinygemm_hip.h, TG_NUM_FAST: no per-weight rounding, results within the reference's own weight
    rounding).  'reference': bit-identical dequantised weights (w = RNE16(fma(lut, scale, zero))), as the reference kernels.
    'fast_mfma': 'fast' with the m = 1 contraction of stacked launches on the MFMA instead of v_dot2 (TG_NUM_FAST_MFMA)."""
    return getattr(_tls, "numerics", _numerics)


def set_numerics(name: str) -> None:
    """Process-wide default (ANY4_NUMERICS in the environment sets the initial value)."""
    global _numerics
    if name not in _NUMERICS:
        raise ValueError(f"numerics must be one of {sorted(_NUMERICS)}")
    _numerics = name


@contextlib.contextmanager
def numerics(name: str):
    """Thread-local override: `with any4_amd.numerics("reference"): ...`"""
    if name not in _NUMERICS:
        raise ValueError(f"numerics must be one of {sorted(_NUMERICS)}")
    prev = getattr(_tls, "numerics", None)
    _tls.numerics = name
    try:
        yield
    finally:
        if prev is None:
            del _tls.numerics
        else:
            _tls.numerics = prev


# ---------------------------------------------------------------------------------------------
# packed format of weights-on-the-left (Aint4) tensors
# ---------------------------------------------------------------------------------------------
_WFORMATS = {"native": _lib.TG_WFMT_ROWS, "reference": _lib.TG_WFMT_M16N8K16}
_wformat = os.environ.get("ANY4_WEIGHT_FORMAT", "native")
if _wformat not in _WFORMATS:
    raise ImportError(f"ANY4_WEIGHT_FORMAT must be one of {sorted(_WFORMATS)}, got {_wformat!r}")


def get_weight_format() -> str:
    """What `convert_matrix_to_m16n8k16_Aint4_layout` PRODUCES (the packed tensor is opaque to every caller of the reference:
    TinyGemm_int4.cu:322-364 only checks its shape).  What a GEMM op CONSUMES is decided by the tensor itself, never
    by this setting: the two formats have different shapes (`aside_format`), so a tensor packed under either setting, in any
    process, by this library or by the CUDA implementation, is multiplied correctly or rejected -- and `state_dict` carries it.
    'native' (default): the Bint4 tensor of the weight rows padded to 16, [2 ceil(m/16)][k/(16 J)][32][J/2] with J = 4 (k % 64
    == 0) or 2 -- row-per-lane order, a packed word holds 8 codes of ONE weight row instead of 4 + 4 of rows r and r + 8; the
    A-side ops then run the B-side kernels (tg_w4_gemm.w_format = TG_WFMT_ROWS).  'reference': the reference's Aint4 tensor
    [ceil(m/16)][k/(16 I)][32][I], bit for bit (`relayout_Aint4` converts either way, losslessly)."""
    return getattr(_tls, "wformat", _wformat)


def set_weight_format(name: str) -> None:
    global _wformat
    if name not in _WFORMATS:
        raise ValueError(f"weight format must be one of {sorted(_WFORMATS)}")
    _wformat = name


@contextlib.contextmanager
def weight_format(name: str):
    """Thread-local override: `with any4_amd.weight_format("reference"): ...`"""
    if name not in _WFORMATS:
        raise ValueError(f"weight format must be one of {sorted(_WFORMATS)}")
    prev = getattr(_tls, "wformat", None)
    _tls.wformat = name
    try:
        yield
    finally:
        if prev is None:
            del _tls.wformat
        else:
            _tls.wformat = prev


_auto_relayout = os.environ.get("ANY4_AUTO_RELAYOUT", "1") not in ("0", "false", "False", "")


def get_auto_relayout() -> bool:
    """Whether a module that RECEIVES a weights-on-the-left tensor in the reference's Aint4 words through load_state_dict (a checkpoint
    packed by the CUDA implementation, modules.py:197-205) repacks it once to the row-per-lane order (lossless; relayout_Aint4).  On by
    default while the process default weight format is 'native'; ANY4_AUTO_RELAYOUT=0 or set_auto_relayout(False) keeps the words."""
    return _auto_relayout


def set_auto_relayout(on: bool) -> None:
    global _auto_relayout
    _auto_relayout = bool(on)


def _rows_inner(k: int) -> int:
    """innerKTiles of the Bint4 word order of a native weights-on-the-left tensor (tg_w4_gemm.w_format = TG_WFMT_ROWS)."""
    return 4 if k % 64 == 0 else 2


def aside_format(w: torch.Tensor, k: int) -> str:
    """Which packed format a weights-on-the-left tensor holds, from its shape and the activations' k alone.  The reference's
    Aint4 tensor [m/16][k/(16 I)][32][I] covers size(1) * size(3) * 16 = k; the native tensor (a Bint4 tensor,
    [m/8][k/(16 J)][32][J/2]) covers size(1) * size(3) * 32 = k: no tensor satisfies both for one k."""
    span = w.size(1) * w.size(3) * 16
    if span == k:
        return "reference"
    _check(span * 2 == k and w.size(0) % 2 == 0 and w.size(3) * 2 == _rows_inner(k),
           "weights: k super-tiles do not match the activations' k")
    return "native"


_PLANS = {_lib.TG_PLAN_SPLITK: "splitk", _lib.TG_PLAN_STREAM: "stream", _lib.TG_PLAN_PAIR: "pair", _lib.TG_PLAN_PAIR_XR: "pair_xr", _lib.TG_PLAN_GEMV: "gemv",
          _lib.TG_PLAN_TILE: "tile"}


def gemm_w4_plan(m: int, wrows: int, k: int, group: int, qtype: int, weight_on_right: bool = True, inner_k_tiles: int = 4,
                 dtype=torch.bfloat16, batch: int = 1, numerics: str | None = None, workspace: bool = True, detail: bool = False,
                 weight_format: str | None = None) -> str:
    """Which kernel family tg_gemm_w4 launches for this problem (tg_gemm_w4_plan; nothing is launched, no GPU needed):
    'pair' = pair-table kernels, group-scaled numerics; 'stream' / 'splitk' = reference-numerics kernels.
    `workspace`: the caller provides the scratch tg_gemm_w4_workspace_bytes asks for (the ops of this module do).
    `detail`: name the member of the pair-table family too ('pair_xr' = w4_gemm_xr_kernel, activations resident in registers)."""
    buf = ctypes.create_string_buffer(256)
    p = (ctypes.addressof(buf) + 63) & ~63  # a non-NULL, aligned dummy: the planner never dereferences data pointers
    args = W4Gemm(x=p, w=p, qinfo=p, lut=p, y=p, m=m, wrows=wrows, k=k, group=group, qtype=qtype,
                  dtype=TG_BF16 if dtype == torch.bfloat16 else TG_F16, w_on_right=1 if weight_on_right else 0,
                  inner_k_tiles=inner_k_tiles, batch=batch, stride_x=16, stride_w=16, stride_qinfo=16, stride_lut=16, stride_y=16,
                  numerics=_NUMERICS[numerics or get_numerics()],
                  w_format=0 if weight_on_right else _WFORMATS[weight_format or get_weight_format()])
    if workspace:
        need = _L.tg_gemm_w4_workspace_bytes(ctypes.byref(args))
        _lib.check(need if need < 0 else 0, "tg_gemm_w4_workspace_bytes")
        if need > 0:
            args.workspace, args.workspace_bytes = p, need
    rc = _L.tg_gemm_w4_plan(ctypes.byref(args), 0)
    _lib.check(rc if rc < 0 else 0, "tg_gemm_w4_plan")
    plan = _PLANS[rc]
    return plan if detail else plan.split("_")[0]


class _FusedBias:
    def __init__(self, bias):
        self.bias = bias
        self.consumed = False


@contextlib.contextmanager
def fused_bias(bias: torch.Tensor):
    """Offer `bias` ([weight rows], 16-bit) to the next row-major 4-/8-bit GEMM op called on this thread: the kernel adds it
    in its output store (bit-identical to the reference module's separate `y + bias`, modules.py:221-222).  The op takes
    it only if it fits (same dtype / device, one value per tile-padded weight row); `.consumed` tells the caller."""
    fb = _FusedBias(bias)
    prev = getattr(_tls, "bias", None)
    _tls.bias = fb
    try:
        yield fb
    finally:
        _tls.bias = prev


def _take_bias(wrows: int, x: torch.Tensor):
    fb = getattr(_tls, "bias", None)
    if fb is None or fb.consumed:
        return None
    b = fb.bias
    if b.dim() != 1 or b.numel() != wrows or b.dtype != x.dtype or b.device != x.device or not b.is_contiguous() or b.data_ptr() % 8:
        return None
    fb.consumed = True
    return b


def _check(cond: bool, msg: str) -> None:
    if not cond:
        raise RuntimeError(f"tinygemm: {msg}")


def _cdiv(a: int, b: int) -> int:
    return (a + b - 1) // b


def _dt(t: torch.Tensor) -> int:
    return TG_BF16 if t.dtype == torch.bfloat16 else TG_F16


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # (the handle without building a torch.cuda.Stream object: ~2 us per call less)


def _stream(t: torch.Tensor) -> int:
    if _raw_stream is not None:
        return _raw_stream(_dev(t))
    return torch.cuda.current_stream(t.device).cuda_stream


def _dev(t: torch.Tensor) -> int:
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


# ---------------------------------------------------------------------------------------------
# layout conversion
# ---------------------------------------------------------------------------------------------

def convert_matrix_to_m16n8k16_Bint4_layout(t: torch.Tensor, innerKTiles: int) -> torch.Tensor:
    _check(t.dim() == 2, "Bint4 layout: input must be 2-D [n][k]")
    _check(t.dtype == torch.int32, "Bint4 layout: input must be int32")
    _check(t.is_contiguous(), "Bint4 layout: input must be contiguous")
    _check(innerKTiles in (2, 4, 8), "Bint4 layout: innerKTiles must be 2, 4 or 8")
    n, k = t.shape
    _check(k % (innerKTiles * 16) == 0, "Bint4 layout: k must be a multiple of innerKTiles * 16")
    out = torch.empty((_cdiv(n, 8), k // (innerKTiles * 16), 32, innerKTiles // 2), dtype=torch.int32, device=t.device)
    _lib.check(_L.tg_convert_to_Bint4(t.data_ptr(), n, k, innerKTiles, out.data_ptr(), _dev(t), _stream(t)),
               "convert_matrix_to_m16n8k16_Bint4_layout")
    return out


def convert_matrix_to_m16n8k16_Aint4_layout(t: torch.Tensor, innerKTiles: int) -> torch.Tensor:
    _check(t.dim() == 2, "Aint4 layout: input must be 2-D [m][k]")
    _check(t.dtype == torch.int32, "Aint4 layout: input must be int32")
    _check(t.is_contiguous(), "Aint4 layout: input must be contiguous")
    _check(innerKTiles in (1, 2, 4), "Aint4 layout: innerKTiles must be 1, 2 or 4")
    m, k = t.shape
    if get_weight_format() == "native" and k % 32 == 0 and k % (innerKTiles * 16) == 0:
        # the row-per-lane order (any k a weights-on-the-left GEMM accepts); other k: the reference's words (no GEMM takes them).
        # The tensor says what it is by its shape (aside_format): the Bint4 tensor of the rows padded to 16.
        j = _rows_inner(k)
        alloc = torch.zeros if _cdiv(m, 8) % 2 else torch.empty  # an odd number of 8-row tiles: the pad tile is all zero codes
        out = alloc((2 * _cdiv(m, 16), k // (16 * j), 32, j // 2), dtype=torch.int32, device=t.device)
        _lib.check(_L.tg_convert_to_Bint4(t.data_ptr(), m, k, j, out.data_ptr(), _dev(t), _stream(t)),
                   "convert_matrix_to_m16n8k16_Aint4_layout")
        return out
    out = torch.empty((_cdiv(m, 16), _cdiv(k, innerKTiles * 16), 32, innerKTiles), dtype=torch.int32, device=t.device)
    _lib.check(_L.tg_convert_to_Aint4(t.data_ptr(), m, k, innerKTiles, out.data_ptr(), _dev(t), _stream(t)),
               "convert_matrix_to_m16n8k16_Aint4_layout")
    return out


def unpack_int4(packed: torch.Tensor, rows: int, k: int, layout: str) -> torch.Tensor:
    """Packed 4-bit words -> int32 codes [rows][k] (tg_unpack_int4).  layout: 'B' (Bint4 words, innerKTiles = 2 size(3); a native
    weights-on-the-left tensor is one) or 'A' (the reference's Aint4 words, innerKTiles = size(3))."""
    _check(packed.dim() == 4 and packed.dtype == torch.int32 and packed.is_contiguous(), "unpack_int4: a contiguous 4-D int32 tensor")
    _check(layout in ("A", "B"), "unpack_int4: layout must be 'A' or 'B'")
    inner = packed.size(3) if layout == "A" else 2 * packed.size(3)
    out = torch.empty((rows, k), dtype=torch.int32, device=packed.device)
    _lib.check(_L.tg_unpack_int4(packed.data_ptr(), 1 if layout == "A" else 0, rows, k, inner, out.data_ptr(), _dev(packed),
                                 _stream(packed)), "unpack_int4")
    return out


def relayout_Aint4(packed: torch.Tensor, k: int, to: str, inner_k_tiles: int | None = None) -> torch.Tensor:
    """Lossless repack of a weights-on-the-left tensor between the reference's Aint4 words ('reference') and the row-per-lane
    order ('native'), e.g. for a checkpoint packed by the CUDA implementation.  The tensor's own shape says which it holds
    (aside_format); already in `to`: returned as it is.  inner_k_tiles: the Aint4 innerKTiles of a 'reference' result (default 4
    when k % 64 == 0, else 2)."""
    _check(to in _WFORMATS, f"relayout_Aint4: `to` must be one of {sorted(_WFORMATS)}")
    _check(packed.dim() == 4 and packed.dtype == torch.int32 and packed.is_contiguous() and packed.size(2) == 32,
           "relayout_Aint4: a contiguous 4-D int32 weights-on-the-left tensor")
    have = aside_format(packed, k)
    if have == to:
        return packed
    rows = packed.size(0) * (16 if have == "reference" else 8)
    codes = unpack_int4(packed, rows, k, "A" if have == "reference" else "B")
    with weight_format(to):
        return convert_matrix_to_m16n8k16_Aint4_layout(codes, inner_k_tiles or _rows_inner(k))


def convert_matrix_to_m16n8k16_A_layout(t: torch.Tensor, innerKTiles: int) -> torch.Tensor:
    _check(innerKTiles == 1, "A layout: innerKTiles must be 1")
    _check(t.dtype in _F16_TYPES, "A layout: input must be bfloat16 or float16")
    _check(t.dim() == 2 and t.is_contiguous(), "A layout: input must be a contiguous 2-D matrix")
    m, k = t.shape
    out = torch.empty((_cdiv(m, 16), _cdiv(k, 16), 32, 8), dtype=t.dtype, device=t.device)
    _lib.check(_L.tg_convert_to_A16(t.data_ptr(), m, k, out.data_ptr(), _dev(t), _stream(t)),
               "convert_matrix_to_m16n8k16_A_layout")
    return out


def convert_matrix_from_m16n8k16_A_layout(t: torch.Tensor, m: int, k: int) -> torch.Tensor:
    _check(t.dtype in _F16_TYPES, "A layout: input must be bfloat16 or float16")
    _check(t.dim() == 4 and t.is_contiguous(), "A layout: input must be a contiguous 4-D tensor")
    _check(_cdiv(m, 16) == t.size(0) and _cdiv(k, 16) == t.size(1), "A layout: tile counts do not match (m, k)")
    _check(t.size(2) == 32 and t.size(3) == 8, "A layout: expected [.., .., 32, 8]")
    out = torch.empty((m, k), dtype=t.dtype, device=t.device)
    _lib.check(_L.tg_convert_from_A16(t.data_ptr(), m, k, out.data_ptr(), _dev(t), _stream(t)),
               "convert_matrix_from_m16n8k16_A_layout")
    return out


def convert_matrix_to_m16n8k16_B_layout(t: torch.Tensor, innerKTiles: int) -> torch.Tensor:
    _check(t.dtype in _F16_TYPES, "B layout: input must be bfloat16 or float16")
    _check(t.dim() == 2 and t.is_contiguous(), "B layout: input must be a contiguous 2-D matrix")
    _check(innerKTiles in (1, 2), "B layout: innerKTiles must be 1 or 2")
    n, k = t.shape
    out = torch.empty((_cdiv(n, 8), _cdiv(k, 16 * innerKTiles), 32, innerKTiles * 4), dtype=t.dtype, device=t.device)
    _lib.check(_L.tg_convert_to_B16(t.data_ptr(), n, k, innerKTiles, out.data_ptr(), _dev(t), _stream(t)),
               "convert_matrix_to_m16n8k16_B_layout")
    return out


def convert_matrix_from_m16n8k16_B_layout(t: torch.Tensor, n: int, k: int) -> torch.Tensor:
    _check(t.dtype in _F16_TYPES, "B layout: input must be bfloat16 or float16")
    _check(t.dim() == 4 and t.is_contiguous(), "B layout: input must be a contiguous 4-D tensor")
    _check(t.size(3) % 4 == 0 and t.size(3) // 4 in (1, 2), "B layout: innermost dim must be 4 or 8")
    inner = t.size(3) // 4
    _check(_cdiv(n, 8) == t.size(0) and _cdiv(k, 16 * inner) == t.size(1) and t.size(2) == 32,
           "B layout: tile counts do not match (n, k)")
    out = torch.empty((n, k), dtype=t.dtype, device=t.device)
    _lib.check(_L.tg_convert_from_B16(t.data_ptr(), n, k, inner, out.data_ptr(), _dev(t), _stream(t)),
               "convert_matrix_from_m16n8k16_B_layout")
    return out


def tinygemm_dequant_int4(t: torch.Tensor) -> torch.Tensor:
    _check(t.dtype == torch.int32 and t.dim() == 1, "dequant_int4: expected a 1-D int32 tensor")
    t = t.contiguous()
    out = torch.empty((t.numel() * 8,), dtype=torch.bfloat16, device=t.device)
    if t.numel():
        _lib.check(_L.tg_dequant_int4(t.data_ptr(), t.numel(), out.data_ptr(), _dev(t), _stream(t)), "tinygemm_dequant_int4")
    return out


# ---------------------------------------------------------------------------------------------
# 4-bit weight GEMMs
# ---------------------------------------------------------------------------------------------

class _FragX:
    """Activations in A-fragment order [m/16][k/16][32][8] standing in for the row-major [m][k] matrix of _w4_rm."""

    def __init__(self, t):
        self.t = t
        self.shape = (t.size(0) * 16, t.size(1) * 16)
        self.dtype, self.device = t.dtype, t.device

    def dim(self):
        return 2

    def is_contiguous(self):
        return self.t.is_contiguous()

    def data_ptr(self):
        return self.t.data_ptr()


_WS_BYTES: dict = {}  # tg_gemm_w4_workspace_bytes per problem shape (pure function of the key below)


_LARGE_M = None


def large_m_rows(weights: int = 0) -> int:
    """Activation rows from which a 4-bit GEMM call takes the OPT-IN library route -- dequantise the weights (tg_dequant_w4, in bounded
    row panels) and multiply with the GEMM library (torch.matmul = hipBLASLt) -- instead of the library's own kernels; 0 / unset = never
    (the default: every call runs this library's kernels; beyond 64 rows that is the LDS-tiled MFMA GEMM of w4_gemm_tile.cuh, plan
    'tile').  ANY4_LARGE_M_GEMM=library turns the route on from 65 rows (96 for layers of 32 M weights or more), ANY4_LARGE_M=<rows> from
    that many.  Measured on MI355X, one 4096 x 4096 layer per graph node at 128 / 512 / 1024 / 2048 rows: own kernel 29.3 / 32.1 / 52.1 / 102 us,
    dequantise (14 us) + hipBLASLt 32.9 / 39.8 / 53.1 / 76 us (profiles/r06_tile_gemm.txt): the own kernel 1.1-1.25 x faster up to 512 rows, equal
    at 1024, the vendor GEMM 1.3-1.5 x faster beyond, at the cost of a transient 16-bit copy of a weight panel (at most ANY4_DEQUANT_PANEL_MB, default
    256 MB)."""
    global _LARGE_M
    if _LARGE_M is None:
        v = os.environ.get("ANY4_LARGE_M")
        if v is not None:
            _LARGE_M = int(v) if int(v) > 0 else 1 << 62
        else:
            _LARGE_M = -1 if os.environ.get("ANY4_LARGE_M_GEMM", "") == "library" else 1 << 62
    if _LARGE_M >= 0:
        return _LARGE_M
    return 96 if weights >= (1 << 25) else 65


def dequant_w4(w: torch.Tensor, qinfo: torch.Tensor, lut, q_group: int, qtype: int, k: int, inner: int, wrows: int, rows=None) -> torch.Tensor:
    """[wrows][k] 16-bit = the dequantised weights of a Bint4-packed tensor (or the native weights-on-the-left tensor: the same words):
    RNE16(fma(lut[row][code], scale, zero)) per element (tg_dequant_w4).  rows = (r0, r1): that panel of weight rows only (multiples of 8)."""
    r0, r1 = (0, wrows) if rows is None else rows
    out = torch.empty((r1 - r0, k), dtype=qinfo.dtype, device=w.device)
    ksuper = k // (16 * inner)
    wp = w.data_ptr() + (r0 // 8) * ksuper * 32 * (inner // 2) * 4
    # qinfo is [k / g][wrows][2]: a row panel starts r0 entries into every group's row (the kernel indexes group * wrows + row)
    qp = qinfo.data_ptr() + r0 * 2 * qinfo.element_size()
    lp = None if lut is None else (lut.data_ptr() + (r0 * 16 * lut.element_size() if lut.dim() == 2 else 0))
    _lib.check(_L.tg_dequant_w4_panel(wp, qp, lp, r1 - r0, wrows, k, q_group, qtype,
                                      TG_BF16 if qinfo.dtype == torch.bfloat16 else TG_F16, inner, out.data_ptr(), _dev(w), _stream(w)), "tg_dequant_w4")
    return out


def _library_gemm_w4(x, w, qinfo, lut, q_group, qtype, k, inner, wrows):
    """The opt-in route of large_m_rows: y = x . dequant(W)^T with the weights dequantised in row panels of bounded size."""
    cap = int(max(0.01, float(os.environ.get("ANY4_DEQUANT_PANEL_MB", "256"))) * (1 << 20))
    panel = max(8, min(wrows, (cap // (2 * k)) // 8 * 8))
    if x.data_ptr() % 16:
        x = x.clone()
    if panel >= wrows:
        return torch.matmul(x, dequant_w4(w, qinfo, lut, q_group, qtype, k, inner, wrows).t())
    y = torch.empty((x.shape[0], wrows), dtype=x.dtype, device=x.device)
    for r0 in range(0, wrows, panel):
        r1 = min(wrows, r0 + panel)
        torch.matmul(x, dequant_w4(w, qinfo, lut, q_group, qtype, k, inner, wrows, rows=(r0, r1)).t(), out=y[:, r0:r1])
    return y


def _w4_rm(A, B, q_group, qinfo, lut, qtype, weight_on_right, opname, frag=False):
    """Row-major activations / output.  Mirrors tinygemm_y_FT16RM_x_FT16RM_w_int4TC
    (TinyGemm_int4.cu:294-548).  frag=True (weights on the right only): A is a _FragX, the output comes back in A-fragment
    order [m/16][ceil(wrows/16)][32][8], or None when the library has no kernel that reads / writes fragment order itself
    for this problem (TG_E_LAYOUT: the caller converts around a row-major call)."""
    _check(A.device == B.device, "A and B must be on the same device")
    if weight_on_right:
        x, w = A, B
        _check(x.dim() == 2 and x.is_contiguous(), "activations must be a contiguous 2-D matrix")
        _check(w.dim() == 4 and w.dtype == torch.int32 and w.is_contiguous(), "weights must be a contiguous 4-D int32 tensor")
        inner = w.size(3) * 2
        _check(inner in (2, 4, 8), "Bint4 weights: innermost dim must be 1, 2 or 4")
        wrows = w.size(0) * 8
    else:
        w, x = A, B
        _check(w.dim() == 4 and w.dtype == torch.int32 and w.is_contiguous(), "weights must be a contiguous 4-D int32 tensor")
        _check(x.dim() == 2 and x.is_contiguous(), "activations must be a contiguous 2-D matrix")
        inner = w.size(3)
        _check(inner in (1, 2, 4), "Aint4 weights: innermost dim must be 1, 2 or 4")
        wrows = w.size(0) * 16
    m, k = x.shape
    k_tiles = _cdiv(k, 16)
    w_format = _lib.TG_WFMT_M16N8K16
    if not weight_on_right and w.size(1) != _cdiv(k_tiles, inner) and aside_format(w, k) == "native":
        # the tensor is the Bint4 tensor of the 16-row-padded weights (what the convert op returns by default): its shape says so
        w_format, inner, wrows = _lib.TG_WFMT_ROWS, w.size(3) * 2, w.size(0) * 8
    _check(w.size(1) == _cdiv(k_tiles, inner), "weights: k super-tiles do not match the activations' k")
    _check(w.size(2) == 32, "weights: dim 2 must be 32")
    _check(x.dtype in _F16_TYPES, "activation dtype must be bfloat16 or float16")
    _check(q_group in (32, 64, 128, 256), "qGroupSize must be 32, 64, 128 or 256")
    _check(qinfo.device == x.device, "quantization info must be on the activations' device")
    if qtype == TG_Q_MX4:
        _check(x.dtype == torch.bfloat16, "mx4 supports bfloat16 activations only")
        _check(k % q_group == 0, "qGroupSize must divide k")
        _check(qinfo.dtype == torch.uint8 and qinfo.dim() == 2, "mx4Exponents must be a 2-D uint8 tensor")
        _check(qinfo.size(0) == wrows and qinfo.size(1) == k // q_group, "mx4Exponents must be [weight rows (tile padded)][k / qGroupSize]")
    else:
        _check(qinfo.dim() == 3, "qScaleAndZeros must be 3-D [k / qGroupSize][weight rows][2]")
        n_groups = qinfo.size(0)
        _check(n_groups > 0 and k % n_groups == 0, "number of q-groups must divide k")
        _check(k // n_groups == q_group, "qScaleAndZeros.size(0) must equal k / qGroupSize")
        _check(qinfo.size(1) == wrows, "qScaleAndZeros.size(1) must equal the tile-padded weight rows")
        _check(qinfo.size(2) == 2, "qScaleAndZeros.size(2) must be 2")
        _check(qinfo.dtype == x.dtype, "qScaleAndZeros dtype must match the activations")
    if lut is not None:
        _check(lut.device == x.device, "int4DequantValues must be on the activations' device")
        _check(lut.dtype == x.dtype, "int4DequantValues dtype must match the activations")
        if lut.dim() == 1:
            _check(lut.size(0) == 16, "int4DequantValues must have 16 entries")
            qtype = TG_Q_ANY4_GLOBAL
        else:
            _check(lut.dim() == 2 and lut.size(0) == wrows and lut.size(1) == 16,
                   "row-wise int4DequantValues must be [weight rows (tile padded)][16]")
            qtype = TG_Q_ANY4_ROWWISE
        lut = lut.contiguous()
    qinfo = qinfo.contiguous()
    _check(k % 32 == 0 and k_tiles % inner == 0, "k must be a multiple of 32 and of innerKTiles * 16")
    if lut is not None and lut.data_ptr() % 16:
        lut = lut.clone()
    if not frag and m >= large_m_rows(wrows * k) and qtype != TG_Q_MX4 and k % 512 == 0 and (weight_on_right or w_format == _lib.TG_WFMT_ROWS):
        # OPT-IN (ANY4_LARGE_M_GEMM=library / ANY4_LARGE_M): dequantise in bounded row panels and hand the product to the GEMM library
        # (hipBLASLt behind torch.matmul).  The default keeps every call on this library's own kernels (tg_gemm_w4 -> plan 'tile').
        y = _library_gemm_w4(x, w, qinfo, lut, q_group, qtype, k, inner, wrows)
        bias = _take_bias(wrows, x)
        return y if bias is None else y + bias
    if frag:
        if m == 0 or get_numerics() == "reference":
            return None
        # zero-filled when the last 16-column tile is half used (wrows = 8 * odd): the kernel writes wrows columns
        alloc = torch.zeros if wrows % 16 else torch.empty
        y = alloc((m // 16, _cdiv(wrows, 16), 32, 8), dtype=x.dtype, device=x.device)
        bias = None
        layout = _lib.TG_LAYOUT_TC_A
    else:
        if x.data_ptr() % 16:
            x = x.clone()
        y = torch.empty((m, wrows), dtype=x.dtype, device=x.device)
        if m == 0:
            return y
        bias = _take_bias(wrows, x)
        layout = _lib.TG_LAYOUT_RM
    args = W4Gemm(
        x=x.data_ptr(), w=w.data_ptr(), qinfo=qinfo.data_ptr(), lut=(lut.data_ptr() if lut is not None else None),
        y=y.data_ptr(), m=m, wrows=wrows, k=k, group=q_group, qtype=qtype, dtype=_dt(x),
        w_on_right=1 if weight_on_right else 0, inner_k_tiles=inner, batch=1,
        numerics=_NUMERICS[get_numerics()], bias=(bias.data_ptr() if bias is not None else None),
        x_layout=layout, y_layout=layout, w_format=w_format,
    )
    # The planner's answer depends on the problem's shape only: asked once per (shape, layout, numerics), not once per call
    # (m = 1 latency path: one planner pass and no allocation when no scratch is needed).
    key = (m, wrows, k, q_group, qtype, args.dtype, args.w_on_right, inner, args.numerics, layout, bias is not None, args.w_format,
           _dev(x))
    ws_bytes = _WS_BYTES.get(key)
    if ws_bytes is None:
        ws_bytes = _L.tg_gemm_w4_workspace_bytes(ctypes.byref(args))
        # (an error code may come from THIS call's pointers -- a misaligned view -- and must not stick to the shape)
        if len(_WS_BYTES) < 4096 and (ws_bytes >= 0 or ws_bytes == _lib.TG_E_LAYOUT):
            _WS_BYTES[key] = ws_bytes
    if frag and ws_bytes == _lib.TG_E_LAYOUT:
        return None
    if ws_bytes < 0:
        _lib.check(ws_bytes, opname)
    if ws_bytes > 0:  # scratch from torch's caching allocator: stream-ordered like every other temporary of the op
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        args.workspace, args.workspace_bytes = ws.data_ptr(), ws_bytes
    _lib.check(_L.tg_gemm_w4(ctypes.byref(args), _dev(x), _stream(x)), opname)
    sink = getattr(_tls, "plan_sink", None)
    if sink is not None and not frag:
        # a caller (modules._PackedLinear) keeps the validated argument struct to re-issue the same launch with new x / y (/ scratch) pointers
        sink.append((W4Gemm.from_buffer_copy(args), x, (w, qinfo, lut, bias), opname, max(ws_bytes, 0)))
    return y



class LaunchPlan:
    """One validated row-major 4-bit GEMM launch of a module, re-issued with new activation / output pointers: the eager
    `forward` of Any4Linear / Int4Linear spends ~20 us in Python (35 precondition checks, the op dispatcher, building the argument
    struct) around a 5 us kernel; the checks only depend on the parameters and the activations' shape / dtype / device, which the
    plan pins.  Anything else (another shape, a re-assigned parameter, another numerics / weight-format setting, a
    non-contiguous or misaligned input) takes the full path again.

    Re-entrant like the reference's host functions (TinyGemm_int4.cu:41-42: no state, the current stream of the calling thread):
    the recorded struct is a template that is never written after construction; every host thread fills in x / y in ITS OWN copy
    (made once per thread), so two threads running the same module on two streams cannot see each other's pointers.  A launch
    that takes a workspace (split-K tile launches at 17 ... ~256 rows, activation pre-passes) gets a fresh one from torch's caching allocator
    per call like the full path does (stream-ordered, never shared between calls in flight).

    `try_run` is the whole eager hot path of a module (measured on MI355X, dev/host_path.py: 10.7 -> ~8 us per forward at 4096 x 4096,
    of which the HIP launch is 3-4 and torch.empty 1.5): no tuples built, no views, the parameters checked by POINTER (the struct points
    at their storage: an in-place update needs no new plan, a re-assigned or re-allocated parameter has a new pointer)."""

    __slots__ = ("args", "key", "m", "n", "k", "dtype", "device", "dev_index", "opname", "keep", "_per_thread", "ptrs", "numerics", "wformat", "attrs", "ws_bytes")

    def __init__(self, args, x, keep, opname, key, ws_bytes=0):
        self.args, self.key, self.opname, self.keep = args, key, opname, keep  # (keep: the tensors the struct points at stay alive)
        self.ws_bytes = ws_bytes
        self.m, self.n, self.k, self.dtype, self.device, self.dev_index = x.shape[0], args.wrows, x.shape[1], x.dtype, x.device, _dev(x)
        self._per_thread = {}   # thread id -> (that thread's private copy of the struct, its byref) (dict get / set are atomic under the GIL)
        self.ptrs = (args.w, args.qinfo, args.lut)
        self.numerics, self.wformat = get_numerics(), get_weight_format()
        self.attrs = None       # (set by the module: its kernel / group size / innerKTiles at recording time)

    def thread_args(self):
        """The calling thread's private copy of the argument struct (live threads never share an ident) and its ctypes reference."""
        tid = threading.get_ident()
        a = self._per_thread.get(tid)
        if a is None:
            if len(self._per_thread) >= 64:      # (threads come and go: do not grow without bound)
                self._per_thread.clear()
            c = W4Gemm.from_buffer_copy(self.args)
            a = self._per_thread[tid] = (c, ctypes.byref(c))
        return a

    def _launch(self, xp, y):
        a, ref = self.thread_args()
        a.x, a.y = xp, y.data_ptr()
        if self.ws_bytes:
            ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)   # (alive until the launch is enqueued: the allocator is stream-ordered)
            a.workspace = ws.data_ptr()
        rc = _L.tg_gemm_w4(ref, self.dev_index, _raw_stream(self.dev_index) if _raw_stream is not None else torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            _lib.check(rc, self.opname)
        return y

    def run(self, x):
        return self._launch(x.data_ptr(), torch.empty((self.m, self.n), dtype=self.dtype, device=self.device))

    def try_run(self, input, w, q, lut, attrs):
        """The recorded launch on `input` ([..., k], any leading shape with m rows in all) if everything the plan pins still holds, else
        None (the caller takes the full path and records again).  w / q / lut: the module's parameter tensors NOW."""
        if input.dtype is not self.dtype or attrs != self.attrs or not input.is_contiguous():
            return None
        shp = input.shape
        if shp[-1] != self.k or input.numel() != self.m * self.k or input.device != self.device:
            return None
        xp = input.data_ptr()
        pw, pq, pl = self.ptrs
        if (xp & 15) or w.data_ptr() != pw or q.data_ptr() != pq or (pl is not None and (lut is None or lut.data_ptr() != pl)):
            return None
        if get_numerics() != self.numerics or get_weight_format() != self.wformat:
            return None
        return self._launch(xp, torch.empty((*shp[:-1], self.n), dtype=self.dtype, device=self.device))


def record_plan(fn, x, key, params=()):
    """Runs fn(x) (a functional of this module that ends in ONE row-major 4-bit GEMM) and returns (y, LaunchPlan or None).
    `params`: the module's own parameter tensors -- a plan is only kept when the recorded struct points at THEM (not at a
    contiguous / re-aligned copy the op made, which a later in-place update of the parameter would not reach)."""
    prev = getattr(_tls, "plan_sink", None)
    _tls.plan_sink = sink = []       # thread-local: another thread's launches never land in this recording
    try:
        y = fn(x)
    finally:
        _tls.plan_sink = prev
    if len(sink) != 1 or sink[0][1].data_ptr() != x.data_ptr() or tuple(y.shape) != (x.shape[0], sink[0][0].wrows):
        return y, None
    args, _, keep, opname, ws_bytes = sink[0]
    own = {t.data_ptr() for t in params if t is not None}
    if params and any(p is not None and p not in own for p in (args.w, args.qinfo, args.lut)):
        return y, None
    return y, LaunchPlan(args, x, keep, opname, key, ws_bytes)


def w4_linear_fused(x, w, q_group, qinfo, lut=None, *, residual=None, norm_weight=None, norm_eps=1e-5, swiglu=False, out=None):
    """The row-major 4-bit GEMM y = x . dequant(W)^T (weights on the right: `w` in the Bint4 layout, as Any4Linear / Int4Linear
    hold it) with stages of a decoder layer fused into the SAME launch (include/tinygemm_hip.h, ABI 5) -- not part of the
    reference's op surface; the decode harness (any4_amd/decode.py) calls it instead of Linear + glue kernel:

      norm_weight [k]        the activations pass through LlamaRMSNorm in the kernel's staging: x' = rmsnorm(x, eps) * norm_weight
      residual    [m][n]     y = RNE16(RNE16(acc) + residual)   (the residual stream; may be `out` itself: updated in place)
      swiglu                 the weight rows come in blocks of 8 gate + 8 up rows and y is [m][n / 2] = silu(gate) * up
      out                    where to write y (default: a new tensor)

    Returns y, or None when the library has no kernel with these stages for this problem (TG_E_FUSION: e.g. reference numerics,
    k % 2048 != 0, an activation block too large to stage on chip): the caller then runs the stage as its own launch."""
    _check(x.dim() == 2 and x.is_contiguous() and x.dtype in _F16_TYPES, "activations must be a contiguous 2-D bf16 / fp16 matrix")
    _check(w.dim() == 4 and w.dtype == torch.int32 and w.is_contiguous() and w.size(2) == 32, "weights must be a contiguous Bint4 tensor")
    inner, wrows = w.size(3) * 2, w.size(0) * 8
    m, k = x.shape
    _check(w.size(1) * inner * 16 == k, "weights: k super-tiles do not match the activations' k")
    _check(m > 0, "activations must have at least one row")
    _check(inner in (2, 4, 8), "Bint4 weights: innermost dim must be 1, 2 or 4")
    _check(q_group in (32, 64, 128, 256) and k % q_group == 0, "qGroupSize must be 32, 64, 128 or 256 and divide k")
    _check(w.device == x.device, "weights must be on the activations' device")
    # the operand shapes the kernels index by (a mismatched tensor would be read out of bounds, not rejected): the checks of _w4_rm
    qtype = TG_Q_INT4
    if lut is not None:
        _check(lut.dtype == x.dtype and lut.is_contiguous() and lut.device == x.device, "LUT must be contiguous, of the activations' dtype, on their device")
        _check((lut.dim() == 1 and lut.size(0) == 16) or (lut.dim() == 2 and lut.size(0) == wrows and lut.size(1) == 16),
               "int4DequantValues must be [16] or [weight rows (tile padded)][16]")
        qtype = TG_Q_ANY4_GLOBAL if lut.dim() == 1 else TG_Q_ANY4_ROWWISE
        if lut.data_ptr() % 16:
            lut = lut.clone()
    elif qinfo.dtype == torch.uint8:
        qtype = TG_Q_MX4
    _check(qinfo.is_contiguous() and qinfo.device == x.device, "quantization info must be contiguous on the activations' device")
    if qtype == TG_Q_MX4:
        _check(x.dtype == torch.bfloat16, "mx4 supports bfloat16 activations only")
        _check(qinfo.dim() == 2 and qinfo.size(0) == wrows and qinfo.size(1) == k // q_group, "mx4Exponents must be [weight rows (tile padded)][k / qGroupSize]")
    else:
        _check(qinfo.dim() == 3 and qinfo.dtype == x.dtype and tuple(qinfo.shape) == (k // q_group, wrows, 2),
               "qScaleAndZeros must be [k / qGroupSize][weight rows (tile padded)][2] of the activations' dtype")
    ycols = wrows // 2 if swiglu else wrows
    if out is None:
        out = torch.empty((m, ycols), dtype=x.dtype, device=x.device)
    _check(out.shape == (m, ycols) and out.dtype == x.dtype and out.is_contiguous() and out.device == x.device,
           "out must be a contiguous [m][n] tensor of the activations' dtype on their device")
    if residual is not None:
        _check(residual.dtype == x.dtype and residual.dim() == 2 and residual.shape[0] == m and residual.shape[1] >= wrows
               and residual.stride(1) == 1 and residual.device == x.device, "residual must be [m][>= n] with unit inner stride on the activations' device")
    if norm_weight is not None:
        _check(norm_weight.dtype == x.dtype and norm_weight.numel() == k and norm_weight.is_contiguous() and norm_weight.device == x.device,
               "norm_weight must be [k] of the activations' dtype on their device")
    args = W4Gemm(
        x=x.data_ptr(), w=w.data_ptr(), qinfo=qinfo.data_ptr(), lut=(lut.data_ptr() if lut is not None else None), y=out.data_ptr(),
        m=m, wrows=wrows, k=k, group=q_group, qtype=qtype, dtype=_dt(x), w_on_right=1, inner_k_tiles=inner, batch=1,
        numerics=_NUMERICS[get_numerics()],
        bias=(residual.data_ptr() if residual is not None else None),
        bias_row_stride=(residual.stride(0) if residual is not None else 0),
        norm_weight=(norm_weight.data_ptr() if norm_weight is not None else None), norm_eps=float(norm_eps),
        epilogue=_lib.TG_EPI_SWIGLU if swiglu else _lib.TG_EPI_NONE,
    )
    ws_bytes = _L.tg_gemm_w4_workspace_bytes(ctypes.byref(args))
    if ws_bytes == _lib.TG_E_FUSION:
        return None
    if ws_bytes > 0:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        args.workspace, args.workspace_bytes = ws.data_ptr(), ws_bytes
    rc = _L.tg_gemm_w4(ctypes.byref(args), _dev(x), _stream(x))
    if rc == _lib.TG_E_FUSION:
        return None
    _lib.check(rc, "w4_linear_fused")
    return out


def _w4_tc(A, B, q_group, qinfo, lut, qtype, weight_on_right, opname):
    """Fragment-order activations / output (TinyGemm_int4.cu:28-292).  Implemented as
    un-layout -> row-major GEMM -> re-layout, all on the device."""
    with fused_bias(torch.empty(0)):  # an offered bias is for a row-major result: hide it from the inner call
        return _w4_tc_impl(A, B, q_group, qinfo, lut, qtype, weight_on_right, opname)


def _w4_tc_impl(A, B, q_group, qinfo, lut, qtype, weight_on_right, opname):
    _check(A.dim() == 4 and A.is_contiguous() and A.size(2) == 32, "A must be a contiguous 4-D tensor-core layout tensor")
    _check(B.dim() == 4 and B.is_contiguous() and B.size(2) == 32, "B must be a contiguous 4-D tensor-core layout tensor")
    if weight_on_right:
        _check(A.size(3) == 8, "activations (A layout) must have innermost dim 8")
        _check(B.size(3) in (1, 2, 4) and B.dtype == torch.int32, "weights (Bint4 layout) must be int32 with innermost dim 1, 2 or 4")
        k_tiles_a, k_tiles_b = A.size(1), B.size(1) * B.size(3) * 2
        _check(k_tiles_a == k_tiles_b, "A and B disagree on k")
        m_pad, k = A.size(0) * 16, k_tiles_a * 16
        # native: the pair-table kernels read the fragment-order activations and write the fragment-order output themselves
        # (MatrixLayoutA.cuh:211-373 in the reference): one launch instead of convert -> GEMM -> convert
        y = _w4_rm(_FragX(A), B, q_group, qinfo, lut, qtype, True, opname, frag=True)
        if y is not None:
            return y
        x = convert_matrix_from_m16n8k16_A_layout(A, m_pad, k)
        y = _w4_rm(x, B, q_group, qinfo, lut, qtype, True, opname)       # [m_pad][n_pad]
        return convert_matrix_to_m16n8k16_A_layout(y, 1)                 # [mTiles][ceil(nTiles/2)][32][8]
    _check(A.size(3) in (1, 2, 4) and A.dtype == torch.int32, "weights (Aint4 layout) must be int32 with innermost dim 1, 2 or 4")
    _check(B.size(3) in (4, 8), "activations (B layout) must have innermost dim 4 or 8")
    b_inner = B.size(3) // 4
    k_tiles_a, k_tiles_b = A.size(1) * A.size(3), B.size(1) * b_inner
    # (the reference's Aint4 tensor covers k_tiles_a k-tiles, the native one -- a Bint4 tensor, `aside_format` -- twice that)
    _check(k_tiles_a == k_tiles_b or (2 * k_tiles_a == k_tiles_b and A.size(3) * 2 == _rows_inner(k_tiles_b * 16)), "A and B disagree on k")
    n_pad, k = B.size(0) * 8, k_tiles_b * 16
    x = convert_matrix_from_m16n8k16_B_layout(B, n_pad, k)
    y = _w4_rm(A, x, q_group, qinfo, lut, qtype, False, opname)          # [n_pad][m_pad]
    return convert_matrix_to_m16n8k16_B_layout(y, b_inner)               # [nTiles][ceil(mTiles/I)][32][4 I]


def tinygemm_y_f16RM_x_f16RM_w_int4TC(A, B, qGroupSize, qScaleAndZeros, weightOnRight):
    return _w4_rm(A, B, qGroupSize, qScaleAndZeros, None, TG_Q_INT4, weightOnRight, "tinygemm_y_f16RM_x_f16RM_w_int4TC")


def tinygemm_y_f16TC_x_f16TC_w_int4TC(A, B, qGroupSize, qScaleAndZeros, weightOnRight):
    return _w4_tc(A, B, qGroupSize, qScaleAndZeros, None, TG_Q_INT4, weightOnRight, "tinygemm_y_f16TC_x_f16TC_w_int4TC")


def tinygemm_y_f16RM_x_f16RM_w_any4TC(A, B, qGroupSize, qScaleAndZeros, int4DequantValues, weightOnRight):
    return _w4_rm(A, B, qGroupSize, qScaleAndZeros, int4DequantValues, TG_Q_ANY4_ROWWISE, weightOnRight,
                  "tinygemm_y_f16RM_x_f16RM_w_any4TC")


def tinygemm_y_f16TC_x_f16TC_w_any4TC(A, B, qGroupSize, qScaleAndZeros, int4DequantValues, weightOnRight):
    return _w4_tc(A, B, qGroupSize, qScaleAndZeros, int4DequantValues, TG_Q_ANY4_ROWWISE, weightOnRight,
                  "tinygemm_y_f16TC_x_f16TC_w_any4TC")


def tinygemm_y_f16RM_x_f16RM_w_mx4TC(A, B, qGroupSize, mx4Exponents, weightOnRight):
    return _w4_rm(A, B, qGroupSize, mx4Exponents, None, TG_Q_MX4, weightOnRight, "tinygemm_y_f16RM_x_f16RM_w_mx4TC")


def tinygemm_y_f16TC_x_f16TC_w_mx4TC(A, B, qGroupSize, mx4Exponents, weightOnRight):
    return _w4_tc(A, B, qGroupSize, mx4Exponents, None, TG_Q_MX4, weightOnRight, "tinygemm_y_f16TC_x_f16TC_w_mx4TC")


# ---------------------------------------------------------------------------------------------
# 16-bit weights (TinyGemm_bf16.cu)
# ---------------------------------------------------------------------------------------------

def tinygemm_y_f16RM_x_f16RM_w_f16TC(A, B, weightOnRight):
    _check(A.device == B.device, "A and B must be on the same device")
    if weightOnRight:
        x, w = A, B
        _check(w.dim() == 4 and w.is_contiguous() and w.size(2) == 32 and w.size(3) in (4, 8), "weights must be in B layout")
        inner, wrows = w.size(3) // 4, w.size(0) * 8
    else:
        w, x = A, B
        _check(w.dim() == 4 and w.is_contiguous() and w.size(2) == 32 and w.size(3) == 8, "weights must be in A layout")
        inner, wrows = 1, w.size(0) * 16
    _check(x.dim() == 2 and x.is_contiguous(), "activations must be a contiguous 2-D matrix")
    _check(x.dtype in _F16_TYPES and w.dtype == x.dtype, "activations and weights must share a 16-bit float dtype")
    m, k = x.shape
    _check(w.size(1) == _cdiv(_cdiv(k, 16), inner), "weights: k tiles do not match the activations' k")
    _check(k % 32 == 0, "k must be a multiple of 32")
    if x.data_ptr() % 16:
        x = x.clone()
    y = torch.empty((m, wrows), dtype=x.dtype, device=x.device)
    if m:
        _lib.check(_L.tg_gemm_f16(x.data_ptr(), w.data_ptr(), y.data_ptr(), m, wrows, k, _dt(x), 1 if weightOnRight else 0,
                                  inner, _dev(x), _stream(x)), "tinygemm_y_f16RM_x_f16RM_w_f16TC")
    return y


def tinygemm_y_f16TC_x_f16TC_w_f16TC(A, B, weightOnRight):
    _check(A.dim() == 4 and B.dim() == 4, "A and B must be 4-D tensor-core layout tensors")
    if weightOnRight:
        _check(A.size(3) == 8, "activations (A layout) must have innermost dim 8")
        m_pad, k = A.size(0) * 16, A.size(1) * 16
        x = convert_matrix_from_m16n8k16_A_layout(A, m_pad, k)
        y = tinygemm_y_f16RM_x_f16RM_w_f16TC(x, B, True)
        return convert_matrix_to_m16n8k16_A_layout(y, 1)
    b_inner = B.size(3) // 4
    n_pad, k = B.size(0) * 8, B.size(1) * b_inner * 16
    x = convert_matrix_from_m16n8k16_B_layout(B, n_pad, k)
    y = tinygemm_y_f16RM_x_f16RM_w_f16TC(A, x, False)
    return convert_matrix_to_m16n8k16_B_layout(y, b_inner)


# ---------------------------------------------------------------------------------------------
# int8 weights
# ---------------------------------------------------------------------------------------------

def convert_matrix_to_m16n8k16_Bint8_layout(t: torch.Tensor, innerKTiles: int) -> torch.Tensor:
    """TinyGemmConvertB.cu:415-465: int32 [n][k] byte codes -> [ceil(n/8)][k/(16 I)][32][I]."""
    _check(t.dim() == 2 and t.dtype == torch.int32 and t.is_contiguous(), "Bint8 layout: input must be a contiguous 2-D int32 matrix")
    _check(innerKTiles in (1, 2, 4), "Bint8 layout: innerKTiles must be 1, 2 or 4")
    n, k = t.shape
    _check(k % (innerKTiles * 16) == 0, "Bint8 layout: k must be a multiple of innerKTiles * 16")
    out = torch.empty((_cdiv(n, 8), k // (innerKTiles * 16), 32, innerKTiles), dtype=torch.int32, device=t.device)
    _lib.check(_L.tg_convert_to_Bint8(t.data_ptr(), n, k, innerKTiles, out.data_ptr(), _dev(t), _stream(t)),
               "convert_matrix_to_m16n8k16_Bint8_layout")
    return out


def convert_matrix_to_m16n8k16_Aint8_layout(t: torch.Tensor, innerKTiles: int) -> torch.Tensor:
    """TinyGemmConvertA.cu:400-440: int32 [m][k] byte codes -> [ceil(m/16)][ceil(ceil(k/16)/I)][32][2 I]."""
    _check(t.dim() == 2 and t.dtype == torch.int32 and t.is_contiguous(), "Aint8 layout: input must be a contiguous 2-D int32 matrix")
    _check(innerKTiles in (1, 2), "Aint8 layout: innerKTiles must be 1 or 2")
    m, k = t.shape
    out = torch.empty((_cdiv(m, 16), _cdiv(_cdiv(k, 16), innerKTiles), 32, 2 * innerKTiles), dtype=torch.int32, device=t.device)
    _lib.check(_L.tg_convert_to_Aint8(t.data_ptr(), m, k, innerKTiles, out.data_ptr(), _dev(t), _stream(t)),
               "convert_matrix_to_m16n8k16_Aint8_layout")
    return out


def tinygemm_y_f16RM_x_f16RM_w_int8TC(A, B, qGroupSize, qScaleAndZeros, weightOnRight):
    """Row-major activations / output with int8 weights (TinyGemm_int8.cu:216-399)."""
    opname = "tinygemm_y_f16RM_x_f16RM_w_int8TC"
    _check(A.device == B.device and A.device == qScaleAndZeros.device, "A, B and qScaleAndZeros must be on the same device")
    if weightOnRight:
        x, w = A, B
        _check(x.dim() == 2 and x.is_contiguous(), "activations must be a contiguous 2-D matrix")
        _check(w.dim() == 4 and w.dtype == torch.int32 and w.is_contiguous(), "weights must be a contiguous 4-D int32 tensor")
        inner = w.size(3)
        _check(inner in (1, 2, 4), "Bint8 weights: innermost dim must be 1, 2 or 4")
        wrows = w.size(0) * 8
    else:
        w, x = A, B
        _check(w.dim() == 4 and w.dtype == torch.int32 and w.is_contiguous(), "weights must be a contiguous 4-D int32 tensor")
        _check(x.dim() == 2 and x.is_contiguous(), "activations must be a contiguous 2-D matrix")
        _check(w.size(3) % 2 == 0, "Aint8 weights: innermost dim must be even")
        inner = w.size(3) // 2
        _check(inner in (1, 2), "Aint8 weights: innermost dim must be 2 or 4")
        wrows = w.size(0) * 16
    m, k = x.shape
    k_tiles = _cdiv(k, 16)
    _check(w.size(1) == _cdiv(k_tiles, inner), "weights: k super-tiles do not match the activations' k")
    _check(w.size(2) == 32, "weights: dim 2 must be 32")
    _check(x.dtype in _F16_TYPES, "activation dtype must be bfloat16 or float16")
    _check(qGroupSize in (32, 64, 128, 256), "qGroupSize must be 32, 64, 128 or 256")
    qinfo = qScaleAndZeros
    _check(qinfo.dim() == 3, "qScaleAndZeros must be 3-D [k / qGroupSize][weight rows][2]")
    _check(k_tiles * 16 >= qGroupSize and (k_tiles * 16) % qGroupSize == 0, "qGroupSize must divide k")
    _check(qinfo.size(0) == (k_tiles * 16) // qGroupSize, "qScaleAndZeros.size(0) must equal k / qGroupSize")
    _check(qinfo.size(1) == wrows, "qScaleAndZeros.size(1) must equal the tile-padded weight rows")
    _check(qinfo.size(2) == 2, "qScaleAndZeros.size(2) must be 2")
    _check(qinfo.dtype == x.dtype, "qScaleAndZeros dtype must match the activations")
    _check(k % 32 == 0 and k_tiles % inner == 0, "k must be a multiple of 32 and of innerKTiles * 16")
    qinfo = qinfo.contiguous()
    y = torch.empty((m, wrows), dtype=x.dtype, device=x.device)
    if m == 0:
        return y
    if x.data_ptr() % 16:   # (a view at an odd element offset: the kernels load the activations in 16-byte pieces)
        x = x.clone()
    bias = _take_bias(wrows, x)
    args = W4Gemm(x=x.data_ptr(), w=w.data_ptr(), qinfo=qinfo.data_ptr(), lut=None, y=y.data_ptr(), m=m, wrows=wrows, k=k,
                  group=qGroupSize, qtype=_lib.TG_Q_INT8, dtype=_dt(x), w_on_right=1 if weightOnRight else 0,
                  inner_k_tiles=inner, batch=1, bias=(bias.data_ptr() if bias is not None else None))
    # many activation rows of innerKTiles-2 words: the tile GEMM's int8 flavour, split-K with a scratch from the caching allocator
    key = ("w8", m, wrows, k, qGroupSize, args.dtype, args.w_on_right, inner, bias is not None, _dev(x))
    ws_bytes = _WS_BYTES.get(key)
    if ws_bytes is None:
        ws_bytes = _L.tg_gemm_w8_workspace_bytes(ctypes.byref(args))
        if len(_WS_BYTES) < 4096 and ws_bytes >= 0:
            _WS_BYTES[key] = ws_bytes
    if ws_bytes > 0:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        args.workspace, args.workspace_bytes = ws.data_ptr(), ws_bytes
    _lib.check(_L.tg_gemm_w8(ctypes.byref(args), _dev(x), _stream(x)), opname)
    return y


def tinygemm_y_f16TC_x_f16TC_w_int8TC(A, B, qGroupSize, qScaleAndZeros, weightOnRight):
    """Fragment-order activations / output (TinyGemm_int8.cu:23-214): un-layout -> row-major GEMM -> re-layout."""
    _check(A.dim() == 4 and A.is_contiguous() and A.size(2) == 32, "A must be a contiguous 4-D tensor-core layout tensor")
    _check(B.dim() == 4 and B.is_contiguous() and B.size(2) == 32, "B must be a contiguous 4-D tensor-core layout tensor")
    if weightOnRight:
        _check(A.size(3) == 8, "activations (A layout) must have innermost dim 8")
        _check(B.size(3) in (1, 2, 4) and B.dtype == torch.int32, "weights (Bint8 layout) must be int32 with innermost dim 1, 2 or 4")
        k_tiles_a, k_tiles_b = A.size(1), B.size(1) * B.size(3)
        _check(k_tiles_a == k_tiles_b, "A and B disagree on k")
        m_pad, k = A.size(0) * 16, k_tiles_a * 16
        x = convert_matrix_from_m16n8k16_A_layout(A, m_pad, k)
        y = tinygemm_y_f16RM_x_f16RM_w_int8TC(x, B, qGroupSize, qScaleAndZeros, True)
        return convert_matrix_to_m16n8k16_A_layout(y, 1)
    _check(A.size(3) in (2, 4) and A.dtype == torch.int32, "weights (Aint8 layout) must be int32 with innermost dim 2 or 4")
    _check(B.size(3) in (4, 8), "activations (B layout) must have innermost dim 4 or 8")
    b_inner = B.size(3) // 4
    k_tiles_a, k_tiles_b = A.size(1) * (A.size(3) // 2), B.size(1) * b_inner
    _check(k_tiles_a == k_tiles_b, "A and B disagree on k")
    n_pad, k = B.size(0) * 8, k_tiles_b * 16
    x = convert_matrix_from_m16n8k16_B_layout(B, n_pad, k)
    y = tinygemm_y_f16RM_x_f16RM_w_int8TC(A, x, qGroupSize, qScaleAndZeros, False)
    return convert_matrix_to_m16n8k16_B_layout(y, b_inner)


_IMPLS = {
    "convert_matrix_to_m16n8k16_A_layout": convert_matrix_to_m16n8k16_A_layout,
    "convert_matrix_to_m16n8k16_Aint4_layout": convert_matrix_to_m16n8k16_Aint4_layout,
    "convert_matrix_to_m16n8k16_Aint8_layout": convert_matrix_to_m16n8k16_Aint8_layout,
    "convert_matrix_from_m16n8k16_A_layout": convert_matrix_from_m16n8k16_A_layout,
    "convert_matrix_to_m16n8k16_B_layout": convert_matrix_to_m16n8k16_B_layout,
    "convert_matrix_to_m16n8k16_Bint4_layout": convert_matrix_to_m16n8k16_Bint4_layout,
    "convert_matrix_to_m16n8k16_Bint8_layout": convert_matrix_to_m16n8k16_Bint8_layout,
    "convert_matrix_from_m16n8k16_B_layout": convert_matrix_from_m16n8k16_B_layout,
    "tinygemm_y_f16TC_x_f16TC_w_int4TC": tinygemm_y_f16TC_x_f16TC_w_int4TC,
    "tinygemm_y_f16RM_x_f16RM_w_int4TC": tinygemm_y_f16RM_x_f16RM_w_int4TC,
    "tinygemm_y_f16TC_x_f16TC_w_any4TC": tinygemm_y_f16TC_x_f16TC_w_any4TC,
    "tinygemm_y_f16RM_x_f16RM_w_any4TC": tinygemm_y_f16RM_x_f16RM_w_any4TC,
    "tinygemm_y_f16TC_x_f16TC_w_mx4TC": tinygemm_y_f16TC_x_f16TC_w_mx4TC,
    "tinygemm_y_f16RM_x_f16RM_w_mx4TC": tinygemm_y_f16RM_x_f16RM_w_mx4TC,
    "tinygemm_y_f16TC_x_f16TC_w_int8TC": tinygemm_y_f16TC_x_f16TC_w_int8TC,
    "tinygemm_y_f16RM_x_f16RM_w_int8TC": tinygemm_y_f16RM_x_f16RM_w_int8TC,
    "tinygemm_y_f16TC_x_f16TC_w_f16TC": tinygemm_y_f16TC_x_f16TC_w_f16TC,
    "tinygemm_y_f16RM_x_f16RM_w_f16TC": tinygemm_y_f16RM_x_f16RM_w_f16TC,
    "tinygemm_dequant_int4": tinygemm_dequant_int4,
}

_library = None


def register() -> None:
    """Define the schemas and bind the CUDA(=ROCm)-key implementations.  Idempotent."""
    global _library
    if _library is not None:
        return
    lib = torch.library.Library(NAMESPACE, "DEF")
    for name, schema in SCHEMAS.items():
        lib.define(name + schema)
        lib.impl(name, _IMPLS[name], "CUDA")
    _library = lib


register()
