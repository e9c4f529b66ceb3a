"""Functional API with the reference's names and argument order
(tinygemm_lib/functional.py:10-259 in facebookresearch/any4).

Name grammar, as in the reference:  linear_y_<out layout>_x_<act layout>_W_<w fmt>  puts the weight on
the right / B side (weightOnRight=True);  linear_y_..._W_..._x_...  puts it on the left / A side.
`RM` = row-major activations, `TC` = activations pre-shuffled into m16n8k16 fragment order.
`reshape_weight=True` packs `w_int32` ([n][k] codes) on every call; modules pack once and pass False.

Every function bottoms out in torch.ops.tinygemm.* (any4_amd/ops.py -> HIP C ABI).
"""
from __future__ import annotations

import torch

from . import ops as _ops  # noqa: F401  (registers torch.ops.tinygemm)

_T = torch.ops.tinygemm

_W_LEFT_INNER_K = (1, 2, 4)   # Aint4 / A layouts
_W_RIGHT_INNER_K = (2, 4, 8)  # Bint4 layout


def valid_tinygemm_kernel_call(functional_api, w_inner_k):
    """Which (api, w_inner_k) pairs the any4 kernels accept (reference functional.py:10-18).
    Returns True or None, like the reference."""
    right = functional_api in ("linear_y_f16RM_x_f16RM_W_any4TC", "linear_y_f16TC_x_f16TC_W_any4TC")
    left = functional_api in ("linear_y_f16TC_W_any4TC_x_f16TC", "linear_y_f16RM_W_any4TC_x_f16RM")
    if (right and w_inner_k in _W_RIGHT_INNER_K) or (left and w_inner_k in _W_LEFT_INNER_K):
        return True
    return None


# -- helpers ---------------------------------------------------------------------------------------

def _w_right(w, inner_k, reshape, kind):
    """weight -> B-side operand"""
    if not reshape:
        return w
    if kind == "int4":
        return _T.convert_matrix_to_m16n8k16_Bint4_layout(w, inner_k)
    if kind == "int8":
        return _T.convert_matrix_to_m16n8k16_Bint8_layout(w, inner_k)
    return _T.convert_matrix_to_m16n8k16_B_layout(w, inner_k)


def _w_left(w, inner_k, reshape, kind):
    """weight -> A-side operand"""
    if not reshape:
        return w
    if kind == "int4":
        return _T.convert_matrix_to_m16n8k16_Aint4_layout(w, inner_k)
    if kind == "int8":
        return _T.convert_matrix_to_m16n8k16_Aint8_layout(w, inner_k)
    return _T.convert_matrix_to_m16n8k16_A_layout(w, inner_k)


# -- uniform int4 ------------------------------------------------------------------------------------

def linear_y_f16TC_x_f16TC_W_int4TC(x, w_int32, w_scales_and_zeros, q_group, w_inner_k=4, x_inner_k=1, reshape_weight=True):
    xa = _T.convert_matrix_to_m16n8k16_A_layout(x, x_inner_k)
    wb = _w_right(w_int32, w_inner_k, reshape_weight, "int4")
    y2 = _T.tinygemm_y_f16TC_x_f16TC_w_int4TC(xa, wb, q_group, w_scales_and_zeros, True)
    return _T.convert_matrix_from_m16n8k16_A_layout(y2, x.size(0), w_int32.size(0))


def linear_y_f16TC_W_int4TC_x_f16TC(x, w_int32, w_scales_and_zeros, q_group, w_inner_k=4, x_inner_k=1, reshape_weight=True):
    wa = _w_left(w_int32, w_inner_k, reshape_weight, "int4")
    xb = _T.convert_matrix_to_m16n8k16_B_layout(x, x_inner_k)
    y2 = _T.tinygemm_y_f16TC_x_f16TC_w_int4TC(wa, xb, q_group, w_scales_and_zeros, False)
    return _T.convert_matrix_from_m16n8k16_B_layout(y2, x.size(0), w_int32.size(0))


def linear_y_f16RM_x_f16RM_W_int4TC(x, w_int32, w_scales_and_zeros, q_group, w_inner_k=4, reshape_weight=True):
    wb = _w_right(w_int32, w_inner_k, reshape_weight, "int4")
    return _T.tinygemm_y_f16RM_x_f16RM_w_int4TC(x, wb, q_group, w_scales_and_zeros, True)


def linear_y_f16RM_W_int4TC_x_f16RM(x, w_int32, w_scales_and_zeros, q_group, w_inner_k=4, reshape_weight=True):
    wa = _w_left(w_int32, w_inner_k, reshape_weight, "int4")
    return _T.tinygemm_y_f16RM_x_f16RM_w_int4TC(wa, x, q_group, w_scales_and_zeros, False)


# -- int8 (byte codes, value = byte - 128; tinygemm_lib/functional.py:86-137) -----------------------

def linear_y_f16TC_x_f16TC_W_int8TC(x, w_int32, w_scales_and_zeros, q_group, w_inner_k=4, reshape_weight=True):
    xa = _T.convert_matrix_to_m16n8k16_A_layout(x, 1)
    wb = _w_right(w_int32, w_inner_k, reshape_weight, "int8")
    y2 = _T.tinygemm_y_f16TC_x_f16TC_w_int8TC(xa, wb, q_group, w_scales_and_zeros, True)
    return _T.convert_matrix_from_m16n8k16_A_layout(y2, x.shape[0], w_int32.shape[0])


def linear_y_f16TC_W_int8TC_x_f16TC(x, w_int32, w_scales_and_zeros, q_group, w_inner_k=4, x_inner_k=1, reshape_weight=True):
    wa = _w_left(w_int32, w_inner_k, reshape_weight, "int8")
    xb = _T.convert_matrix_to_m16n8k16_B_layout(x, x_inner_k)
    y2 = _T.tinygemm_y_f16TC_x_f16TC_w_int8TC(wa, xb, q_group, w_scales_and_zeros, False)
    return _T.convert_matrix_from_m16n8k16_B_layout(y2, x.shape[0], x.shape[1])


def linear_y_f16RM_x_f16RM_W_int8TC(x, w_int32, w_scales_and_zeros, q_group, w_inner_k=4, reshape_weight=True):
    wb = _w_right(w_int32, w_inner_k, reshape_weight, "int8")
    return _T.tinygemm_y_f16RM_x_f16RM_w_int8TC(x, wb, q_group, w_scales_and_zeros, True)


def linear_y_f16RM_W_int8TC_x_f16RM(x, w_int32, w_scales_and_zeros, q_group, w_inner_k=4, reshape_weight=True):
    wa = _w_left(w_int32, w_inner_k, reshape_weight, "int8")
    return _T.tinygemm_y_f16RM_x_f16RM_w_int8TC(wa, x, q_group, w_scales_and_zeros, False)


# -- any4 (LUT: 1-D = one table for the matrix, 2-D = one table per weight row) ----------------------

def linear_y_f16TC_x_f16TC_W_any4TC(x, w_int32, w_lut, w_scales_and_zeros, q_group, w_inner_k=4, x_inner_k=1, reshape_weight=True):
    xa = _T.convert_matrix_to_m16n8k16_A_layout(x, x_inner_k)
    wb = _w_right(w_int32, w_inner_k, reshape_weight, "int4")
    y2 = _T.tinygemm_y_f16TC_x_f16TC_w_any4TC(xa, wb, q_group, w_scales_and_zeros, w_lut, True)
    return _T.convert_matrix_from_m16n8k16_A_layout(y2, x.size(0), w_int32.size(0))


def linear_y_f16TC_W_any4TC_x_f16TC(x, w_int32, w_lut, w_scales_and_zeros, q_group, w_inner_k=4, x_inner_k=1, reshape_weight=True):
    wa = _w_left(w_int32, w_inner_k, reshape_weight, "int4")
    xb = _T.convert_matrix_to_m16n8k16_B_layout(x, x_inner_k)
    y2 = _T.tinygemm_y_f16TC_x_f16TC_w_any4TC(wa, xb, q_group, w_scales_and_zeros, w_lut, False)
    return _T.convert_matrix_from_m16n8k16_B_layout(y2, x.size(0), w_int32.size(0))


def linear_y_f16RM_x_f16RM_W_any4TC(x, w_int32, w_lut, w_scales_and_zeros, q_group, w_inner_k=4, reshape_weight=True):
    wb = _w_right(w_int32, w_inner_k, reshape_weight, "int4")
    return _T.tinygemm_y_f16RM_x_f16RM_w_any4TC(x, wb, q_group, w_scales_and_zeros, w_lut, True)


def linear_y_f16RM_W_any4TC_x_f16RM(x, w_int32, w_lut, w_scales_and_zeros, q_group, w_inner_k=4, reshape_weight=True):
    wa = _w_left(w_int32, w_inner_k, reshape_weight, "int4")
    return _T.tinygemm_y_f16RM_x_f16RM_w_any4TC(wa, x, q_group, w_scales_and_zeros, w_lut, False)


# -- un-quantised 16-bit weights -----------------------------------------------------------------------

def linear_y_f16TC_x_f16TC_W_f16TC(x, w, w_inner_k=4, reshape_weight=True):
    xa = _T.convert_matrix_to_m16n8k16_A_layout(x, 1)
    wb = _w_right(w, w_inner_k, reshape_weight, "f16")
    y2 = _T.tinygemm_y_f16TC_x_f16TC_w_f16TC(xa, wb, True)
    return _T.convert_matrix_from_m16n8k16_A_layout(y2, x.shape[0], w.shape[0])


def linear_y_f16TC_W_f16TC_x_f16TC(x, w, x_inner_k=4, reshape_weight=True):
    wa = _w_left(w, 1, reshape_weight, "f16")
    xb = _T.convert_matrix_to_m16n8k16_B_layout(x, x_inner_k)
    y2 = _T.tinygemm_y_f16TC_x_f16TC_w_f16TC(wa, xb, False)
    return _T.convert_matrix_from_m16n8k16_B_layout(y2, x.shape[0], w.shape[0])


def linear_y_f16RM_x_f16RM_W_f16TC(x, w, w_inner_k=4, reshape_weight=True):
    wb = _w_right(w, w_inner_k, reshape_weight, "f16")
    return _T.tinygemm_y_f16RM_x_f16RM_w_f16TC(x, wb, True)


def linear_y_f16RM_W_f16TC_x_f16RM(x, w, w_inner_k=4, reshape_weight=True):
    wa = _w_left(w, w_inner_k, reshape_weight, "f16")
    return _T.tinygemm_y_f16RM_x_f16RM_w_f16TC(wa, x, False)
