"""Input generators of the kernel tests, restated from the reference
(tinygemm_lib/utils.py:27-232 in facebookresearch/any4): asymmetric per-group int4 quantisation and the
MX4 (fp4-e2m1 + e8m0 shared exponent) quantiser.  Plain torch; runs on any device."""
from __future__ import annotations

import torch

# fp4-e2m1 magnitudes in code order; sign bit is code bit 3 (reference utils.py:201-218)
_MX4_MAGNITUDES = (0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0)


def group_quantize_tensor(w_orig: torch.Tensor, n_bit: int, q_group_size: int = 128):
    """[n][k] -> (int32 codes in [0, 2^n_bit), scales_and_zeros [k/g][n][2] in w_orig's dtype) with
    dequant = (code - 2^(n_bit-1)) * scale + zero   (reference utils.py:27-67)."""
    assert w_orig.dim() == 2 and q_group_size > 1 and w_orig.shape[-1] % q_group_size == 0
    n, k = w_orig.shape
    groups = w_orig.float().reshape(-1, q_group_size)
    assert not torch.isnan(groups).any()
    lo = groups.amin(dim=1, keepdim=True)
    hi = groups.amax(dim=1, keepdim=True)
    levels = 2 ** n_bit - 1
    # (a device-tensor divisor: on the GPU torch evaluates `tensor / python_scalar` as a product with the rounded reciprocal,
    # one ulp off the IEEE quotient of the reference's CPU path)
    scale = (hi - lo).clamp(min=1e-6) / torch.tensor(float(levels), device=groups.device)
    zero = lo + scale * (2 ** (n_bit - 1))
    codes = groups.sub(lo).div(scale).round().clamp_(0, levels).to(torch.int32).reshape(n, k)
    sz = torch.stack([scale.view(n, -1), zero.view(n, -1)], dim=2)  # [n][k/g][2]
    return codes, sz.transpose(0, 1).contiguous().to(w_orig.dtype)


def expand_q_groups(x: torch.Tensor, orig_size, q_group_size: int) -> torch.Tensor:
    rows, cols = orig_size
    return x.reshape(rows, cols // q_group_size, 1).expand(rows, cols // q_group_size, q_group_size).reshape(rows, cols)


def extract_scales_and_zeros(scales_and_zeros: torch.Tensor, w_shape, q_group_size: int):
    per_row = scales_and_zeros.transpose(0, 1)
    return (expand_q_groups(per_row[:, :, 0], w_shape, q_group_size),
            expand_q_groups(per_row[:, :, 1], w_shape, q_group_size))


def round_to_mx4(x: torch.Tensor, q_group_size: int):
    """Round each group of `q_group_size` values along the last dim to fp4-e2m1 times a shared power of
    two.  Returns (values / 2^e as float32, e as float32 [rows][groups])   (reference utils.py:85-134 and
    the microxcaling emulation it calls: shared exponent = floor(log2(max|x| rounded to even at 1 mantissa
    bit... i.e. +2^22 on the f32 bits)), element rounding = nearest, half away from zero, saturating)."""
    x = x.float()
    rows, cols = x.shape
    assert cols % q_group_size == 0
    g = x.reshape(rows, cols // q_group_size, q_group_size)
    amax = g.abs().amax(dim=-1, keepdim=True)
    # "even" rounding of the max before taking the exponent (mx_ops.py:78-90)
    bits = amax.view(torch.int32)
    bits = (bits + (1 << 22)) & (((1 << 9) - 1) << 23)
    amax_r = bits.view(torch.float32)
    tiny = torch.finfo(torch.float32).tiny
    shared = torch.floor(torch.log2(amax_r + tiny * (amax_r == 0).float()))
    g = g * (shared > -127).float()       # flush groups whose exponent is subnormal
    emax = 2.0                            # fp4-e2m1: largest normal exponent
    e = shared - emax
    e = torch.where(e > 127, torch.full_like(e, float("nan")), e).clamp(min=-127)
    a = g / (2.0 ** e)
    # element quantisation to e2m1: 1 implicit + 1 explicit mantissa bit, min normal exponent 0
    pe = torch.floor(torch.log2(a.abs() + (a == 0).float())).clamp(min=0.0)
    q = a / (2.0 ** pe) * 2.0
    q = torch.sign(q) * torch.floor(q.abs() + 0.5)
    q = q / 2.0 * (2.0 ** pe)
    q = q.clamp(min=-6.0, max=6.0)
    return q.reshape(rows, cols), e.reshape(rows, cols // q_group_size)


def quantize_mx4(x: torch.Tensor, q_group_size: int):
    """-> (int32 codes [rows][cols] in fp4 sign-magnitude order, uint8 exponents [rows][cols/g] = e + 127)."""
    q, e = round_to_mx4(x, q_group_size)
    mags = torch.tensor(_MX4_MAGNITUDES, dtype=torch.float32, device=q.device)
    idx = (q.abs().unsqueeze(-1) == mags).float().argmax(dim=-1).to(torch.int32)
    assert bool(((mags[idx.long()] == q.abs())).all()), "value not representable in fp4-e2m1"
    negative = torch.signbit(q)
    codes = idx + 8 * negative.to(torch.int32)
    assert bool((e <= 128).all())
    return codes, (e + 127).to(torch.uint8)


def dequantize_mx4(q: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
    groups = e.size(1)
    assert q.size(1) % groups == 0
    mags = torch.tensor(_MX4_MAGNITUDES + tuple(-m for m in _MX4_MAGNITUDES), dtype=torch.float32, device=q.device)
    v = mags[q.long()].reshape(q.size(0), groups, -1)
    return (v * (2.0 ** (e.float() - 127)).unsqueeze(-1)).reshape(q.shape)
