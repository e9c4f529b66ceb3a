"""Quantizer on the box: produces the inputs of the tinygemm path -- codes, per-row
16-entry LUT, per-group scale/zero -- from 16/32-bit weights, and swaps `nn.Linear` modules for the quantized
modules, with the reference's function names, arguments and return conventions:

  group_q / degroup_q / pack_scales_and_zeros / extract_scales_and_zeros     quantize.py:87-181
  intq_quantize_tensor / intq_dequantize_tensor / intq_reconstruct_tensor    quantize.py:183-221
  anyq_quantize_tensor / anyq_dequantize_tensor / anyq_reconstruct_tensor    quantize.py:523-637, 810-825
  intq_layer / anyq_layer                                                    quantize.py:333-391, 827-905
  quantize_model                                                             quantize.py:32-85

What is different by design: the reference clusters every weight row with scikit-learn's KMeans on the host
(one Python task per row under joblib, quantize.py:406-418, 506-521 -- minutes per 4096x4096 layer).  Here all
rows are clustered at once by a batched weighted 1-D Lloyd iteration in torch on the device that holds the
weights (`kmeans_rows`): a 1-D assignment is a `searchsorted` against the 15 midpoints of the sorted
centroids, the update two `scatter_add`s, so a layer takes milliseconds on the GPU.  Parity with scikit-learn
is statistical, not bit-exact (different seeding): tests pin it by reconstruction error against the reference's
own output on the fixture of BASELINE config 1 and by exact recovery when a row has <= 16 distinct values.
This module is the step BEFORE the hot path; none of it runs per token.
"""
from __future__ import annotations

import gc
import warnings
from typing import Callable, Optional

import torch


# --------------------------------------------------------------------------------------------------
# grouping (quantize.py:87-181)
# --------------------------------------------------------------------------------------------------

def pack_scales_and_zeros(scales: torch.Tensor, zeros: torch.Tensor, w_shape) -> torch.Tensor:
    """[n * k/g] scales and zeros -> [k/g][n][2] (scale, zero of one q-group adjacent: one 32-bit load)."""
    n = w_shape[0]
    return torch.stack([scales.reshape(n, -1), zeros.reshape(n, -1)], dim=2).transpose(0, 1).contiguous()


def expand_q_groups(x: torch.Tensor, orig_size, q_group_size: int) -> torch.Tensor:
    n, k = orig_size
    return x.reshape(n, k // q_group_size, 1).expand(n, k // q_group_size, q_group_size).reshape(n, k)


def extract_scales_and_zeros(scales_and_zeros: torch.Tensor, w_shape, q_group_size: int):
    s = scales_and_zeros.transpose(0, 1)
    return expand_q_groups(s[:, :, 0], w_shape, q_group_size), expand_q_groups(s[:, :, 1], w_shape, q_group_size)


_DIVISORS: dict = {}


def _divisor(value: float, device) -> torch.Tensor:
    key = (value, str(device))
    t = _DIVISORS.get(key)
    if t is None:
        t = _DIVISORS[key] = torch.tensor(value, device=device)
    return t


def group_q(w_orig: torch.Tensor, n_bit: int, q_group_size: int = 128, assymetric: bool = True, unsigned: bool = True,
            zero_point: bool = True):
    """Scale every q-group onto the integer grid WITHOUT rounding: returns (w in grid units, the image of 0,
    scales_and_zeros).  Asymmetric: scale = max(max - min, 1e-6) / (2^n - 1), zero = min (+ 2^(n-1) scale)."""
    w = w_orig.float()
    if q_group_size <= 1 or w.dim() != 2 or w.shape[-1] % q_group_size != 0:
        raise ValueError("group_q needs a 2-D tensor whose last dimension is a multiple of q_group_size > 1")
    tq = w.reshape(-1, q_group_size)
    if torch.isnan(tq).any():
        raise ValueError("NaN in the tensor to quantize")
    if assymetric:
        mx, mn = tq.amax(dim=1, keepdim=True), tq.amin(dim=1, keepdim=True)
        lo, hi = (0, 2 ** n_bit - 1) if unsigned else (-(2 ** (n_bit - 1)), 2 ** (n_bit - 1) - 1)
        # (divisor as a tensor on w's device: torch turns `tensor / python_scalar` on the GPU into a multiplication by the rounded
        # reciprocal, one ulp off the IEEE quotient the reference's CPU path computes -- enough to flip a bf16 scale.  PARITY TARGET: the
        # reference run on the CPU, which is what tests/golden/group_quant.npz captured and tests/test_quantize_cpu.py /
        # test_gpu_decode.py check bit for bit on both devices; a checkpoint quantised by the reference ON A GPU can differ from this
        # by one ulp of a scale.  The divisor tensor is made once per (device, value): no host-to-device copy per call.)
        scales = (mx - mn).clamp(min=1e-6) / _divisor(float(hi - lo), w.device)
        zeros = mn + scales * (2 ** (n_bit - 1)) if zero_point else mn
        w_new = tq.sub(mn).div(scales).reshape(w.shape)
        w_zero = torch.zeros_like(tq).sub(mn).div(scales).reshape(w.shape)
    else:
        scales = tq.abs().amax(dim=1, keepdim=True).clamp(min=1e-6) / _divisor(float(2 ** (n_bit - 1) - 1), w.device)
        zeros = torch.zeros_like(scales)
        w_new = tq.div(scales).reshape(w.shape)
        w_zero = torch.zeros_like(w_new)
    return w_new, w_zero, pack_scales_and_zeros(scales, zeros, w.shape)


def degroup_q(w_c: torch.Tensor, scales_and_zeros=None, scales=None, zeros=None, n_bit: int = 4, q_group_size: int = 128,
              centering: bool = True) -> torch.Tensor:
    if scales_and_zeros is not None:
        s1, z1 = extract_scales_and_zeros(scales_and_zeros, w_c.shape, q_group_size)
        scales = s1 if scales is None else scales
        zeros = z1 if zeros is None else zeros
    if not q_group_size:
        return w_c
    if centering:
        w_c = w_c - (2 ** (n_bit - 1))
    return w_c * scales + zeros


# --------------------------------------------------------------------------------------------------
# uniform integer quantization (quantize.py:183-215)
# --------------------------------------------------------------------------------------------------

def intq_quantize_tensor(x: torch.Tensor, n_bit: int = 4, q_group_size: int = 128, scale_only: bool = False,
                         new_grouping=False, unsigned: bool = False, zero_point: bool = False, **_):
    """-> (int32 codes, the image of zero, scales_and_zeros [k/g][n][2] in x's dtype).
    new_grouping="tinygemm": the packing-ready grid of tinygemm_lib.utils.group_quantize_tensor (codes 0..15,
    zero = min + 8 scale) -- what the Int4Linear kernels consume; otherwise group_q + round."""
    if new_grouping == "tinygemm":
        from .utils import group_quantize_tensor

        intq, sz = group_quantize_tensor(x, n_bit=n_bit, q_group_size=q_group_size)
        _, zg = extract_scales_and_zeros(sz, x.shape, q_group_size)
    elif new_grouping:
        raise NotImplementedError("group_q1 (new_grouping=True) is not part of the tinygemm path")
    else:
        intq, zg, sz = group_q(x, n_bit, q_group_size=q_group_size, assymetric=not scale_only, unsigned=unsigned,
                               zero_point=zero_point)
        intq = intq.round()
    return intq.to(torch.int32), zg, sz.to(x.dtype)


def intq_dequantize_tensor(intq: torch.Tensor, scales_and_zeros=None, scales=None, zeros=None, n_bit: int = 4,
                           q_group_size: int = 128, dtype=torch.float16, **_):
    return degroup_q(intq, scales_and_zeros=scales_and_zeros, scales=scales, zeros=zeros, n_bit=n_bit,
                     q_group_size=q_group_size, centering=False).to(dtype)


def intq_reconstruct_tensor(x, n_bit: int = 4, q_group_size: int = 128, unsigned: bool = False, zero_point: bool = False,
                            scale_only: bool = False, dtype=torch.float16, **_):
    intq, _, sz = intq_quantize_tensor(x, n_bit=n_bit, q_group_size=q_group_size, scale_only=scale_only, unsigned=unsigned,
                                       zero_point=zero_point)
    return intq_dequantize_tensor(intq, scales_and_zeros=sz, n_bit=n_bit, q_group_size=q_group_size, dtype=dtype)


# --------------------------------------------------------------------------------------------------
# batched weighted 1-D k-means (replaces quantize.py:406-418 + kmeans.py:139-287)
# --------------------------------------------------------------------------------------------------

def _init_centers(kind: str, x, xs, lo, span, C: int):
    rows, k = x.shape
    steps = (torch.arange(C, device=x.device, dtype=torch.float32) + 0.5) / C
    if kind == "uniform":      # equally spaced over the row's range
        return lo + span * steps
    if kind == "quantile":     # equal-population cells
        return xs[:, (steps * k).long().clamp(0, k - 1)]
    if kind == "density":      # Panter-Dite: the optimal 1-D quantizer's point density goes with pdf^(1/3)
        B = 512
        b = ((x - lo) / span * B).long().clamp(0, B - 1)
        dens = torch.zeros(rows, B, device=x.device).scatter_add_(1, b, torch.ones_like(x)).pow(1.0 / 3.0)
        cdf = dens.cumsum(1)
        cdf = (cdf / cdf[:, -1:]).contiguous()
        bi = torch.searchsorted(cdf, steps.expand(rows, C).contiguous()).clamp(0, B - 1)
        return lo + (bi.float() + 0.5) / B * span
    raise ValueError(f"unknown init {kind!r}")


def _lloyd(x, wts, xw, c, span, max_iter: int, tol: float):
    """Weighted Lloyd iterations from centres c; returns (assign, centres sorted, weighted SSE per row)."""
    rows, k = x.shape
    C = c.shape[1]
    for _ in range(max_iter):
        c = torch.sort(c, dim=1).values
        mid = ((c[:, 1:] + c[:, :-1]) * 0.5).contiguous()
        assign = torch.searchsorted(mid, x)                                   # [rows][k] in 0..C-1
        cnt = torch.zeros(rows, C, device=x.device).scatter_add_(1, assign, wts)
        tot = torch.zeros(rows, C, device=x.device).scatter_add_(1, assign, xw)
        new = torch.where(cnt > 0, tot / cnt.clamp_min(1e-30), c)
        empty = cnt <= 0
        if bool(empty.any()):
            # re-seed the j-th empty cluster of a row at its j-th worst-represented point
            err = (x - new.gather(1, assign)).abs() * wts
            top_e, top_i = torch.topk(err, min(C, k), dim=1)
            rank = (torch.cumsum(empty.int(), dim=1) - 1).clamp(0, top_i.shape[1] - 1)
            cand, cand_e = x.gather(1, top_i).gather(1, rank), top_e.gather(1, rank)
            new = torch.where(empty & (cand_e > 0), cand, new)
        moved = bool(((new - c).abs() > tol * span).any())
        c = new
        if not moved:
            break
    c = torch.sort(c, dim=1).values
    mid = ((c[:, 1:] + c[:, :-1]) * 0.5).contiguous()
    assign = torch.searchsorted(mid, x)
    sse = ((x - c.gather(1, assign)) ** 2 * wts).sum(1)
    return assign, c, sse


@torch.no_grad()
def kmeans_rows(x: torch.Tensor, n_clusters: int = 16, sample_weight: Optional[torch.Tensor] = None,
                max_iter: int = 300, tol: float = 1e-6, init=("uniform", "density", "quantile")):
    """Independent 1-D k-means on every row of x [rows][k] (float32, any device).

    sample_weight: None, [k] (shared by all rows -- activation-aware any4, quantize.py:483-489) or [rows][k].
    init: one seeding or a tuple of seedings ("uniform" | "density" | "quantile"); with several, every row keeps
    the clustering with the smallest weighted squared error (scikit-learn's `n_init`, deterministic here).
    Empty clusters are re-seeded at the points with the largest weighted error, so a row with at most C distinct
    values is reproduced exactly.  Stops when no centroid moves by more than tol * row range (max_iter as sklearn).
    Returns (assign int32 [rows][k], centers float32 [rows][C] sorted ascending).
    """
    x = x.float().contiguous()
    rows, k = x.shape
    if sample_weight is None:
        wts = torch.ones_like(x)
    else:
        wts = sample_weight.to(x.device, torch.float32).abs()
        wts = (wts.reshape(1, k).expand(rows, k) if wts.dim() == 1 else wts.reshape(rows, k)).contiguous()
        wts = wts.clamp_min(1e-12)  # a zero weight must not orphan a point (kmeans.py run_kmeans: sample_weight_eps)
    xw = x * wts
    xs = torch.sort(x, dim=1).values
    lo = xs[:, :1]
    span = (xs[:, -1:] - lo).clamp_min(1e-12)
    best = None
    # the reference forwards `init` to scikit-learn (None / "k-means++" / "random", quantize.py:413): those have no meaning
    # for this batched Lloyd -> its default deterministic seedings
    if init is None or (isinstance(init, str) and init in ("k-means++", "random")):
        if init is not None and init not in _WARNED_INIT:
            _WARNED_INIT.add(init)
            warnings.warn(f"kmeans_rows: init={init!r} is scikit-learn's seeding; this batched Lloyd has no random state and "
                          f"uses its deterministic seedings {_SEEDINGS} instead (results do not depend on a seed)", stacklevel=2)
        init = _SEEDINGS
    for kind in ((init,) if isinstance(init, str) else tuple(init)):
        a, c, sse = _lloyd(x, wts, xw, _init_centers(kind, x, xs, lo, span, n_clusters), span, max_iter, tol)
        if best is None:
            best = [a, c, sse]
        else:
            better = sse < best[2]
            best[0] = torch.where(better[:, None], a, best[0])
            best[1] = torch.where(better[:, None], c, best[1])
            best[2] = torch.where(better, sse, best[2])
    return best[0].to(torch.int32), best[1]


# --------------------------------------------------------------------------------------------------
# any4 (quantize.py:523-637, 810-825)
# --------------------------------------------------------------------------------------------------

_SEEDINGS = ("uniform", "density", "quantile")
_WARNED_INIT: set = set()  # scikit-learn seeding names already warned about (kmeans_rows)


@torch.no_grad()
def anyq_quantize_tensor(W: torch.Tensor, n_bit: int = 4, q_group_size: int = 128, per_row: bool = True,
                         zero_point: bool = True, scale_only: bool = False, sample_weight=None,
                         scale_sample_weight: bool = False, abs_weight_sample_weight: bool = False,
                         init=("uniform", "density", "quantile"), max_iter: int = 300, device=None, **_):
    """-> (assign int32 [n][k], any4 LUT [n][2^n_bit] (or [2^n_bit] when per_row=False) in the group-scaled
    [0, 2^n_bit - 1] domain, scales_and_zeros [k/g][n][2]); LUT and scales in W's dtype, on W's device.
    `device` = where to do the work (default: W's device; pass "cuda" to cluster a CPU checkpoint on the GPU)."""
    orig_device, dtype = W.device, W.dtype
    if device is not None:
        W = W.to(device)
    orig_shape = W.shape
    if not per_row:
        if q_group_size:
            Wg, _, sz = group_q(W, n_bit, q_group_size=q_group_size, assymetric=not scale_only, zero_point=zero_point)
            scales, _ = extract_scales_and_zeros(sz, Wg.shape, q_group_size)
        else:
            Wg, sz, scales = W.float(), None, None
        Wg = Wg.reshape(1, -1)
    elif q_group_size:
        Wg, _, sz = group_q(W, n_bit, q_group_size=q_group_size, assymetric=not scale_only, zero_point=zero_point)
        scales, _ = extract_scales_and_zeros(sz, Wg.shape, q_group_size)
    else:
        Wg = W.float()
        sz = pack_scales_and_zeros(torch.ones(W.shape[0], device=W.device), torch.zeros(W.shape[0], device=W.device), W.shape)
        scales = None
    sw = sample_weight
    if isinstance(sw, torch.Tensor):
        sw = sw.to(W.device, torch.float32)
    if scale_sample_weight and scales is not None:
        sw = (torch.ones_like(W[0], dtype=torch.float32) if sw is None else sw) * scales
    if abs_weight_sample_weight:
        sw = (torch.ones_like(W[0], dtype=torch.float32) if sw is None else sw) * W.abs().float()
    if sw is not None and not per_row:
        sw = sw.expand(orig_shape).reshape(1, -1) if sw.dim() == 1 else sw.reshape(1, -1)
    assign, any4 = kmeans_rows(Wg, n_clusters=2 ** n_bit, sample_weight=sw, max_iter=max_iter, init=init)
    if not per_row:
        assign = assign.reshape(orig_shape)
        any4 = any4.squeeze(0)
    return assign.to(orig_device), any4.to(orig_device, dtype), sz.to(orig_device, dtype)


def anyq_dequantize_tensor(assign, any4, scales_and_zeros, n_bit: int = 4, q_group_size: int = 128, per_row: bool = True,
                           scale_only: bool = False, **_):
    """(lut[code] - 2^(n_bit-1)) * scale + zero, evaluated op by op in the LUT's dtype (quantize.py:612-637)."""
    Wc = torch.gather(any4, 1, assign.long()) if per_row else any4[assign.long()]
    if not q_group_size:
        return Wc
    scales, zeros = extract_scales_and_zeros(scales_and_zeros, assign.shape, q_group_size)
    return degroup_q(Wc, scales=scales, zeros=zeros, n_bit=n_bit, q_group_size=q_group_size, centering=not scale_only)


def anyq_reconstruct_tensor(W, n_bit: int = 4, q_group_size: int = 128, per_row: bool = True, scale_only: bool = False, **kw):
    assign, any4, sz = anyq_quantize_tensor(W, n_bit=n_bit, q_group_size=q_group_size, per_row=per_row,
                                            scale_only=scale_only, **kw)
    return anyq_dequantize_tensor(assign, any4, sz, n_bit=n_bit, q_group_size=q_group_size, per_row=per_row,
                                  scale_only=scale_only)


# --------------------------------------------------------------------------------------------------
# module level (quantize.py:32-85, 333-391, 827-905)
# --------------------------------------------------------------------------------------------------

def anyq_layer(module: torch.nn.Module, name: str = "", n_bit: int = 4, group_size: int = 128, per_row: bool = True,
               scale_only: bool = False, sample_weight=None, pseudo: Optional[bool] = None,
               kernel: str = "linear_y_f16RM_x_f16RM_W_any4TC", w_inner_k: int = 4, **kwargs) -> torch.nn.Module:
    """nn.Linear -> Any4Linear on the HIP kernels (pseudo=False, default) or the same nn.Linear with
    reconstructed weights (pseudo=True: fake quantization for accuracy studies)."""
    if pseudo is None:
        pseudo = n_bit != 4
    w = module.weight
    if isinstance(sample_weight, dict):
        sample_weight = sample_weight[name]
    if pseudo:
        w_deq = anyq_reconstruct_tensor(w, n_bit=n_bit, q_group_size=group_size, per_row=per_row, scale_only=scale_only,
                                        sample_weight=sample_weight, **kwargs)
        module.weight.data = w_deq.to(device=w.device, dtype=w.dtype)
        return module
    if n_bit != 4:
        raise ValueError(f"No quantized module for n_bit={n_bit}; use pseudo=True")
    from .modules import Any4Linear  # needs the HIP library: loud ImportError otherwise

    codes, lut, sz = anyq_quantize_tensor(w, n_bit=n_bit, q_group_size=group_size, per_row=per_row, scale_only=scale_only,
                                          sample_weight=sample_weight, **kwargs)
    q = Any4Linear(module.in_features, module.out_features, bias=module.bias is not None, device=w.device, dtype=w.dtype,
                   group_size=group_size if group_size else w.shape[1], per_row=per_row, kernel=kernel, w_inner_k=w_inner_k)
    q.weight.data = codes.to(w.device)
    q.scales_and_zeros.data = sz.to(w.device)
    q.lut.data = lut.to(w.device) - (2 ** (n_bit - 1))  # the kernel's LUT is centred (quantize.py:893)
    q.bias = module.bias
    q.reshape_weight(w_inner_k)
    return q


def intq_layer(module: torch.nn.Module, name: str = "", n_bit: int = 4, group_size: int = 128, pseudo: Optional[bool] = None,
               **kwargs) -> torch.nn.Module:
    """nn.Linear -> Int4Linear (uniform int4 on tinygemm's grid) or, pseudo=True, reconstructed weights in place."""
    if pseudo is None:
        pseudo = n_bit not in (4, 8)
    w = module.weight
    if pseudo:
        module.weight.data = intq_reconstruct_tensor(w, n_bit=n_bit, q_group_size=group_size, **kwargs).to(device=w.device, dtype=w.dtype)
        return module
    if n_bit == 8:  # quantize.py:337-360 builds Int8Linear from group_quantize_tensor(n_bit=8)
        from .modules import Int8Linear
        from .utils import group_quantize_tensor

        codes, sz = group_quantize_tensor(w, n_bit=8, q_group_size=group_size)
        q = Int8Linear(module.in_features, module.out_features, bias=module.bias is not None, device=w.device, dtype=w.dtype,
                       group_size=group_size)
        q.weight.data = codes.to(w.device)
        q.scales_and_zeros.data = sz.to(w.device)
        q.bias = module.bias
        q.reshape_weight()
        return q
    if n_bit != 4:
        raise ValueError(f"No int quantized module built for n_bit={n_bit}; use pseudo=True")
    from .modules import Int4Linear

    codes, _, sz = intq_quantize_tensor(w, n_bit=n_bit, q_group_size=group_size, new_grouping="tinygemm")
    q = Int4Linear(module.in_features, module.out_features, bias=module.bias is not None, device=w.device, dtype=w.dtype,
                   group_size=group_size)
    q.weight.data = codes.to(w.device)
    q.scales_and_zeros.data = sz.to(w.device)
    q.bias = module.bias
    q.reshape_weight()
    return q


@torch.no_grad()
def nf4_quantize_tensor(W: torch.Tensor, q_group_size: int = 128):
    """NormalFloat4 with absmax group scaling (what bitsandbytes' quantize_nf4 does, quantize.py:923-937 uses it for the
    pseudo path): -> (int32 codes [n][k], lut [16] = NF4 code book, scales_and_zeros [k/g][n][(absmax, 0)])."""
    from .modules import NF4_VALUES

    n, k = W.shape
    grp = W.float().reshape(-1, q_group_size)
    absmax = grp.abs().amax(dim=1, keepdim=True).clamp_min(1e-12)
    book = torch.tensor(NF4_VALUES, device=W.device, dtype=torch.float32)
    mid = ((book[1:] + book[:-1]) * 0.5).contiguous()
    codes = torch.searchsorted(mid, (grp / absmax).contiguous()).to(torch.int32).reshape(n, k)
    sz = pack_scales_and_zeros(absmax, torch.zeros_like(absmax), W.shape)
    return codes, book.to(W.dtype), sz.to(W.dtype)


def nf4_layer(module: torch.nn.Module, name: str = "", n_bit: int = 4, group_size: int = 128, pseudo: Optional[bool] = None,
              kernel: str = "linear_y_f16RM_x_f16RM_W_any4TC", w_inner_k: int = 4, **_) -> torch.nn.Module:
    """nn.Linear -> NF4Linear (real kernels) or fake-quantized weights in place (pseudo=True)."""
    assert n_bit == 4, "nf4 only supports 4-bit"
    w = module.weight
    codes, lut, sz = nf4_quantize_tensor(w, q_group_size=group_size)
    if pseudo:
        s, _ = extract_scales_and_zeros(sz.float(), w.shape, group_size)
        module.weight.data = (lut.float()[codes.long()] * s).to(w.dtype)
        return module
    from .modules import NF4Linear

    q = NF4Linear(module.in_features, module.out_features, bias=module.bias is not None, device=w.device, dtype=w.dtype,
                  group_size=group_size, kernel=kernel, w_inner_k=w_inner_k)
    q.weight.data, q.scales_and_zeros.data, q.bias = codes.to(w.device), sz.to(w.device), module.bias
    q.reshape_weight(w_inner_k)
    return q


def mx4_layer(module: torch.nn.Module, name: str = "", group_size: int = 32, pseudo: Optional[bool] = None,
              kernel: str = "linear_y_f16RM_x_f16RM_W_mx4TC", w_inner_k: int = 4, **_) -> torch.nn.Module:
    """nn.Linear (bf16) -> MX4Linear, or fake-quantized weights in place (pseudo=True)."""
    from .utils import dequantize_mx4, quantize_mx4

    w = module.weight
    codes, exps = quantize_mx4(w.float(), group_size)
    if pseudo:
        module.weight.data = dequantize_mx4(codes, exps).to(w.dtype)
        return module
    from .modules import MX4Linear

    q = MX4Linear(module.in_features, module.out_features, bias=module.bias is not None, device=w.device, dtype=w.dtype,
                  group_size=group_size, kernel=kernel, w_inner_k=w_inner_k)
    q.weight.data, q.exponents.data, q.bias = codes.to(w.device), exps.to(w.device), module.bias
    q.reshape_weight(w_inner_k)
    return q


def quantize_model(model: torch.nn.Module, layer_from=torch.nn.Linear, layer_to: Callable = anyq_layer, skip_modules=None,
                   **kwargs) -> torch.nn.Module:
    """Replace every `layer_from` submodule by `layer_to(module, name=..., **kwargs)`, in place.
    skip_modules: names or modules to leave alone; default = the LM head when the model has one
    (`lm_head` attribute or `get_output_embeddings()`), as quantization papers do (quantize.py:34-36)."""
    if skip_modules is None:
        head = getattr(model, "lm_head", None)
        if head is None and hasattr(model, "get_output_embeddings"):
            head = model.get_output_embeddings()
        skip_modules = [] if head is None else [head]
    if isinstance(skip_modules, str):
        skip_modules = [s.strip() for s in skip_modules.split(",") if s.strip()]
    if isinstance(skip_modules, torch.nn.Module):
        skip_modules = [skip_modules]
    todo = [(n, m) for n, m in model.named_modules()
            if isinstance(m, layer_from) and n not in skip_modules and not any(m is s for s in skip_modules)]
    for name, module in todo:
        new = layer_to(module, name=name, **kwargs)
        parent = model.get_submodule(".".join(name.split(".")[:-1])) if "." in name else model
        setattr(parent, name.split(".")[-1], new)
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        gc.collect()
    return model
