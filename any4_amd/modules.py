"""Quantized Linear modules with the reference's class names, constructor kwargs, parameter /
attribute names and `reshape_weight()` protocol (modules.py:12-230 in facebookresearch/any4), so that
`quantize.{intq,anyq}_layer`, `eval.py` and `benchmark.py` can use them unchanged.

Parameters:
  weight            int32 [out][in] codes 0..15; after reshape_weight(): the packed 4-D int32 tensor
  scales_and_zeros  [in / group_size][out][2]
  lut               Any4Linear only: [out][16] (per_row) or [16]
  bias              optional [out]
The string attribute `kernel` names the functional (any4_amd/functional.py) used by forward().
"""
from __future__ import annotations

import torch

from . import functional as F
from . import ops as _ops

_T = torch.ops.tinygemm


class _PackedLinear(torch.nn.Module):
    """Shared machinery: parameter creation, one-off packing and the forward epilogue."""

    # kernel name -> which packer produces the layout that kernel consumes
    _PACKERS: dict = {}
    _DEFAULT_INNER_K = 4

    def _make_common(self, in_features, out_features, bias, device, dtype, group_size, kernel, w_inner_k, zero_init):
        self.in_features = in_features
        self.out_features = out_features
        self.group_size = group_size
        alloc = torch.zeros if zero_init else torch.empty
        self.weight = torch.nn.Parameter(alloc((out_features, in_features), device=device, dtype=torch.int32), requires_grad=False)
        self.scales_and_zeros = torch.nn.Parameter(alloc((in_features // group_size, out_features, 2), device=device, dtype=dtype))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(out_features, device=device, dtype=dtype))
        else:
            self.register_parameter("bias", None)
        self.kernel = kernel
        self.w_inner_k = w_inner_k
        self.weight_reshaped = False

    def reshape_weight(self, w_inner_k: int | None = None):
        """Pack `weight` once into the layout `self.kernel` consumes."""
        if w_inner_k is None:
            w_inner_k = self._DEFAULT_INNER_K
        packer = self._PACKERS.get(self.kernel)
        if packer is None:
            raise ValueError(f"Unsupported kernel type {self.kernel}")
        self.weight.data = getattr(_T, packer)(self.weight, w_inner_k)
        self.weight_reshaped = True
        self.w_inner_k = w_inner_k

    @property
    def weight_format(self):
        """'native' / 'reference' for a packed weights-on-the-left tensor (read off the tensor's own shape, any4_amd.ops.aside_format:
        the two packed formats differ in shape, so the tag travels with the tensor through state_dict, pickling and other
        processes), None otherwise."""
        if not self.weight_reshaped or "Aint4" not in self._PACKERS.get(self.kernel, "") or self.weight.dim() != 4:
            return None
        return _ops.aside_format(self.weight, self.in_features)

    def relayout(self, to: str):
        """Repack an already packed weights-on-the-left tensor ('reference' <-> 'native', lossless; any4_amd.ops.relayout_Aint4),
        e.g. once after loading a checkpoint packed by the CUDA implementation."""
        if self.weight_format is None:
            raise ValueError("relayout() applies to a packed weights-on-the-left (Aint4) tensor")
        self.weight.data = _ops.relayout_Aint4(self.weight.data, self.in_features, to, self.w_inner_k)

    # ---- checkpoints (eval.py:180-210 saves / loads state_dicts).  The reference keeps kernel / w_inner_k / weight_reshaped in plain
    # attributes (modules.py:38-41); here a state_dict holds TENSORS ONLY -- exactly the reference's keys, packed or not, so tensor-only
    # consumers (safetensors, `{k: v.cpu()}`) and the reference's strict load_state_dict take it -- and everything that decides how
    # `weight` has to be read is recovered from the tensor's own shape (_read_packed_shape).
    _TAG = torch.nn.modules.module._EXTRA_STATE_KEY_SUFFIX

    def _read_packed_shape(self, w: torch.Tensor):
        """(w_inner_k or None) of a packed 4-D checkpoint tensor for THIS module's kernel; raises when the tensor was packed for
        the other operand side / another problem size (it would be mis-multiplied or read out of bounds, never 'just slower')."""
        packer = self._PACKERS.get(self.kernel, "")
        n, k = self.out_features, self.in_features
        s0, s1, s2, s3 = w.shape
        ok, inner = False, None
        if s2 == 32:
            if "Bint4" in packer:     # [ceil(n/8)][k/(16 I)][32][I/2], TinyGemm_int4.cu:322-364
                ok, inner = s0 == -(-n // 8) and s3 in (1, 2, 4) and s1 * s3 * 32 == k, s3 * 2
            elif "Aint4" in packer:   # the reference's [ceil(n/16)][k/(16 I)][32][I] or the native Bint4 tensor of the 16-row-padded rows
                if s0 == -(-n // 16) and s3 in (1, 2, 4) and s1 * s3 * 16 == k:
                    ok, inner = True, s3
                elif s0 == 2 * -(-n // 16) and s1 * s3 * 32 == k and s3 * 2 == _ops._rows_inner(k):
                    ok, inner = True, None   # (the Aint4 innerKTiles of a native tensor is a request for relayout('reference') only)
            elif "Bint8" in packer:   # [ceil(n/8)][k/(16 I)][32][I]
                ok, inner = s0 == -(-n // 8) and s3 in (1, 2, 4) and s1 * s3 * 16 == k, s3
            elif "Aint8" in packer:   # [ceil(n/16)][ceil(ceil(k/16)/I)][32][2 I]
                ok, inner = s0 == -(-n // 16) and s3 in (2, 4) and s1 == -(-(-(-k // 16)) // (s3 // 2)), s3 // 2
        if not ok:
            raise RuntimeError(f"checkpoint weight {tuple(w.shape)} was not packed for kernel {self.kernel!r} of a "
                               f"[{n}][{k}] layer (packed for kernel of the other operand side, or another layer size)")
        return inner

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        w = state_dict.get(prefix + "weight")
        tag = state_dict.pop(prefix + self._TAG, None)   # (round-5 checkpoints carried a dict here; the shape says the same)
        inner = None
        if w is not None:
            if w.dim() == 4:
                inner = self._read_packed_shape(w)
            if w.shape != self.weight.shape:
                # a packed checkpoint into an unpacked module, the reverse, or another packed format / innerKTiles of the same
                # matrix (the reference's Aint4 words into a natively packed module, I = 2 into I = 4): take the checkpoint's shape
                self.weight.data = torch.empty(w.shape, dtype=self.weight.dtype, device=self.weight.device)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        if w is not None:
            self.weight_reshaped = w.dim() == 4
            if inner is not None:
                self.w_inner_k = inner
            elif isinstance(tag, dict) and self.weight_reshaped and "w_inner_k" in tag:
                self.w_inner_k = int(tag["w_inner_k"])
            # A checkpoint packed by the CUDA implementation for the weights-on-the-left kernels holds the reference's Aint4 words,
            # which only the reference-numerics kernels read (config 3: 0.59 vs 0.79 of the HBM roofline): repacked ONCE, losslessly,
            # to the row-per-lane order -- here when the parameter already lives on the GPU, else at the first forward on it.
            # Opt out: any4_amd.set_auto_relayout(False) / ANY4_AUTO_RELAYOUT=0, or a process default of weight_format 'reference'.
            self.__dict__["_relayout_pending"] = bool(
                self.weight_reshaped and _ops.get_auto_relayout() and _ops.get_weight_format() == "native"
                and self.weight_format == "reference" and self.in_features % 32 == 0)
            if self.__dict__["_relayout_pending"] and self.weight.is_cuda:
                self._auto_relayout()
        self.__dict__.pop("_plan", None)
        self.__dict__.pop("_no_plan", None)

    def _auto_relayout(self):
        self.__dict__["_relayout_pending"] = False
        if self.weight_format == "reference":
            self.relayout("native")
            self.__dict__.pop("_plan", None)

    def _gemm(self, x2d: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return self._forward(input)

    def __getstate__(self):
        # (the recorded launch plan holds raw device pointers in a ctypes struct: never copied or pickled with the module)
        state = self.__dict__.copy()
        state.pop("_plan", None)
        state.pop("_no_plan", None)
        return state

    _PLAN_PARAMS = ("weight", "scales_and_zeros", "exponents", "lut")

    def _forward(self, input: torch.Tensor) -> torch.Tensor:
        d = self.__dict__
        plan = d.get("_plan")
        p = self._parameters
        if plan is not None and p.get("bias") is None:   # (a bias assigned after the recording: the full path, which offers it to the kernel)
            # the validated launch of this (module, activation shape) re-issued with new pointers (ops.LaunchPlan.try_run): the eager hot path
            y = plan.try_run(input, p["weight"], p.get("scales_and_zeros") if "scales_and_zeros" in p else p.get("exponents"), p.get("lut"),
                             (self.kernel, self.group_size, self.w_inner_k))
            if y is not None:
                return y
        if d.get("_relayout_pending") and self.weight.is_cuda:
            self._auto_relayout()    # (a checkpoint in the reference's Aint4 words that was loaded on the CPU: repacked once, see above)
        lead = input.shape[:-1]
        if input.is_cuda and self.bias is None and self.weight_reshaped and d.get("_no_plan") != (input.shape, _ops.get_numerics()):
            # a packed weight only: the plan points at the parameters themselves
            x2d = input.view(-1, input.shape[-1])
            if x2d.is_contiguous() and x2d.data_ptr() % 16 == 0:
                y, lp = _ops.record_plan(self._gemm, x2d, None, [self._parameters.get(n) for n in self._PLAN_PARAMS])
                if lp is not None:
                    lp.attrs = (self.kernel, self.group_size, self.w_inner_k)
                    d["_plan"] = lp
                else:
                    d.pop("_plan", None)
                    d["_no_plan"] = (input.shape, _ops.get_numerics())   # (this flavour has no single-launch plan: remembered, not retried per call)
                return y.view(*lead, y.shape[-1])
        if self.bias is None:
            y = self._gemm(input.view(-1, input.shape[-1]))
        else:
            # the row-major GEMM kernels add the bias in their output store (same bits as the reference's separate
            # `y + bias`, modules.py:221-222, one launch fewer); layouts that cannot take it get the separate add
            with _ops.fused_bias(self.bias) as fb:
                y = self._gemm(input.view(-1, input.shape[-1]))
            if not fb.consumed:
                y = y + self.bias
        return y.view(*lead, y.shape[-1])

    def extra_repr(self) -> str:
        return (f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, "
                f"group_size={self.group_size}")


class Int4Linear(_PackedLinear):
    _PACKERS = {
        "linear_y_f16RM_x_f16RM_W_int4TC": "convert_matrix_to_m16n8k16_Bint4_layout",
        "linear_y_f16RM_W_int4TC_x_f16RM": "convert_matrix_to_m16n8k16_Aint4_layout",
        "linear_y_f16TC_x_f16TC_W_int4TC": "convert_matrix_to_m16n8k16_Bint4_layout",
    }
    _KERNELS = ("linear_y_f16RM_x_f16RM_W_int4TC", "linear_y_f16RM_W_int4TC_x_f16RM",
                "linear_y_f16TC_W_int4TC_x_f16TC", "linear_y_f16TC_x_f16TC_W_int4TC")

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None,
                 group_size: int = 128, kernel: str = "linear_y_f16RM_W_int4TC_x_f16RM", w_inner_k: int = 4) -> None:
        super().__init__()
        self._make_common(in_features, out_features, bias, device, dtype, group_size, kernel, w_inner_k, zero_init=True)

    def _gemm(self, x):
        if self.kernel not in self._KERNELS:
            raise ValueError(f"Unsupported kernel type {self.kernel}")
        return getattr(F, self.kernel)(x, self.weight, self.scales_and_zeros, self.group_size,
                                       w_inner_k=self.w_inner_k, reshape_weight=not self.weight_reshaped)


class Int8Linear(_PackedLinear):
    """int8 weights (modules.py:85-152): codes 0..255 from group_quantize_tensor(n_bit=8), value = code - 128."""
    _PACKERS = {
        "linear_y_f16RM_x_f16RM_W_int8TC": "convert_matrix_to_m16n8k16_Bint8_layout",
        "linear_y_f16RM_W_int8TC_x_f16RM": "convert_matrix_to_m16n8k16_Aint8_layout",
    }
    _KERNELS = ("linear_y_f16RM_x_f16RM_W_int8TC", "linear_y_f16RM_W_int8TC_x_f16RM", "linear_y_f16TC_W_int8TC_x_f16TC")
    _DEFAULT_INNER_K = 2

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None,
                 group_size: int = 128, kernel: str = "linear_y_f16RM_W_int8TC_x_f16RM", w_inner_k: int = 2) -> None:
        super().__init__()
        self._make_common(in_features, out_features, bias, device, dtype, group_size, kernel, w_inner_k, zero_init=True)

    def _gemm(self, x):
        if self.kernel not in self._KERNELS:
            raise ValueError(f"Unsupported kernel type {self.kernel}")
        return getattr(F, self.kernel)(x, self.weight, self.scales_and_zeros, self.group_size,
                                       w_inner_k=self.w_inner_k, reshape_weight=not self.weight_reshaped)


class Any4Linear(_PackedLinear):
    _N_BIT = 4
    _PACKERS = {
        "linear_y_f16RM_x_f16RM_W_any4TC": "convert_matrix_to_m16n8k16_Bint4_layout",
        "linear_y_f16RM_W_any4TC_x_f16RM": "convert_matrix_to_m16n8k16_Aint4_layout",
    }

    @property
    def N_BIT(self):
        return self._N_BIT

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None,
                 group_size: int = 128, kernel: str = "linear_y_f16RM_x_f16RM_W_any4TC", w_inner_k: int = 4,
                 per_row: bool = True) -> None:
        super().__init__()
        self.n_bit = 4
        self._make_common(in_features, out_features, bias, device, dtype, group_size, kernel, w_inner_k, zero_init=False)
        self.per_row = per_row
        lut_shape = (out_features, 2 ** self.N_BIT) if per_row else (2 ** self.N_BIT,)
        self.lut = torch.nn.Parameter(torch.empty(*lut_shape, device=device, dtype=dtype))

    def _gemm(self, x):
        if self.kernel not in self._PACKERS:
            raise ValueError(f"Unsupported kernel type {self.kernel}")
        return getattr(F, self.kernel)(x, self.weight, self.lut, self.scales_and_zeros, self.group_size,
                                       w_inner_k=self.w_inner_k, reshape_weight=not self.weight_reshaped)

    def extra_repr(self) -> str:
        return super().extra_repr() + f", per_row={self.per_row}"


# NF4 code book of QLoRA / bitsandbytes (the reference's kmeans.py:17 carries the same 16 values), ascending
NF4_VALUES = (-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
              -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
              0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0)


class NF4Linear(Any4Linear):
    """One of the modules the reference lists as TODO (modules.py:10): NormalFloat4 weights on the any4 kernel with ONE
    16-entry LUT for the whole matrix (the reference's "NF4" benchmark rows are exactly this, README.md:448-455).
    weight = codes 0..15, scales_and_zeros = [k/g][n][(absmax, 0)], lut = NF4_VALUES: w = lut[code] * absmax."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None,
                 group_size: int = 128, kernel: str = "linear_y_f16RM_x_f16RM_W_any4TC", w_inner_k: int = 4) -> None:
        super().__init__(in_features, out_features, bias=bias, device=device, dtype=dtype, group_size=group_size,
                         kernel=kernel, w_inner_k=w_inner_k, per_row=False)
        self.lut.data = torch.tensor(NF4_VALUES, device=device, dtype=dtype)


class MX4Linear(_PackedLinear):
    """The other TODO of modules.py:10: MX4 (fp4-e2m1 codes, one e8m0 exponent per group of 32) on
    tinygemm_y_f16RM_x_f16RM_w_mx4TC (bf16 only, TinyGemm_int4.cu:758).  Parameters: weight (codes, packed by
    reshape_weight), exponents uint8 [out][in / group_size]."""
    _PACKERS = {
        "linear_y_f16RM_x_f16RM_W_mx4TC": "convert_matrix_to_m16n8k16_Bint4_layout",
        "linear_y_f16RM_W_mx4TC_x_f16RM": "convert_matrix_to_m16n8k16_Aint4_layout",
    }

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None,
                 group_size: int = 32, kernel: str = "linear_y_f16RM_x_f16RM_W_mx4TC", w_inner_k: int = 4) -> None:
        super().__init__()
        self._make_common(in_features, out_features, bias, device, dtype, group_size, kernel, w_inner_k, zero_init=True)
        del self.scales_and_zeros
        self.exponents = torch.nn.Parameter(torch.full((out_features, in_features // group_size), 127, dtype=torch.uint8, device=device),
                                            requires_grad=False)

    def _gemm(self, x):
        if self.kernel not in self._PACKERS:
            raise ValueError(f"Unsupported kernel type {self.kernel}")
        if not self.weight_reshaped:
            self.reshape_weight(self.w_inner_k)
        on_right = self.kernel == "linear_y_f16RM_x_f16RM_W_mx4TC"
        a, b = (x, self.weight) if on_right else (self.weight, x)
        return _T.tinygemm_y_f16RM_x_f16RM_w_mx4TC(a, b, self.group_size, self.exponents, on_right)
