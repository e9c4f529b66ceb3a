"""any4_amd -- MI355X-native tinygemm: the any4 / int4 / nf4 / mx4 W4A16 small-batch GEMM hot path of
facebookresearch/any4 as hand-written gfx950 HIP kernels behind a C ABI, with the reference's
`torch.ops.tinygemm.*`, `tinygemm_lib.functional` and `modules` API on top.

Importing this package loads the HIP library and registers the ops; it raises ImportError if the
library has not been built (`python -m any4_amd.build`).  There is no CPU fallback.
"""
import sys as _sys

# `python -m any4_amd.build` imports this package before it can (re)build the library: do not load a missing or
# stale .so in that one invocation (everything else fails loudly, see _lib.load()).
_argv = getattr(_sys, "orig_argv", [])
_BUILD_CLI = any(a == "-m" and b == "any4_amd.build" for a, b in zip(_argv, _argv[1:]))

if not _BUILD_CLI:
    from . import ops as _ops  # noqa: F401
    from . import functional, modules, utils  # noqa: F401
    from .modules import Any4Linear, Int4Linear, Int8Linear  # noqa: F401
    from .ops import (get_auto_relayout, get_numerics, get_weight_format, numerics, set_auto_relayout, set_numerics,  # noqa: F401
                      set_weight_format, weight_format)

__all__ = ["functional", "modules", "utils", "Any4Linear", "Int4Linear", "Int8Linear", "get_numerics", "set_numerics", "numerics",
           "get_weight_format", "set_weight_format", "weight_format", "get_auto_relayout", "set_auto_relayout"]
