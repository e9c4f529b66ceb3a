"""`import tinygemm` -- the name the reference's pybind stub exports (tinygemm_lib/TinyGemm.cpp:13-15).
Importing it registers torch.ops.tinygemm.*; the reference's callers use a successful import as the
signal that real kernels (not the pseudo-quantised path) are available (quantize.py:337, 829)."""
import any4_amd.ops as _ops  # noqa: F401

__doc__ = "tinygemm: low-bit GEMM library (MI355X / gfx950 HIP build)"
