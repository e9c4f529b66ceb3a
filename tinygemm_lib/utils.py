"""Drop-in name for the reference's tinygemm_lib/utils.py; implementation in any4_amd/utils.py."""
from any4_amd.utils import (  # noqa: F401
    dequantize_mx4,
    expand_q_groups,
    extract_scales_and_zeros,
    group_quantize_tensor,
    quantize_mx4,
    round_to_mx4,
)
