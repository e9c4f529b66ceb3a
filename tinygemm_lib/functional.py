"""Drop-in name for the reference's tinygemm_lib/functional.py; implementation in any4_amd/functional.py."""
import tinygemm  # noqa: F401  (hard dependency, as in the reference: functional.py:8)
from any4_amd.functional import *  # noqa: F401,F403
from any4_amd.functional import valid_tinygemm_kernel_call  # noqa: F401
