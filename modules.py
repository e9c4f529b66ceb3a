"""Drop-in name for the reference's top-level modules.py; implementation in any4_amd/modules.py."""
import tinygemm_lib.functional  # noqa: F401
from any4_amd.modules import Any4Linear, Int4Linear, Int8Linear, MX4Linear, NF4Linear  # noqa: F401
