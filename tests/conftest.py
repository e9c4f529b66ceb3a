import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


# ---- torch <-> bit-pattern helpers shared by the tests ----

def bits16(t):
    """torch bf16/fp16 tensor (any device) -> numpy uint16 bit patterns"""
    import torch

    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def from_bits16(a, dtype, device="cpu"):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16).copy()).view(dtype).to(device)


def bf16_ulp(x):
    """size of one bf16 ulp at |x| (float32 numpy)"""
    x = np.abs(np.asarray(x, np.float32))
    e = np.floor(np.log2(np.maximum(x, 1e-38)))
    return np.exp2(e - 7).astype(np.float32)


@pytest.fixture
def reference_numerics():
    """Run a test with bit-identical dequantised weights (TG_NUM_REFERENCE): the kernels whose results the tight
    oracle tolerances of test_gpu_parity.py / test_gpu_decode.py were written for.  The default (fast, group-scaled)
    numerics have their own tests in test_gpu_fast.py."""
    import any4_amd

    with any4_amd.numerics("reference"):
        yield


@pytest.fixture
def reference_weight_format():
    """Weights-on-the-left tensors in the reference's own Aint4 word order (any4_amd.weight_format("reference")): the modules
    whose tests were written for those words -- bit-exact packer tests, the Aint4 kernels -- run under it; the default
    (row-per-lane order, TG_WFMT_ROWS) has its own tests in test_gpu_aside.py."""
    import any4_amd

    with any4_amd.weight_format("reference"):
        yield
