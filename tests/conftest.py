"""Shared pytest configuration.  `-m "not gpu"` runs here (no GPU); `-m gpu` runs on an MI355X."""
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


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def bits_equal(a, b, dtype=np.float64):
    """Bit-for-bit equality of floating-point arrays: the IEEE bit patterns are compared (so +0 and
    -0 differ), except that any NaN equals any NaN (a NaN's payload and sign are not part of the
    result the reference defines)."""
    a = np.ascontiguousarray(a, dtype=dtype)
    b = np.ascontiguousarray(b, dtype=dtype)
    if a.shape != b.shape:
        return False
    iv = np.uint64 if np.dtype(dtype).itemsize == 8 else np.uint32
    return bool(np.all((a.view(iv) == b.view(iv)) | (np.isnan(a) & np.isnan(b))))


def bits_equal32(a, b):
    return bits_equal(a, b, dtype=np.float32)


@pytest.fixture(scope="session")
def oracle():
    from oracle.bindings import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def oracle32():
    """The restatement typed float (oracle/liboracle_mpc_f32.so): checker of the fp32 kernels."""
    from oracle.bindings import Oracle
    return Oracle("f32")


@pytest.fixture(scope="session")
def dlibref():
    from oracle.bindings import DlibRef
    if not DlibRef.available():
        pytest.skip("real-dlib reference build not present (no /root/reference and no prebuilt _ref)")
    return DlibRef()
