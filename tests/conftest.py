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


def bits_equal(a, b):
    """Bit-for-bit equality of fp64 arrays, treating NaN == NaN and +0 == -0."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


@pytest.fixture(scope="session")
def oracle():
    from oracle.bindings import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def dlibref():
    from oracle.bindings import DlibRef
    if not DlibRef.available():
        pytest.skip("real-dlib reference build not present (no /root/reference and no prebuilt _ref)")
    return DlibRef()
