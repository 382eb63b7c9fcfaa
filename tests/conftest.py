import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """a plain `pytest tests` on a host without a GPU skips the gpu-marked tests instead of failing in them"""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no HIP device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "cspn2d_golden.npz")
    z = np.load(path)
    cases = {}
    for key in z.files:
        name, field = key.split("/")
        cases.setdefault(name, {})[field] = z[key]
    return cases


@pytest.fixture(scope="session")
def norm_golden():
    """gate_wb / gate_sum of the unmodified reference's affinity_normalization for cases of `golden` (tests/golden/make_norm_golden.py)"""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "cspn2d_norm_golden.npz"))
    cases = {}
    for key in z.files:
        name, field = key.split("/")
        cases.setdefault(name, {})[field] = z[key]
    return cases


@pytest.fixture(scope="session", autouse=True)
def _built_lib():
    """The HIP library must exist before any test touches cspn_amd (hipcc cross-compiles on CPU)."""
    import cspn_amd
    if not os.path.exists(cspn_amd._lib.LIB_PATH):
        cspn_amd.build()
    yield
