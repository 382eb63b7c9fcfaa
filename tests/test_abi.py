"""CPU: the C-ABI library loads, exports every symbol include/cspn_amd.h declares and
rejects bad arguments before touching the GPU.  No compute happens here."""
import ctypes
import os
import re

import pytest
import torch

import cspn_amd
from cspn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cspn_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cspn\w*)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    lib = cspn_amd.load()
    syms = _declared_symbols()
    assert {"cspn2d_forward_f32", "cspn3d_forward_f32", "cspn2d_workspace_bytes", "cspn_abi_version",
            "cspn_last_error"} <= set(syms)
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s


def test_abi_version_and_workspace_sizes():
    lib = cspn_amd.load()
    assert lib.cspn_abi_version() == _lib.ABI_VERSION
    assert lib.cspn2d_workspace_bytes(1, 10, 10, 0) == 0
    assert lib.cspn2d_workspace_bytes(2, 228, 304, 24) >= 2 * 228 * 304 * 4
    assert lib.cspn3d_workspace_bytes(1, 4, 8, 8, 3) >= 27 * 4 * 8 * 8 * 4
    assert lib.cspn2d_auto_algo(64, 304, 1216, 24) in (_lib.ALGOS["stepwise"], _lib.ALGOS["fused"])
    # round 5: W % 4 != 0 no longer means one launch per iteration -- AUTO pads the rows to a multiple of 4 columns in the workspace
    assert lib.cspn2d_auto_algo(64, 304, 1218, 24) == _lib.ALGOS["fused_padded"] == 4
    assert lib.cspn2d_auto_algo(1, 7, 5, 3) == _lib.ALGOS["fused_padded"]
    assert lib.cspn2d_workspace_bytes(2, 30, 302, 24) >= 11 * 2 * 30 * 304 * 4
    assert lib.cspn2d_workspace_bytes(2, 30, 302, 30) >= 12 * 2 * 30 * 304 * 4   # + the ping buffer of the second pass


def test_argument_errors_are_reported_without_gpu():
    lib = cspn_amd.load()
    rc = lib.cspn2d_forward_f32(None, None, None, None, 1, 4, 4, 3, 0, None, 0, None)
    assert rc == -1 and b"null" in lib.cspn_last_error()
    rc = lib.cspn2d_forward_f32(None, None, None, None, 1, 0, 4, 3, 0, None, 0, None)
    assert rc == -1 and b"shape" in lib.cspn_last_error()
    fake = ctypes.c_void_p(4096)
    rc = lib.cspn2d_forward_f32(fake, fake, None, fake, 1, 4, 4, 3, 7, None, 0, None)
    assert rc == -1 and b"norm_type" in lib.cspn_last_error()
    rc = lib.cspn2d_forward_f32(fake, fake, None, fake, 1, 4, 5, 3, 0, None, 0, None)  # W%4 != 0 -> stepwise
    assert rc == -2 and b"workspace" in lib.cspn_last_error()
    rc = lib.cspn2d_forward_f32_algo(fake, fake, None, fake, 1, 4, 5, 3, 0, 2, None, 0, None)  # fused refused
    assert rc == -3 and b"fused" in lib.cspn_last_error()
    rc = lib.cspn2d_forward_f32(fake, fake, None, fake, 1, 4, 4, -1, 0, None, 0, None)
    assert rc == -1
    rc = lib.cspn3d_forward_f32(fake, fake, None, fake, 1, 2, 4, 4, 3, 0, None, 0, None)
    assert rc == -2


def test_backward_and_aux_argument_errors_are_reported_without_gpu():
    lib = cspn_amd.load()
    fake = ctypes.c_void_p(4096)
    assert lib.cspn2d_backward_workspace_bytes(2, 64, 320, 24) >= (5 + 8 + 5 + 2) * 2 * 64 * 320 * 4   # checkpoints of both sweeps, coefficients, A_0, a scratch output
    assert lib.cspn2d_backward_workspace_bytes(2, 9, 10, 5) >= (4 + 5) * 2 * 9 * 10 * 4
    assert lib.cspn2d_backward_workspace_bytes(1, 4, 4, 0) == 0
    rc = lib.cspn2d_backward_f32(fake, fake, None, None, fake, fake, 1, 4, 4, 3, 0, None, 0, None)
    assert rc == -1 and b"grad_out" in lib.cspn_last_error()
    rc = lib.cspn2d_backward_f32(fake, fake, None, fake, fake, fake, 1, 4, 4, 0, 0, None, 0, None)
    assert rc == -1 and b"n_iter" in lib.cspn_last_error()
    rc = lib.cspn2d_backward_f32(fake, fake, None, fake, fake, fake, 1, 4, 4, 3, 0, None, 0, None)
    assert rc == -2 and b"workspace" in lib.cspn_last_error()
    assert lib.cspn_metrics_workspace_bytes(1000) >= 80
    rc = lib.cspn_metrics_f32(fake, fake, 1000, fake, None, 0, None)
    assert rc == -2
    rc = lib.cspn_metrics_f32(None, fake, 1000, fake, fake, 1 << 20, None)
    assert rc == -1
    assert lib.cspn_unpool_f32(fake, fake, 1, 0, 4, 2, None) == -1
    assert lib.cspn_unpool_backward_f32(None, fake, 1, 4, 4, 2, None) == -1
    assert lib.cspn_l1_backward_f32(fake, fake, None, fake, fake, 10, None) == -1


def test_module_mirrors_reference_interface():
    m = cspn_amd.Affinity_Propagate(24, 3, "8sum")  # positional like torch_resnet_cspn_nyu.py:344-347
    assert m.prop_time == 24 and m.prop_kernel == 3 and m.norm_type == "8sum"
    assert list(m.state_dict().keys()) == [] and list(m.parameters()) == []
    with pytest.raises(AssertionError):
        cspn_amd.Affinity_Propagate(24, 5)  # cspn.py:33
    with pytest.raises(AssertionError):
        cspn_amd.Affinity_Propagate(24, 3, "4sum")  # cspn.py:36
    x = torch.zeros(1, 1, 4, 4)
    assert cspn_amd.Affinity_Propagate(0, 3)(torch.zeros(1, 8, 4, 4), x) is x  # N=0 identity, same object
    assert m(torch.zeros(1, 8, 4, 4), x, None, n_iter=0) is x


def test_no_cpu_fallback():
    m = cspn_amd.Affinity_Propagate(3, 3)
    with pytest.raises(cspn_amd.CspnError):
        m(torch.zeros(1, 8, 4, 4), torch.zeros(1, 1, 4, 4))


def test_dropin_module_name():
    import importlib.util
    spec = importlib.util.spec_from_file_location("cspn", os.path.join(ROOT, "cspn_amd", "dropin", "cspn.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.Affinity_Propagate is cspn_amd.Affinity_Propagate


def test_graft_entry_build_runs_and_checks_the_abi_version():
    """what the driver runs as its build check (no GPU needed): compiles whatever is out of date, loads the library, compares its ABI version
    with the binding's and the header's"""
    import re
    import __graft_entry__ as ge
    from cspn_amd import _lib
    ge.build()
    hdr = open(os.path.join(ROOT, "include", "cspn_amd.h")).read()
    assert int(re.search(r"#define CSPN_ABI_VERSION (\d+)", hdr).group(1)) == _lib.ABI_VERSION == _lib.load().cspn_abi_version()
