"""CPU: the oracle (oracle/cspn_oracle.c) against the reference's own outputs."""
import os

import numpy as np
import pytest
import torch

from helpers import RTOL, make_inputs, rel_err
from oracle import cspn2d_gate_wb_oracle, cspn2d_oracle, cspn3d_oracle
from oracle import ref_harness

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NORMS = {0: "8sum", 1: "8sum_abs"}


def test_oracle_matches_golden(golden):
    assert len(golden) >= 10
    for name, c in golden.items():
        B, H, W, N, norm = [int(v) for v in c["meta"]]
        out = cspn2d_oracle(c["guidance"], c["blur"], c.get("sparse"), N, NORMS[norm])
        err = rel_err(out, c["out"])
        assert err <= 1e-5, (name, err)  # oracle is pinned an order tighter than the product gate


def test_gate_wb_oracle_and_prenorm_mode_match_golden(golden, norm_golden):
    """SURVEY 8f-2 (normalisation done by the producer): the contract is the reference's OWN intermediate, gate_wb of
    affinity_normalization (cspn.py:85-144).  tests/golden/cspn2d_norm_golden.npz holds it as the unmodified reference returned it
    (tests/golden/make_norm_golden.py).  (1) the oracle's restatement of that function reproduces it; (2) the oracle's 'prenorm'
    mode, fed the GOLDEN gate_wb, reproduces the golden outputs; (3) so does the numpy twin the emulator tests use."""
    from tools.tswgen.run_emu import normalized_planes
    assert len(norm_golden) >= 8
    for name, n in norm_golden.items():
        c = golden[name]
        B, H, W, N, norm = [int(v) for v in c["meta"]]
        wb = cspn2d_gate_wb_oracle(c["guidance"], NORMS[norm])
        assert np.array_equal(np.isnan(wb), np.isnan(n["gate_wb"])), name
        assert rel_err(wb, n["gate_wb"]) <= 1e-6, (name, rel_err(wb, n["gate_wb"]))
        tw = normalized_planes(c["guidance"], norm)
        assert np.array_equal(np.isnan(tw), np.isnan(n["gate_wb"])) and rel_err(tw, n["gate_wb"]) <= 1e-6, name
        out = cspn2d_oracle(n["gate_wb"], c["blur"], c.get("sparse"), N, "prenorm")
        assert np.array_equal(np.isnan(out), np.isnan(c["out"])), name
        assert rel_err(out, c["out"]) <= 1e-5, (name, rel_err(out, c["out"]))


@pytest.mark.skipif(not ref_harness.available(), reason="/root/reference not present (GPU box)")
def test_norm_golden_is_what_the_reference_returns(golden, norm_golden):
    for name, n in norm_golden.items():
        c = golden[name]
        wb, gs = ref_harness.reference_gate_wb(torch.from_numpy(c["guidance"]), NORMS[int(c["meta"][4])])
        assert np.array_equal(wb.numpy(), n["gate_wb"], equal_nan=True) and np.array_equal(gs.numpy(), n["gate_sum"], equal_nan=True), name


@pytest.mark.parametrize("channel_sum", ["conv3d", "sum"])
def test_torch_ops_baseline_matches_golden(golden, channel_sum):
    """the torch-op restatement that stands in for "the reference on this GPU" in tools/bench_torch_ops_baseline.py computes what
    the unmodified reference computed (NaN where it has NaN)"""
    from tools.torch_ops_baseline import affinity_propagate_torch_ops
    for name, c in golden.items():
        B, H, W, N, norm = [int(v) for v in c["meta"]]
        sp = torch.from_numpy(c["sparse"]) if "sparse" in c else None
        out = affinity_propagate_torch_ops(torch.from_numpy(c["guidance"]), torch.from_numpy(c["blur"]), sp, N, NORMS[norm], channel_sum).numpy()
        assert np.array_equal(np.isnan(out), np.isnan(c["out"])), name
        assert rel_err(out, c["out"]) <= 1e-5, (name, rel_err(out, c["out"]))


@pytest.mark.skipif(not ref_harness.available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("B,H,W,N,norm,sp", [(1, 228, 304, 12, "8sum", False), (2, 31, 45, 24, "8sum", True),
                                             (1, 64, 200, 24, "8sum_abs", True)])
def test_oracle_vs_live_reference(B, H, W, N, norm, sp):
    g, h, s = make_inputs(B, H, W, seed=7, sparse=sp, neg=sp)
    ref = ref_harness.reference_forward(g, h, s, N, norm).numpy()
    out = cspn2d_oracle(g, h, s, N, norm)
    assert rel_err(out, ref) <= 1e-5


def test_oracle_identity_and_fixed_point():
    g, h, s = make_inputs(2, 13, 17, seed=3)
    assert np.array_equal(cspn2d_oracle(g, h, s, 0), h.numpy())  # cspn.py:61,66,83
    const = torch.full_like(h, 3.25)
    for norm in ("8sum", "8sum_abs"):
        out = cspn2d_oracle(g, const, None, 7, norm)
        assert np.abs(out - 3.25).max() < 1e-5  # constant depth is a fixed point (SURVEY §4 KAT 2)


def test_oracle_mask_pins_blur():
    g, h, s = make_inputs(1, 20, 20, seed=5, p_sparse=0.2)
    out = cspn2d_oracle(g, h, s, 9)
    m = s.numpy() > 0
    assert m.sum() > 10 and np.array_equal(out[m], h.numpy()[m])  # cspn.py:81


def test_oracle_linear_in_depth():
    g, h1, s = make_inputs(1, 15, 19, seed=11)
    _, h2, _ = make_inputs(1, 15, 19, seed=12)
    o1, o2 = cspn2d_oracle(g, h1, s, 6), cspn2d_oracle(g, h2, s, 6)
    o12 = cspn2d_oracle(g, 2.0 * h1 - 0.5 * h2, s, 6)
    assert rel_err(o12, 2.0 * o1 - 0.5 * o2) < 1e-5


def test_oracle3d_inplane_gates_equal_2d():
    """A 3D run whose only non-zero gates are the eight dz==0 channels is the 2D op per slice."""
    gen = torch.Generator().manual_seed(2)
    B, D, H, W = 1, 3, 9, 11
    g2 = torch.randn(B * D, 8, H, W, generator=gen)
    h = torch.rand(B, 1, D, H, W, generator=gen) * 5
    g3 = torch.zeros(B, 26, D, H, W)
    g3[:, 9:17] = g2.view(B, D, 8, H, W).permute(0, 2, 1, 3, 4)
    for norm in ("8sum", "8sum_abs"):
        o3 = cspn3d_oracle(g3, h, None, 5, norm)
        o2 = cspn2d_oracle(g2, h.view(B * D, 1, H, W), None, 5, norm).reshape(B, 1, D, H, W)
        assert rel_err(o3, o2) < 1e-6


def test_oracle3d_constant_fixed_point_and_none_mode():
    gen = torch.Generator().manual_seed(4)
    g = torch.rand(1, 26, 4, 6, 7, generator=gen)
    const = torch.full((1, 1, 4, 6, 7), 2.0)
    out = cspn3d_oracle(g, const, None, 4, "8sum_abs")
    assert np.abs(out - 2.0).max() < 1e-5
    # Paddle contract (cspn_paddle/demo.py:47-49): caller normalises; interior voxels stay constant
    gn = g / g.sum(1, keepdim=True)
    out = cspn3d_oracle(gn, const, None, 1, "none")
    assert np.abs(out[0, 0, 1:-1, 1:-1, 1:-1] - 2.0).max() < 1e-5


@pytest.mark.skipif(not os.path.isdir("/root/reference/cspn_pytorch"), reason="the reference tree only exists in the build container")
def test_aux_golden_is_what_the_reference_files_produce(tmp_path, monkeypatch):
    """tests/golden/aux_golden.npz (metrics + loss vectors for the device mirrors) regenerates bit for bit from the unmodified
    reference utils.py / loss.py through the committed recipe"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_aux_golden", os.path.join(ROOT, "tests", "golden", "make_aux_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(mod, "OUT", str(tmp_path / "aux.npz"))
    mod.main()
    a, b = np.load(str(tmp_path / "aux.npz")), np.load(os.path.join(ROOT, "tests", "golden", "aux_golden.npz"))
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
