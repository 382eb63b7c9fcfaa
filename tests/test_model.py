"""CPU: the executable specification of the fused kernel's schedule (tools/tsw_model.py:
8-wave x 4-slot time-skewed ring, push-form accumulators, planner) against the oracle."""
import os
import sys

import numpy as np
import pytest

from helpers import make_inputs, rel_err
from oracle import cspn2d_oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import tsw_model as tm  # noqa: E402


@pytest.mark.parametrize("B,H,W,N,norm,sp,nwg", [
    (1, 40, 64, 24, "8sum", True, 1),
    (2, 37, 53, 24, "8sum", True, 3),       # ragged sizes, streams cross image boundaries
    (1, 30, 300, 24, "8sum_abs", True, 2),  # two bands with 24-column halos
    (1, 9, 530, 12, "8sum", False, 5),      # three bands, n_iter < 24, more workgroups than comfortable
    (3, 5, 20, 24, "8sum", True, 1),        # images shorter than the pipeline depth
    (1, 70, 40, 7, "none", False, 4),
    (1, 1, 1, 3, "8sum", False, 1),
    (2, 64, 256, 1, "8sum", True, 7),
])
def test_model_matches_oracle(B, H, W, N, norm, sp, nwg):
    g, h, s = make_inputs(B, H, W, seed=H * 7 + W, sparse=sp, neg=sp)
    if norm == "none":
        g = g.abs() / g.abs().sum(1, keepdim=True)
    ref = cspn2d_oracle(g, h, s, N, norm)
    out, _ = tm.cspn2d_model(g.numpy(), h.numpy(), None if s is None else s.numpy(), N,
                             {"8sum": 0, "8sum_abs": 1, "none": 2}[norm], n_wg=nwg)
    assert rel_err(out, ref) <= 2e-6


def test_model_nan_pattern():
    g, h, s = make_inputs(1, 24, 40, seed=5)
    g[:, :, 8:13, 10:16] = 0.0
    ref = cspn2d_oracle(g, h, s, 4)
    out, _ = tm.cspn2d_model(g.numpy(), h.numpy(), s.numpy(), 4, 0, n_wg=2)
    assert np.isnan(ref).sum() > 0 and rel_err(out, ref) <= 2e-6


def test_planner_covers_every_pixel_once():
    for (B, H, W, N, G) in [(64, 304, 1216, 24, 256), (16, 228, 304, 24, 256), (1, 10, 1000, 24, 9), (2, 7, 256, 3, 5)]:
        cover = np.zeros((B, H, W), np.int32)
        for segs in tm.plan_streams(B, H, W, N, G):
            for (b, p0, lo, hi, ys, ye, y0, y1) in segs:
                assert p0 % 4 == 0 and lo % 4 == 0 and p0 <= lo < hi <= min(W, p0 + tm.BW)
                assert ys <= y0 < y1 <= ye and ys == max(0, y0 - N) and ye == min(H, y1 + N)
                if lo > 0:
                    assert lo - p0 >= N          # left halo deep enough
                if hi < W:
                    assert p0 + tm.BW - hi >= N  # right halo deep enough
                cover[b, y0:y1, lo:hi] += 1
        assert (cover == 1).all()


def test_cost_model_reproduces_the_two_measured_plans():
    """tools/tsw_cost_model.py (profiles/r04_decomposition_model.md): time = steps x (instructions per SIMD-step x 5 + 260) cycles /
    2.15 GHz, with the instruction census of the GENERATED loop and the steps of the PLANNER -- must stay within 3 % of the two plans
    measured on one MI355X in round 4 (band groups 0.2905 ms, linear plan 0.2772 ms) when either of them changes"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools import tsw_cost_model as M
    from tools.tswgen.plan import LinearPlan
    c = M.census(dict(norm=0))
    I = 2 * sum(c.values())
    assert 60 <= c["pk_fma"] <= 70 and 240 <= I <= 270, (c, I)
    s_new = M.steps_of(LinearPlan(64, 304, 1216, 24, 256).L)
    s_old = M.steps_of(-(-64 * 304 // 42) + 48 + 1)
    assert (s_old, s_new) == (408, 384)
    assert abs(M.t_ms(s_old, I) - 0.2905) <= 0.03 * 0.2905
    assert abs(M.t_ms(s_new, I) - 0.2772) <= 0.03 * 0.2772
