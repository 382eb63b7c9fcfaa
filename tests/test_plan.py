"""The row-descriptor planner of the assembly loop (tools/tswgen/plan.py = numpy twin of cspn2d_plan_kernel; the GPU test
test_asm_plan_table_matches_python_planner holds the device table to it): every pixel is owned by exactly one
(workgroup, row, column range), halo rows stay inside their image, offsets address the right element."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.tswgen import kernel as K  # noqa: E402
from tools.tswgen.plan import build_plan, plan_bands, plan_geo  # noqa: E402

CASES = [
    # B, H, W, max_wg (CUs), xcd placement?
    (1, 7, 256, 256, False), (3, 33, 304, 256, False), (2, 100, 1216, 256, False), (64, 304, 1216, 256, True),
    (16, 228, 304, 256, True), (5, 19, 516, 64, False), (1, 1, 260, 256, False), (40, 50, 772, 256, True),
]


@pytest.mark.parametrize("B,H,W,max_wg,xcd", CASES)
def test_every_pixel_owned_exactly_once(B, H, W, max_wg, xcd):
    n_iter = 24
    bands = plan_bands(W, n_iter)
    nb = len(bands)
    assert bands[0][0] == 0 and bands[-1][0] + 256 == W and bands[0][1] == 0 and bands[-1][2] == W
    for (p0, lo, hi), (_, lo2, _) in zip(bands, bands[1:] + [(0, W, 0)]):
        assert hi == lo2 and p0 % 4 == 0 and lo % 4 == 0 and lo - p0 in (0,) + tuple(range(24, 257)) and p0 + 256 - hi >= (0 if hi == W else 24)
    n_wg, stride = plan_geo(B, H, W, n_iter, max_wg)
    xcd_geo = None
    if xcd and n_wg == (max_wg // nb) * nb and (max_wg // 8) // nb >= 1:
        per_xcd = max_wg // 8
        gpx = per_xcd // nb
        extra = (8 * (per_xcd - gpx * nb)) // nb
        xcd_geo = (gpx, extra, per_xcd)
        n_wg = 8 * per_xcd
    hdr, tab = build_plan(B, H, W, n_iter, n_wg, xcd_geo)
    assert tab.shape[1] <= K.TAB_MAX_ROWS
    owned = np.zeros((B, H, W), np.int32)
    for g in range(n_wg):
        Q = int(hdr[g, 0])
        lo, hi = int(hdr[g, 2]) & 0xffff, int(hdr[g, 2]) >> 16
        rows = tab[g, K.PADF:K.PADF + Q]
        assert not tab[g, :K.PADF].any() and not tab[g, K.PADF + Q:].any()   # padding rows are inactive
        if Q:
            assert int(hdr[g, 1]) == 3 * ((Q - 1) >> 2) + ((Q - 1) & 3) + n_iter
        for d in rows:
            flags = int(d[3])
            if not flags & 1:
                assert not d.any()   # separator
                continue
            boff = int(d[2])
            goff = int(d[0]) | (int(d[1]) << 32)
            e = boff // 4
            b, rem = divmod(e, H * W)
            y, p0 = divmod(rem, W)
            assert goff == 4 * (b * 8 * H * W + y * W + p0) and p0 + 256 <= W
            assert bool(flags >> K.F_UP & 1) == (y + 1 < H) and bool(flags >> K.F_DN & 1) == (y >= 1)
            assert bool(flags >> K.F_FIRST & 1) == (p0 == 0) and bool(flags >> K.F_LAST & 1) == (p0 + 256 == W)
            assert ((flags >> 8) & 0x1ff, (flags >> 20) & 0x1ff) == (lo, hi)
            if flags >> K.F_OWNED & 1:
                owned[b, y, p0 + lo:p0 + hi] += 1
    assert owned.min() == 1 and owned.max() == 1
