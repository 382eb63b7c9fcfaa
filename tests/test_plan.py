"""The row-descriptor planner of the assembly loop (tools/tswgen/plan.py = numpy twin of cspn2d_plan_kernel; the GPU test
test_asm_plan_table_matches_python_planner holds the device table to it): every pixel is owned by exactly one
(workgroup, row, column range), halo rows stay inside their image, offsets address the right element."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.tswgen import kernel as K  # noqa: E402
from tools.tswgen.plan import LinearPlan, build_plan, build_plan_linear, plan_bands, plan_geo  # noqa: E402

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


LINEAR_CASES = [
    # B, H, W, CUs
    (64, 304, 1216, 256), (32, 304, 1216, 256), (8, 304, 1216, 256), (16, 228, 304, 256), (3, 33, 304, 256), (1, 150, 516, 4),
    (2, 60, 304, 3), (5, 77, 772, 64), (1, 7, 256, 256), (7, 304, 1216, 304),
]


@pytest.mark.parametrize("B,H,W,ncu", LINEAR_CASES)
def test_linear_plan_every_pixel_owned_exactly_once(B, H, W, ncu):
    """round 4, the forward passes' plan: one contiguous piece of the (chunk, band, row) order per CU"""
    n_iter = 24
    lp, hdr, tab = build_plan_linear(B, H, W, n_iter, ncu)
    assert lp.cut[0] == 0 and lp.cut[-1] == lp.total and all(a <= b for a, b in zip(lp.cut, lp.cut[1:]))
    assert max(lp.stream_len(a, b) for a, b in zip(lp.cut, lp.cut[1:])) == lp.L == tab.shape[1] - K.PADF - K.PADB
    owned = np.zeros((B, H, W), np.int32)
    for g in range(lp.n_wg):
        Q = int(hdr[g, 0])
        assert int(hdr[g, 2]) == -1
        assert not tab[g, :K.PADF].any() and not tab[g, K.PADF + Q:].any()
        for d in tab[g, K.PADF:K.PADF + Q]:
            flags = int(d[3])
            if not flags & 1:
                assert not d.any()
                continue
            e = int(d[2]) // 4
            b, rem = divmod(e, H * W)
            y, p0 = divmod(rem, W)
            lo, hi = (flags >> 8) & 0xfff, (flags >> 20) & 0xfff
            assert (p0, p0 + lo, p0 + hi) in lp.bands and p0 + 256 <= W
            assert bool(flags >> K.F_FIRST & 1) == (p0 == 0) and bool(flags >> K.F_LAST & 1) == (p0 + 256 == W)
            if flags >> K.F_OWNED & 1:
                owned[b, y, p0 + lo:p0 + hi] += 1
    assert owned.min() == 1 and owned.max() == 1


def test_linear_plan_of_the_benchmarked_shape():
    """BASELINE config 3 at 64 images on 256 CUs: 256 pieces of exactly 1.5 (image, band) units, one mid-image cut each = 481 stream
    rows (513 with 42 band groups on 252 CUs); chunks of 3 images, so the pieces of neighbouring bands cover the same rows"""
    lp = LinearPlan(64, 304, 1216, 24, 256)
    assert (lp.kimg, lp.L, lp.n_wg, lp.per_xcd) == (3, 481, 256, 32)
    assert lp.cut == [456 * i for i in range(257)]
    lens = [lp.stream_len(a, b) for a, b in zip(lp.cut, lp.cut[1:])]
    assert min(lens) == max(lens) == 481
    for p in (0, 1, 2, 3, 13):   # pieces 2 b, 2 b + 1 of a chunk = band b; band b + 1 covers the same rows two pieces later
        ra, rb = lp.runs(lp.cut[p], lp.cut[p + 1]), lp.runs(lp.cut[p + 2], lp.cut[p + 3])
        if len(ra) == 1 and len(rb) == 1 and rb[0][0] == ra[0][0] + 1:
            assert ra[0][1:] == rb[0][1:]
    last = [lp.runs(lp.cut[p], lp.cut[p + 1]) for p in range(252, 256)]   # the 64th image: 6 band units over 4 CUs
    assert [len(r) for r in last] == [2, 2, 2, 2] and last[0][0] == (0, 63 * 304, 64 * 304)
    legacy_share = -(-64 * 304 // 42)
    assert legacy_share + 48 + 1 >= 513 > lp.L


@pytest.mark.parametrize("B,H,W,ncu", LINEAR_CASES + [(12000, 64, 256, 256)])
def test_cxx_planner_matches_the_numpy_twin(B, H, W, ncu):
    """the optimiser as the library runs it on the host (csrc/cspn2d_tsw_plan.h make_geo_linear, reached through the hook library;
    no GPU involved) against tools/tswgen/plan.py"""
    import ctypes
    from cspn_amd import _lib
    hooks = _lib.load_hooks()
    cut = (ctypes.c_int * 258)()
    kimg, stride = ctypes.c_int(), ctypes.c_int()
    n = hooks.cspn_debug_tsw_plan_cuts(B, H, W, ncu, 1, cut, ctypes.byref(kimg), ctypes.byref(stride))
    if B * H * len(plan_bands(W, 24)) // min(ncu, 256) > K.TAB_MAX_ROWS - K.PADF - K.PADB:
        assert n == 0   # too long for a table: the library falls back to band groups
        return
    lp = LinearPlan(B, H, W, 24, ncu)
    assert n == lp.n_wg and kimg.value == lp.kimg and stride.value == lp.stride
    assert list(cut[:n + 1]) == lp.cut


def test_library_plan_cache_returns_the_right_plan_after_evictions():
    """the library keeps the last four forward plans per host thread (the optimiser costs ~1 ms): six shapes visited twice in a
    row-robin must each get their own plan every time (cspn_debug_tsw_plan_geo goes through the cached entry point; no GPU: the CU
    count defaults to 256)"""
    import ctypes
    from cspn_amd import _lib
    hooks = _lib.load_hooks()
    shapes = [(64, 304, 1216), (16, 228, 304), (3, 33, 304), (8, 304, 1216), (5, 77, 772), (2, 100, 1216)]
    seen = {}
    for rnd in range(2):
        for B, H, W in shapes:
            info = (ctypes.c_int * 8)()
            assert hooks.cspn_debug_tsw_plan_geo(B, H, W, 0, 0, info) == 0
            lp = LinearPlan(B, H, W, 24, 256)
            got = tuple(info)
            assert got[:4] == (1, lp.n_wg, lp.stride, lp.kimg), (B, H, W, got)
            assert seen.setdefault((B, H, W), got) == got
