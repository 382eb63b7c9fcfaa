"""tools/tswgen/plan3.py -- numpy twin of the compact row-descriptor table of the round-3 loop (kernel3.py; the device version is
tsw3_fill_table in cspn_amd/csrc/cspn2d_tsw3.hip): the same streams as plan.py, 4 bytes per stream row:
flags (ACTIVE, UP, DN, OWNED) | (image << ybits | y) << 4, ybits = bits needed for H - 1; everything else a workgroup needs is a
per-band constant (first column, first / last band, owned columns)."""
import numpy as np

from tools.tswgen import plan as P2
from .kernel3 import PADF, PADB, TAB_MAX_ROWS, F_ACTIVE, F_UP, F_DN, F_OWNED, G_FIRST, G_LAST

BW = 256


def ybits_of(H):
    return max(1, int(H - 1).bit_length())


def stride_of(share, H, n_iter):
    return PADF + share + (share // H + 2) * (2 * n_iter + 1) + PADB


def plan_geo(B, H, W, n_iter, max_wg, min_rows=16):
    nb = len(P2.plan_bands(W, n_iter))
    total = B * H
    ng = max(1, min(max_wg // nb, total // min_rows))
    while True:
        share = -(-total // ng)
        if stride_of(share, H, n_iter) <= TAB_MAX_ROWS:
            return ng * nb, stride_of(share, H, n_iter)
        ng += max(1, ng // 8)


def build_plan(B, H, W, n_iter, n_wg, xcd=None):
    """-> (header int32[n_wg][4] = Q, last_step, lo | hi << 16, 4 * p0 ; geom int32[n_wg] ; table uint32[n_wg][stride])"""
    bands = P2.plan_bands(W, n_iter)
    nb = len(bands)
    ng = P2.wg_group(0, nb, n_wg, xcd)[2]
    stride = stride_of(-(-(B * H) // ng), H, n_iter)
    assert stride <= TAB_MAX_ROWS, (stride, TAB_MAX_ROWS)
    yb = ybits_of(H)
    assert (B << yb) < (1 << 28)
    hdr = np.zeros((n_wg, 4), np.int32)
    geom = np.zeros(n_wg, np.int32)
    tab = np.zeros((n_wg, stride), np.uint32)
    for g in range(n_wg):
        segs = P2.share_segments(B, H, W, n_iter, bands, g, n_wg, xcd)
        rows = P2.stream_of(segs)
        Q = len(rows)
        assert PADF + Q + PADB <= stride
        p0b, lob, hib = bands[P2.wg_group(g, nb, n_wg, xcd)[1]]
        hdr[g] = (Q, (3 * ((Q - 1) >> 2) + ((Q - 1) & 3) + n_iter) if Q else -1, (lob - p0b) | ((hib - p0b) << 16), 4 * p0b)
        geom[g] = yb | ((p0b == 0) << G_FIRST) | ((p0b + BW == W) << G_LAST)
        for q, r in enumerate(rows):
            if r is None:
                continue
            si, y = r
            b, bi, ys, ye, y0, y1 = segs[si]
            flags = (1 << F_ACTIVE) | ((y + 1 < H) << F_UP) | ((y >= 1) << F_DN) | ((y0 <= y < y1) << F_OWNED)
            tab[g, PADF + q] = flags | (((b << yb) | y) << 4)
    return hdr, geom, tab
