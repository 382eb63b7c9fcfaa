"""tools/tswgen/plan.py -- reference (numpy) version of the row-descriptor table that cspn2d_plan_kernel builds on the
device: which image rows a workgroup streams, in which order, and what it does with each (cook flags, output range).
Mirrors tools/tsw_model.py's planner except that the last band is shifted left to end exactly at the image edge, so that
every band is 256 real columns wide."""
import numpy as np

from .kernel import PADF, PADB, TAB_MAX_ROWS, F_ACTIVE, F_UP, F_DN, F_FIRST, F_LAST, F_OWNED, F_PLAIN

BW = 256


def halo_of(n_iter):
    return 4 * ((n_iter + 3) // 4)


def plan_bands(W, n_iter):
    assert W >= BW and W % 4 == 0
    h = halo_of(n_iter)
    bands, lo = [], 0
    while True:
        p0 = 0 if lo == 0 else lo - h
        if p0 + BW >= W:
            bands.append((W - BW, lo, W))
            break
        hi = p0 + BW - h
        bands.append((p0, lo, hi))
        lo = hi
    return bands


def wg_group(g, nb, n_wg, xcd=None):
    """workgroup id -> (group index or None for an idle workgroup, band, number of groups); xcd = (gpx, extra, per_xcd): the
    XCD-aware placement of cspn2d_tsw.hip (workgroup id -> XCD round robin; a group's workgroups share one XCD's L2)"""
    if not xcd:
        return g // nb, g % nb, n_wg // nb
    gpx, extra, per_xcd = xcd
    ng = 8 * gpx + extra
    x, sl = g & 7, g >> 3
    if sl < gpx * nb:
        return x * gpx + sl // nb, sl % nb, ng
    t = (sl - gpx * nb) * 8 + x
    return (8 * gpx + t // nb if t < extra * nb else None), t % nb, ng


def share_segments(B, H, W, n_iter, bands, g, n_wg, xcd=None):
    """Workgroup g = nb * G + band: group G owns a contiguous range of the B*H image rows, its nb workgroups take one band
    each, so the workgroups that read overlapping columns of the same rows run side by side (their halo re-reads hit in
    cache instead of HBM)."""
    nb = len(bands)
    G, bi, ng = wg_group(g, nb, n_wg, xcd)
    if G is None:
        return []
    total = B * H
    r0, r1 = G * total // ng, (G + 1) * total // ng
    segs, r = [], r0
    while r < r1:
        b, y0 = divmod(r, H)
        y1 = min(H, y0 + (r1 - r))
        segs.append((b, bi, max(0, y0 - n_iter), min(H, y1 + n_iter), y0, y1))
        r += y1 - y0
    return segs


def stream_of(segs):
    rows = []
    for i, s in enumerate(segs):
        if i:
            rows.append(None)
        rows.extend((i, y) for y in range(s[2], s[3]))
    return rows


def stride_of(share, H, n_iter):
    return PADF + share + (share // H + 2) * (2 * n_iter + 1) + PADB


def plan_geo(B, H, W, n_iter, max_wg, min_rows=16):
    """-> (n_wg, stride): groups of nb workgroups; as many groups as fit on the CUs, more when a share's table would not
    fit in LDS"""
    nb = len(plan_bands(W, n_iter))
    total = B * H
    ng = max(1, min(max_wg // nb, total // min_rows))
    while True:
        share = -(-total // ng)
        if stride_of(share, H, n_iter) <= TAB_MAX_ROWS:
            return ng * nb, stride_of(share, H, n_iter)
        ng += max(1, ng // 8)


def build_plan(B, H, W, n_iter, n_wg, xcd=None):
    """-> (header int32[n_wg][4] = Q, last_step, lo | hi << 16, 0 ; table uint32[n_wg][stride][4])"""
    bands = plan_bands(W, n_iter)
    ng = wg_group(0, len(bands), n_wg, xcd)[2]
    stride = stride_of(-(-(B * H) // ng), H, n_iter)
    assert stride <= TAB_MAX_ROWS
    hdr = np.zeros((n_wg, 4), np.int32)
    tab = np.zeros((n_wg, stride, 4), np.uint32)
    for g in range(n_wg):
        segs = share_segments(B, H, W, n_iter, bands, g, n_wg, xcd)
        rows = stream_of(segs)
        Q = len(rows)
        assert PADF + Q + PADB <= stride
        hdr[g, 0] = Q
        hdr[g, 1] = (3 * ((Q - 1) >> 2) + ((Q - 1) & 3) + n_iter) if Q else -1
        p0b, lob, hib = bands[wg_group(g, len(bands), n_wg, xcd)[1]]
        hdr[g, 2] = (lob - p0b) | ((hib - p0b) << 16)   # owned columns of the workgroup's band
        for q, r in enumerate(rows):
            if r is None:
                continue
            si, y = r
            b, bi, ys, ye, y0, y1 = segs[si]
            p0, lo, hi = bands[bi]
            goff = 4 * (b * 8 * H * W + y * W + p0)
            boff = 4 * (b * H * W + y * W + p0)
            flags = (1 << F_ACTIVE) | ((y + 1 < H) << F_UP) | ((y >= 1) << F_DN) | ((p0 == 0) << F_FIRST) | \
                    ((p0 + BW == W) << F_LAST) | ((y0 <= y < y1) << F_OWNED)
            if 1 <= y < H - 1 and p0 > 0 and p0 + BW < W:
                flags |= 1 << F_PLAIN
            d = tab[g, PADF + q]
            d[0] = goff & 0xffffffff
            d[1] = goff >> 32
            d[2] = boff
            d[3] = flags | ((lo - p0) << 8) | ((hi - p0) << 20)
    return hdr, tab
