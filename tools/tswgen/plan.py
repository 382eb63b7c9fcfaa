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


# ---- the linear plan of the forward passes (cspn2d_tsw_plan.h kind 1; this is its numpy twin) ---------------------------------
MAX_CUT = 256
MIN_ROWS_PER_WG = 16


class LinearPlan(object):
    """nb * B * H band rows in one order -- chunks of kimg images, inside a chunk band after band -- cut into one contiguous piece
    per CU so that the longest stream (rows + warm-up / cool-down rows + separators) is as short as possible."""

    def __init__(self, B, H, W, n_iter, ncu, xcd=True):
        self.B, self.H, self.W, self.n_iter = B, H, W, n_iter
        self.bands = plan_bands(W, n_iter)
        self.nb = nb = len(self.bands)
        self.total = total = B * H * nb
        n = max(1, min(ncu, MAX_CUT, total // MIN_ROWS_PER_WG))
        rho = total / n
        best, self.kimg = 1e30, 1
        for k in range(1, min(8, B) + 1):
            m = k * H / rho
            mr = max(1.0, float(int(m + 0.5)))
            d = abs(m - mr) / mr
            if d < best - 1e-9:
                best, self.kimg = d, k
        lo, hi = max(1, total // n), total // n + 1 + (total // n // H + 2) * (2 * n_iter + 1)
        while self.greedy(n, hi) is None:
            hi += hi
        while lo < hi:
            m = lo + (hi - lo) // 2
            if self.greedy(n, m) is not None:
                hi = m
            else:
                lo = m + 1
        self.L = lo
        self.cut = self.greedy(n, lo)
        self.n_wg = n
        self.stride = PADF + lo + PADB
        self.ok = self.stride <= TAB_MAX_ROWS
        self.per_xcd = ncu // 8 if (xcd and n == ncu and ncu % 8 == 0) else 0

    def runs(self, xa, xe):
        """-> [(band, first row, row behind the last)] (rows global: image * H + y)"""
        Rc, out, x = self.kimg * self.H, [], xa
        while x < xe:
            c = x // (Rc * self.nb)
            r0 = c * Rc
            rows_c = min(Rc, self.B * self.H - r0)
            bi, off = divmod(x - c * Rc * self.nb, rows_c)
            n = min(xe - x, rows_c - off)
            out.append((bi, r0 + off, r0 + off + n))
            x += n
        return out

    def segments(self, xa, xe):
        segs = []
        for bi, ra, rb in self.runs(xa, xe):
            r = ra
            while r < rb:
                b, y0 = divmod(r, self.H)
                y1 = min(self.H, y0 + rb - r)
                segs.append((b, bi, max(0, y0 - self.n_iter), min(self.H, y1 + self.n_iter), y0, y1))
                r += y1 - y0
        return segs

    def stream_len(self, xa, xe):
        segs = self.segments(xa, xe)
        return sum(s[3] - s[2] for s in segs) + max(0, len(segs) - 1)

    def greedy(self, n, L):
        cut, xa = [0], 0
        while xa < self.total:
            if len(cut) - 1 == n:
                return None
            lo, hi = xa, min(self.total, xa + L)
            while lo < hi:
                m = lo + (hi - lo + 1) // 2
                if self.stream_len(xa, m) <= L:
                    lo = m
                else:
                    hi = m - 1
            if lo == xa:
                return None
            xa = lo
            cut.append(xa)
        return cut + [self.total] * (n + 1 - len(cut))

    def piece_of_wg(self, wg):
        return (wg & 7) * self.per_xcd + (wg >> 3) if self.per_xcd else wg

    def wg_segments(self, wg):
        p = self.piece_of_wg(wg)
        return self.segments(self.cut[p], self.cut[p + 1])


def build_plan_linear(B, H, W, n_iter, ncu, xcd=True):
    """the descriptor tables of a linear plan -> (LinearPlan, header, table); header dword 2 = -1: the loop takes a row's owned
    columns from its descriptor"""
    lp = LinearPlan(B, H, W, n_iter, ncu, xcd)
    assert lp.ok
    hdr = np.zeros((lp.n_wg, 4), np.int32)
    tab = np.zeros((lp.n_wg, lp.stride, 4), np.uint32)
    for wg in range(lp.n_wg):
        fill_plan_rows(hdr, tab, wg, lp.wg_segments(wg), lp.bands, H, W, n_iter, -1)
    return lp, hdr, tab


def fill_plan_rows(hdr, tab, g, segs, bands, H, W, n_iter, lohi):
    rows = stream_of(segs)
    Q = len(rows)
    assert PADF + Q + PADB <= tab.shape[1]
    hdr[g, 0] = Q
    hdr[g, 1] = (3 * ((Q - 1) >> 2) + ((Q - 1) & 3) + n_iter) if Q else -1
    hdr[g, 2] = lohi
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


def build_plan(B, H, W, n_iter, n_wg, xcd=None):
    """band groups (kind 0) -> (header int32[n_wg][4] = Q, last_step, lo | hi << 16, 0 ; table uint32[n_wg][stride][4])"""
    bands = plan_bands(W, n_iter)
    ng = wg_group(0, len(bands), n_wg, xcd)[2]
    stride = stride_of(-(-(B * H) // ng), H, n_iter)
    assert stride <= TAB_MAX_ROWS
    hdr = np.zeros((n_wg, 4), np.int32)
    tab = np.zeros((n_wg, stride, 4), np.uint32)
    for g in range(n_wg):
        segs = share_segments(B, H, W, n_iter, bands, g, n_wg, xcd)
        p0b, lob, hib = bands[wg_group(g, len(bands), n_wg, xcd)[1]]
        fill_plan_rows(hdr, tab, g, segs, bands, H, W, n_iter, (lob - p0b) | ((hib - p0b) << 16))   # owned columns of the workgroup's band
    return hdr, tab
