#!/usr/bin/env python
"""tools/tsw_model.py -- executable specification (numpy, row-vector granularity) of the
schedule used by the fused kernel cspn_amd/csrc/cspn2d_fused.hip ("time-skewed wave ring").

It models, for ONE workgroup: 8 waves x 4 row slots, the push-form accumulators (N1/N2),
the flat spots at block boundaries, the double-buffered boundary exchange through LDS,
cooking (normalise + fold) one..two steps ahead into 4 weight buffers + an 8-deep H0 ring,
injection / retirement / inactive separator rows, and the host-side planner that cuts
[B x bands x H] into per-workgroup streams.  tests/test_model.py checks it against the
oracle; the HIP kernel is a transcription of `run_workgroup`.

Not part of the product.  Vocabulary:
  stream row q  -> block beta = q // 4, slot j = q % 4, wave = beta % 8, phase phi = 3*beta + j
  row q completes level n at step phi(q) + n (level 0 = injection of H0)
"""
import numpy as np

NW, R = 8, 4            # waves per workgroup, row slots per wave
LV = NW * (R - 1)       # 24 = max propagation levels fused in one pass
BW = 256                # physical band width (64 lanes x 4 columns)
DY = [1, 1, 1, 0, 0, -1, -1, -1]
DX = [1, 0, -1, 1, -1, 1, 0, -1]


def phase(q):
    return 3 * (q // 4) + (q % 4)


# ----------------------------------------------------------------------------- planner
def plan_bands(W, n_iter):
    """-> list of (p0, own_lo, own_hi): physical start column and owned output range."""
    h = 4 * ((n_iter + 3) // 4)  # horizontal halo, kept a multiple of 4 for 16-B alignment
    if W <= BW:
        return [(0, 0, W)]
    bands, lo = [], 0
    while lo < W:
        p0 = 0 if lo == 0 else lo - h
        hi = W if p0 + BW >= W else p0 + BW - h
        bands.append((p0, lo, hi))
        lo = hi
    return bands


def plan_streams(B, H, W, n_iter, n_wg):
    """Cut the B x bands x H row space into n_wg contiguous shares.  Each share is a list of
    segments (b, p0, own_lo, own_hi, ys, ye, y0, y1): stream rows [ys,ye) of image b / band p0
    are fed through the pipeline, rows [y0,y1) are written."""
    bands = plan_bands(W, n_iter)
    units = [(b, bd) for b in range(B) for bd in bands]
    total = len(units) * H
    n_wg = max(1, min(n_wg, total))
    streams = []
    for g in range(n_wg):
        r0, r1 = g * total // n_wg, (g + 1) * total // n_wg
        segs = []
        r = r0
        while r < r1:
            u, y0 = divmod(r, H)
            y1 = min(H, y0 + (r1 - r))
            b, (p0, lo, hi) = units[u]
            segs.append((b, p0, lo, hi, max(0, y0 - n_iter), min(H, y1 + n_iter), y0, y1))
            r += y1 - y0
        streams.append(segs)
    return streams


def stream_rows(segs):
    """Flatten segments into the stream: entries (seg index, y) or None for an inactive
    separator row between segments."""
    rows = []
    for i, s in enumerate(segs):
        if i:
            rows.append(None)
        rows.extend((i, y) for y in range(s[4], s[5]))
    return rows


# ----------------------------------------------------------------------------- cooking
def cook_row(g, blur, sparse, b, y, p0, H, W, norm):
    """Folded coefficients of image row y, columns [p0, p0+BW): w'[8][BW], c'[BW], H0[BW].
    Neighbour-sited gather + abs-sum normalisation + centre/mask folding
    (reference cspn.py:85-144 and :76,:81).  Columns outside the image
    (and everything the band cannot see: x-1 < p0, x+1 >= p0+BW) read as zero."""
    xs = np.arange(p0, p0 + BW)
    inimg = xs < W
    G = np.zeros((8, BW), np.float32)
    for k in range(8):
        yy = y + DY[k]
        if norm == 2:
            v = np.where(inimg, g[b, k, y, np.minimum(xs, W - 1)], 0)
        else:
            xx = xs + DX[k]
            ok = inimg & (xx >= 0) & (xx < W) & (0 <= yy < H) & (xx >= p0) & (xx < p0 + BW)
            v = np.where(ok, g[b, k, min(max(yy, 0), H - 1), np.clip(xx, 0, W - 1)], 0)
            if norm == 1:
                v = np.abs(v)
        G[k] = v
    h0 = np.where(inimg, blur[b, 0, y, np.minimum(xs, W - 1)], 0).astype(np.float32)
    with np.errstate(all="ignore"):
        if norm == 2:
            w = G.copy()
            c = np.zeros(BW, np.float32)
        else:
            S = np.abs(G).sum(0, dtype=np.float32)
            inv = (np.float32(1) / S).astype(np.float32)
            w = (G * inv).astype(np.float32)
            c = ((np.float32(1) - w.sum(0, dtype=np.float32)) * h0).astype(np.float32)
        if sparse is not None:
            m = np.sign(np.where(inimg, sparse[b, 0, y, np.minimum(xs, W - 1)], 0)).astype(np.float32)
            w = ((1 - m) * w).astype(np.float32)
            c = ((1 - m) * c + m * h0).astype(np.float32)
    w[:, ~inimg] = 0
    c[~inimg] = 0
    return w, c, h0


def shl(v):  # value of the right neighbour (x+1); band edge reads 0 (DPP bound_ctrl)
    return np.concatenate([v[1:], [0]]).astype(np.float32)


def shr(v):  # value of the left neighbour (x-1)
    return np.concatenate([[0], v[:-1]]).astype(np.float32)


def push(w, ks, V):
    """sum over k in ks of w[k] * V(x + DX[k])"""
    acc = np.zeros(BW, np.float32)
    with np.errstate(all="ignore"):
        for k in ks:
            src = V if DX[k] == 0 else (shl(V) if DX[k] == 1 else shr(V))
            acc = acc + w[k] * src
    return acc.astype(np.float32)


BELOW, SELF, ABOVE = (0, 1, 2), (3, 4), (5, 6, 7)


# ----------------------------------------------------------------------------- one workgroup
def run_workgroup(segs, g, blur, sparse, out, n_iter, norm):
    B, _, H, W = g.shape
    assert 1 <= n_iter <= LV
    rows = stream_rows(segs)
    Q = len(rows)
    if Q == 0:
        return 0
    zeros = lambda: np.zeros(BW, np.float32)  # noqa: E731
    # per (wave, slot) state
    Wt = [[np.zeros((8, BW), np.float32) for _ in range(R)] for _ in range(NW)]
    Cp = [[zeros() for _ in range(R)] for _ in range(NW)]
    S = [[[zeros() for _ in range(R)] for _ in range(NW)] for _ in range(2)]  # S[parity][wave][slot]
    active = [[False] * R for _ in range(NW)]
    meta = [[None] * R for _ in range(NW)]      # (seg, y) held by the slot
    # LDS
    bnd = np.zeros((2, NW, 2, BW), np.float32)  # [parity][wave][0=TOP row (slot 0) | 1=BOT row (slot 3)]
    cooked = [None] * 4                          # weight buffers, indexed q % 4
    h0ring = [zeros() for _ in range(8)]         # indexed q % 8
    cooked_tag = [None] * 4
    h0_tag = [None] * 8

    def cook(q):
        if q < 0:
            return
        if q < Q and rows[q] is not None:
            si, y = rows[q]
            b, p0 = segs[si][0], segs[si][1]
            w, c, h0 = cook_row(g, blur, sparse, b, y, p0, H, W, norm)
            cooked[q % 4] = (w, c, True, rows[q])
        else:
            cooked[q % 4] = (np.zeros((8, BW), np.float32), zeros(), False, None)
            h0 = zeros()
        h0ring[q % 8] = h0
        cooked_tag[q % 4] = q
        h0_tag[q % 8] = q

    last_step = phase(Q - 1) + n_iter
    cook(0)  # prologue
    for tau in range(0, last_step + 1):
        par = tau & 1
        new_bnd = np.zeros((NW, 2, BW), np.float32)
        for wv in range(NW):
            N1, N2 = S[par][wv], S[par ^ 1][wv]
            top = bnd[par ^ 1][(wv + NW - 1) % NW][1]   # previous block's slot 3, published last step
            bot = bnd[par ^ 1][(wv + 1) % NW][0]        # next block's slot 0, published last step
            V = [None] * R
            injected = [False] * R
            for j in (3, 2, 1, 0):
                u = (tau - 3 * wv - j) % LV
                u1 = LV if u == 0 else u
                below = bot if j == 3 else V[j + 1]
                acc = N1[j]
                if j == 0:
                    acc = acc + push(Wt[wv][0], ABOVE, top)
                with np.errstate(all="ignore"):
                    v = (acc + push(Wt[wv][j], BELOW, below)).astype(np.float32)
                if not active[wv][j]:
                    v = zeros()
                if u1 == n_iter and active[wv][j] and meta[wv][j] is not None:
                    si, y = meta[wv][j]
                    b, p0, lo, hi, ys, ye, y0, y1 = segs[si]
                    if y0 <= y < y1:
                        out[b, 0, y, lo:hi] = v[lo - p0:hi - p0]
                if u1 == LV:
                    # the stream row entering this slot now: phase(q) == tau
                    beta = (tau - j) // 3
                    q = 4 * beta + j
                    assert (tau - j) % 3 == 0 and beta % NW == wv and phase(q) == tau
                    if 0 <= q:
                        assert cooked_tag[q % 4] == q and h0_tag[q % 8] == q, (tau, q, cooked_tag, h0_tag)
                        w, c, act, m = cooked[q % 4]
                        Wt[wv][j], Cp[wv][j], active[wv][j], meta[wv][j] = w, c, act, m
                        v = h0ring[q % 8].copy()
                        n2 = c + push(w, SELF, v)
                        if j > 0:
                            assert q - 1 < 0 or h0_tag[(q - 1) % 8] == q - 1
                            n2 = n2 + push(w, ABOVE, h0ring[(q - 1) % 8])
                        N2[j] = n2.astype(np.float32)
                        injected[j] = True
                V[j] = v
                if j == 3:
                    new_bnd[wv][1] = v
                if j == 0:
                    new_bnd[wv][0] = v
            # self pushes
            for j in range(R):
                if injected[j]:
                    continue
                with np.errstate(all="ignore"):
                    if j == 0:
                        N2[0] = (Cp[wv][0] + push(Wt[wv][0], SELF, V[0])).astype(np.float32)
                    else:
                        N2[j] = (N2[j] + push(Wt[wv][j], SELF, V[j])).astype(np.float32)
            # above pushes initialise the slot below's NEXT accumulator in the register set V[j+1] vacates
            for j in (2, 1, 0):
                with np.errstate(all="ignore"):
                    N1[j + 1] = (Cp[wv][j + 1] + push(Wt[wv][j + 1], ABOVE, V[j])).astype(np.float32)
            # slot 0's vacated N1 becomes next step's N2_0, initialised by next step's self push
        bnd[par] = new_bnd
        # cooking for later injections (2 of every 3 steps)
        if tau % 3 == 0:
            beta = tau // 3
            cook(4 * beta + 1)
            cook(4 * beta + 2)
        elif tau % 3 == 2:
            beta = (tau - 2) // 3
            cook(4 * beta + 3)
            cook(4 * beta + 4)
    return last_step + 1


def cspn2d_model(g, blur, sparse, n_iter, norm=0, n_wg=4):
    """Whole-problem driver: plan, run every workgroup, handle n_iter > 24 by chaining passes."""
    g = np.asarray(g, np.float32)
    blur = np.asarray(blur, np.float32)
    sparse = None if sparse is None else np.asarray(sparse, np.float32)
    B, _, H, W = g.shape
    assert n_iter <= LV, "model covers a single pass (n_iter <= 24)"
    out = np.full_like(blur, np.nan)
    steps = 0
    for segs in plan_streams(B, H, W, n_iter, n_wg):
        steps += run_workgroup(segs, g, blur, sparse, out, n_iter, norm)
    return out, steps
