"""tools/tswgen/run_emu.py -- run the generated kernel in the CPU emulator against the oracle.
usage: python -m tools.tswgen.run_emu [B H W n_wg norm sparse hin seed]"""
import sys
import time

import numpy as np

from . import kernel as K
from .emu import Emu, EmuError
from .plan import build_plan


def run_case(B, H, W, n_wg, norm=0, sparse=False, hin=False, seed=0, zero_patch=False, verbose=True, sched=True, hist=False, hist_every=1,
             s8=False, elastic=False, sched_seed=None, linear=None, cfg_extra=None, early_n=None):
    """early_n = n (1..23): the `early` variant -- a pass that delivers level n (checked against the oracle run for n iterations).
    linear = number of CUs: the forward passes' linear plan (tools/tswgen/plan.py LinearPlan: one contiguous piece of the
    band-row order per CU; a piece may continue in the next band) instead of band groups"""
    sys.path.insert(0, ".")
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((B, 8, H, W)).astype(np.float32)
    if norm == 2:
        g = np.abs(g)
        g /= g.sum(1, keepdims=True) + 0.3
    blur = (rng.random((B, 1, H, W)) * 10).astype(np.float32)
    sp = None
    if sparse:
        m = rng.random((B, 1, H, W)) < 0.05
        sp = (m * (rng.random((B, 1, H, W)) * 10 + 0.1)).astype(np.float32)
    if zero_patch:
        g[:, :, H // 2:H // 2 + 3, 40:48] = 0
    n_iter = 24
    hinv = None
    if hin:  # emulate a second pass: level-0 values differ from blur
        hinv = (rng.random((B, 1, H, W)) * 10).astype(np.float32)
    K.configure(elastic)
    prog = K.build(dict(norm=norm, sparse=sparse, hin=hin, hist=hist, hist_every=hist_every, s8=s8, elastic=elastic, **({"act_and": False} if s8 else {}),
                        **({"early": True} if early_n else {}), **(cfg_extra or {})), sched=sched)
    g_dev = sited8(g, norm) if s8 else g   # what the kernel reads as its guidance tensor
    if norm == 3:   # prenorm: the kernel reads what reference affinity_normalization returns for the RAW guidance g ('8sum'); the oracle
        g_dev = normalized_planes(g, 0)    # below still sees the raw tensors

    histbuf = np.full((23 + 8, B, 1, H, W), np.nan, np.float32) if hist else None   # + the 8 folded coefficient planes
    from .plan import plan_bands
    nb = len(plan_bands(W, n_iter))
    if linear:
        from .plan import build_plan_linear
        lp, hdr, tab = build_plan_linear(B, H, W, n_iter, linear, xcd=False)
        n_wg = lp.n_wg
    else:
        n_wg = -(-n_wg // nb) * nb   # whole groups of nb workgroups
        hdr, tab = build_plan(B, H, W, n_iter, n_wg)
    # global memory image
    def al(n):
        return (n + 4095) // 4096 * 4096
    off, cur = {}, 8192
    for name, arr in (("gd", g_dev), ("blur", blur), ("hin", hinv), ("sp", sp), ("out", np.zeros_like(blur)), ("plan", tab), ("hist", histbuf)):
        if arr is None:
            off[name] = 4096
            continue
        off[name] = cur
        cur += al(arr.nbytes) + 4096 + (8192 if name == "plan" else 0)
    mem = np.zeros(cur + 4096, np.uint8)
    mem.view(np.float32)[:] = np.nan
    for name, arr in (("gd", g_dev), ("blur", blur), ("hin", hinv), ("sp", sp), ("plan", tab)):
        if arr is not None:
            mem[off[name]:off[name] + arr.nbytes] = arr.view(np.uint8).ravel()
    t0 = time.time()
    tot = 0
    icount = {}
    for wg in range(n_wg):
        if hdr[wg, 0] == 0:
            continue
        emu = Emu(prog, mem, K.LDS_BYTES)
        if elastic:   # no barrier epochs to check races in: the tags order the accesses (and the emulator runs every wave
            emu.check_races = False   # until it has to wait for a tag; sched_seed: random order and random stalls on top)
            if sched_seed is not None:
                emu.sched_rng = np.random.default_rng(sched_seed + wg)
        fill_table(emu, tab[wg])
        for w in emu.waves:
            w.v[0] = np.arange(64, dtype=np.uint32)
            def set64(r, val):
                w.s[r.i] = val & 0xffffffff
                w.s[r.i + 1] = val >> 32
            set64(K.S_GD, off["gd"])
            set64(K.S_BLUR, off["blur"])
            set64(K.S_HIN, off["hin"])
            set64(K.S_SP, off["sp"])
            set64(K.S_OUT, off["out"])
            set64(K.S_PLAN, off["plan"] + wg * tab.shape[1] * 16)
            w.s[K.S_NROWS.i] = max(0, g.nbytes - 7 * 4 * H * W - 2 * 4 * W)   # S_GLAST (cfg pf); S_NROWS when tab_in_lds = False
            w.s[K.S_LOHI.i] = int(hdr[wg, 2]) & 0xffffffff
            if hist:
                set64(K.S_HIST, off["hist"])
                set64(K.S_HSTRIDE, blur.nbytes)
            if early_n:
                assert 1 <= early_n <= 23 and not hin
                w.s[K.S_NIT.i] = early_n
                w.s[K.S_EMASK.i] = sum(1 << ((early_n + j) % 24) for j in range(4))
            w.s[K.S_W4.i] = 4 * W
            w.s[K.S_HW4.i] = 4 * H * W
            w.s[K.S_LAST.i] = int(hdr[wg, 1])
            w.s[K.S_WV.i] = w.wid
        tot += emu.run()
        for w in emu.waves:
            for k, v in w.icount.items():
                icount[k] = icount.get(k, 0) + v
    K.configure(False)
    out = mem[off["out"]:off["out"] + blur.nbytes].view(np.float32).reshape(blur.shape)
    if hin:
        # oracle for a continuation pass: propagate hinv with blur as H0 ... the oracle has no such entry; emulate with numpy
        ref = ref_hin(g, blur, sp, hinv, n_iter, norm)
    else:
        ref = O.cspn2d_oracle(g, blur, sp, early_n or n_iter, ["8sum", "8sum_abs", "none", "8sum"][norm])
    if hist:
        assert not hin
        npl = 24 // hist_every - 1   # level planes: levels hist_every, 2 hist_every ..
        wfb = mem[off["hist"] + npl * blur.nbytes:off["hist"] + (npl + 8) * blur.nbytes].view(np.float32).reshape(8, B, H, W)
        wf_ref = folded_planes(g, sp, norm)
        assert np.array_equal(np.isnan(wfb), np.isnan(wf_ref))
        assert np.nanmax(np.abs(wfb - wf_ref)) <= 1e-6 * max(1.0, np.nanmax(np.abs(wf_ref))), "folded coefficient planes"
        hb = mem[off["hist"]:off["hist"] + npl * blur.nbytes].view(np.float32).reshape(npl, B, H, W // 4, 4)
        hb = hb[..., [0, 2, 3, 1]].reshape(npl, B, 1, H, W)   # stored in register order (c0,c3,c1,c2) per 4-column group
        worst = 0.0
        for i in range(npl):
            lv = (i + 1) * hist_every
            r = O.cspn2d_oracle(g, blur, sp, lv, ["8sum", "8sum_abs", "none", "8sum"][norm])
            assert np.array_equal(np.isnan(hb[i]), np.isnan(r)), "history level %d: NaN pattern" % lv
            worst = max(worst, float(np.nanmax(np.abs(hb[i] - r)) / np.nanmax(np.abs(r))))
        if verbose:
            print("   history levels %s: worst rel err %.3g" % (list(range(hist_every, 24, hist_every)), worst))
        assert worst <= 1e-5
    nanmis = np.isnan(out) != np.isnan(ref)
    den = np.nanmax(np.abs(ref))
    err = np.nanmax(np.abs(out - ref)) / den if not nanmis.any() else np.inf
    if verbose:
        steps = int(hdr[:, 1].max()) + 1
        print("B%d H%d W%d wg%d norm%d sp%d hin%d: rel err %.3g  nan mismatch %d  (%d instr, %.1fs, %d NaNs in ref)" % (
            B, H, W, n_wg, norm, sparse, hin, err, nanmis.sum(), tot, time.time() - t0, np.isnan(ref).sum()))
        nv = sum(v for k, v in icount.items() if k.startswith("v_"))
        ns = sum(v for k, v in icount.items() if k.startswith("s_") and k not in ("s_waitcnt", "s_barrier", "s_nop"))
        nn = icount.get("s_nop", 0)
        nm = sum(v for k, v in icount.items() if k.startswith("ds_") or k.startswith("global_"))
        print("   per wave-step: VALU %.1f SALU %.1f nop %.1f mem %.1f (steps %d)" % (
            nv / 8 / steps / n_wg, ns / 8 / steps / n_wg, nn / 8 / steps / n_wg, nm / 8 / steps / n_wg, steps))
    return err, nanmis.sum(), out, ref


def sited8(g, norm):
    """the pre-sited, pair-interleaved guidance layout of cfg s8: [B][H][W/2][8][2], element (k, e) of pair xp = G_k(2 xp + e) with
    G_k(p) = g_k(p + off_k) (zero outside the image) for the normalising modes, g_k(p) for norm 'none'
    (numpy twin of cspn2d_guidance_to_sited8_f32)"""
    B, _, H, W = g.shape
    out = np.zeros((B, H, W // 2, 8, 2), np.float32)
    for k in range(8):
        if norm == 2:
            G = g[:, k]
        else:
            pad = np.zeros((B, H + 2, W + 2), np.float32)
            pad[:, 1:-1, 1:-1] = g[:, k]
            G = pad[:, 1 + K.DY[k]:1 + K.DY[k] + H, 1 + K.DX[k]:1 + K.DX[k] + W]
        out[:, :, :, k, 0] = G[:, :, 0::2]
        out[:, :, :, k, 1] = G[:, :, 1::2]
    return out


def normalized_planes(g, norm):
    """gate_wb of reference cspn.py:85-144 as [B,8,H,W]: w_k(p) = G_k(p) / sum_j |G_j(p)|, G_k(p) = g~_k(p + off_k), zero outside
    the image (IEEE division: 0/0 = NaN) -- numpy twin of cspn2d_normalize_f32"""
    B, _, H, W = g.shape
    gp = np.abs(g) if norm == 1 else g
    G = np.zeros_like(g)
    for k in range(8):
        pad = np.zeros((B, H + 2, W + 2), np.float32)
        pad[:, 1:-1, 1:-1] = gp[:, k]
        G[:, k] = pad[:, 1 + K.DY[k]:1 + K.DY[k] + H, 1 + K.DX[k]:1 + K.DX[k] + W]
    with np.errstate(all="ignore"):
        return (G / np.abs(G).sum(1, keepdims=True)).astype(np.float32)


def fill_table(emu, tab_wg):
    """the C++ part of the kernel (cspn2d_tsw.hip: tsw_fill_table) writes the workgroup's descriptor table into LDS before
    the generated block starts; the emulator does the same with the table of tools/tswgen/plan.py"""
    flat = np.ascontiguousarray(tab_wg, np.uint32).ravel()
    emu.lds[K.LDS_TAB // 4:K.LDS_TAB // 4 + flat.size] = flat


def folded_planes(g, sp, norm):
    """w'_k(p) = (1 - m) G_k / sum|G| (reference cspn.py:85-144 + the mask of :81), planar [8][B,H,W]"""
    B, _, H, W = g.shape
    gp = np.abs(g) if norm == 1 else g
    G = np.zeros((8, B, H, W), np.float32)
    for k in range(8):
        if norm == 2:
            G[k] = gp[:, k]
        else:
            pad = np.zeros((B, H + 2, W + 2), np.float32)
            pad[:, 1:-1, 1:-1] = gp[:, k]
            G[k] = pad[:, 1 + K.DY[k]:1 + K.DY[k] + H, 1 + K.DX[k]:1 + K.DX[k] + W]
    with np.errstate(all="ignore"):
        w = G if norm == 2 else G / np.abs(G).sum(0, keepdims=True)
        if sp is not None:
            w = (1 - np.sign(sp[:, 0]))[None] * w
    return w.astype(np.float32)


def ref_hin(g, blur, sp, hin, n_iter, norm):
    """numpy reference for a continuation pass (H_t starts at hin, H0 = blur), restating oracle/cspn_oracle.c"""
    if norm == 3:
        norm = 0   # (prenorm: g is the raw guidance, the kernel got its '8sum' normalisation)
    B, _, H, W = g.shape
    DY, DX = K.DY, K.DX
    gp = np.abs(g) if norm == 1 else g
    G = np.zeros_like(g)
    for k in range(8):
        if norm == 2:
            G[:, k] = gp[:, k]
        else:
            pad = np.zeros((B, H + 2, W + 2), np.float32)
            pad[:, 1:-1, 1:-1] = gp[:, k]
            G[:, k] = pad[:, 1 + DY[k]:1 + DY[k] + H, 1 + DX[k]:1 + DX[k] + W]
    with np.errstate(all="ignore"):
        if norm == 2:
            w = G
            gs = None
        else:
            w = G / np.abs(G).sum(1, keepdims=True)
            gs = w.sum(1, keepdims=True)
        h = hin.copy()
        m = np.sign(sp) if sp is not None else None
        for _ in range(n_iter):
            pad = np.zeros((B, H + 2, W + 2), np.float32)
            pad[:, 1:-1, 1:-1] = h[:, 0]
            acc = np.zeros_like(h)
            for k in range(8):
                acc[:, 0] += w[:, k] * pad[:, 1 + DY[k]:1 + DY[k] + H, 1 + DX[k]:1 + DX[k] + W]
            if gs is not None:
                acc = (1 - gs) * blur + acc
            if m is not None:
                acc = (1 - m) * acc + m * blur
            h = acc.astype(np.float32)
    return h


if __name__ == "__main__":
    a = sys.argv[1:]
    B, H, W, n_wg = (int(a[0]), int(a[1]), int(a[2]), int(a[3])) if len(a) >= 4 else (1, 12, 256, 1)
    norm = int(a[4]) if len(a) > 4 else 0
    sparse = bool(int(a[5])) if len(a) > 5 else False
    hin = bool(int(a[6])) if len(a) > 6 else False
    seed = int(a[7]) if len(a) > 7 else 0
    try:
        run_case(B, H, W, n_wg, norm, sparse, hin, seed)
    except EmuError as ex:
        print("EMU ERROR:", ex)
        sys.exit(1)
