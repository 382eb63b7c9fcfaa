"""tools/tswgen/run_emu4.py -- run the round-6 loop (kernel4.py: 12 waves x 3 rows) in the CPU emulator against the oracle.
usage: python -m tools.tswgen.run_emu4 [B H W n_wg norm sparse seed [linear_ncu]]"""
import sys
import time

import numpy as np

from . import kernel4 as K
from .emu import Emu, EmuError
from .plan import plan_bands
from .plan4 import build_plan, build_plan_linear
from .run_emu import normalized_planes


def run_case(B, H, W, n_wg, norm=0, sparse=False, seed=0, zero_patch=False, verbose=True, sched=True, linear=None, cfg=None, neg_sparse=True):
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
        if neg_sparse:
            sp[:, :, ::5, 3::17] *= -1   # negative sparse values: m = -1 (cspn.py:64 sign())
    if zero_patch:
        g[:, :, H // 2:H // 2 + 3, 40:48] = 0
    n_iter = 24
    prog = K.build(dict(norm=norm, sparse=sparse, **(cfg or {})), sched=sched)
    g_dev = normalized_planes(g, 0) if norm == 3 else g
    nb = len(plan_bands(W, n_iter))
    if linear:
        lp, hdr, tab = build_plan_linear(B, H, W, n_iter, linear, xcd=False)
        n_wg = lp.n_wg
    else:
        n_wg = -(-n_wg // nb) * nb   # whole groups of nb workgroups
        hdr, tab = build_plan(B, H, W, n_iter, n_wg)

    def al(n):
        return (n + 4095) // 4096 * 4096
    off, cur = {}, 8192
    for name, arr in (("gd", g_dev), ("blur", blur), ("sp", sp), ("out", np.zeros_like(blur))):
        if arr is None:
            off[name] = 4096
            continue
        off[name] = cur
        cur += al(arr.nbytes) + 4096
    mem = np.zeros(cur + 4096, np.uint8)
    mem.view(np.float32)[:] = np.nan
    for name, arr in (("gd", g_dev), ("blur", blur), ("sp", sp)):
        if arr is not None:
            mem[off[name]:off[name] + arr.nbytes] = arr.view(np.uint8).ravel()
    t0 = time.time()
    tot = 0
    icount = {}
    for wg in range(n_wg):
        if hdr[wg, 0] == 0:
            continue
        emu = Emu(prog, mem, K.LDS_BYTES, nwaves=K.NW)
        flat = np.ascontiguousarray(tab[wg], np.uint32).ravel()
        emu.lds[K.LDS_TAB // 4:K.LDS_TAB // 4 + flat.size] = flat
        for w in emu.waves:
            w.v[0] = np.arange(64, dtype=np.uint32)

            def set64(r, val):
                w.s[r.i] = val & 0xffffffff
                w.s[r.i + 1] = val >> 32
            set64(K.S_GD, off["gd"])
            set64(K.S_BLUR, off["blur"])
            set64(K.S_HIN, off["blur"])
            set64(K.S_SP, off["sp"])
            set64(K.S_OUT, off["out"])
            w.s[K.S_W4.i] = 4 * W
            w.s[K.S_HW4.i] = 4 * H * W
            w.s[K.S_LAST.i] = int(hdr[wg, 1])
            w.s[K.S_WV.i] = w.wid
        tot += emu.run()
        for w in emu.waves:
            if w.vm_q:
                raise EmuError("wave %d ended with %d vector-memory operations in flight" % (w.wid, len(w.vm_q)))
            for k, v in w.icount.items():
                icount[k] = icount.get(k, 0) + v
    out = mem[off["out"]:off["out"] + blur.nbytes].view(np.float32).reshape(blur.shape)
    ref = O.cspn2d_oracle(g, blur, sp, n_iter, ["8sum", "8sum_abs", "none", "8sum"][norm])
    nanmis = np.isnan(out) != np.isnan(ref)
    den = np.nanmax(np.abs(ref))
    err = np.nanmax(np.abs(out - ref)) / den if not nanmis.any() else np.inf
    if verbose:
        steps = int(hdr[:, 1].max()) + 1 + K.LEAD
        print("B%d H%d W%d wg%d norm%d sp%d: rel err %.3g  nan mismatch %d  (%d instr, %.1fs, %d NaNs in ref)" % (
            B, H, W, n_wg, norm, sparse, err, nanmis.sum(), tot, time.time() - t0, np.isnan(ref).sum()))
        nv = sum(v for k, v in icount.items() if k.startswith("v_"))
        ns = sum(v for k, v in icount.items() if k.startswith("s_") and k not in ("s_waitcnt", "s_barrier", "s_nop"))
        nn = icount.get("s_nop", 0)
        nm = sum(v for k, v in icount.items() if k.startswith("ds_") or k.startswith("global_"))
        print("   per wave-step: VALU %.1f SALU %.1f nop %.1f mem %.1f (steps %d)" % (
            nv / K.NW / steps / n_wg, ns / K.NW / steps / n_wg, nn / K.NW / steps / n_wg, nm / K.NW / steps / n_wg, steps))
    return err, nanmis.sum(), out, ref


if __name__ == "__main__":
    a = sys.argv[1:]
    B, H, W, n_wg = (int(a[0]), int(a[1]), int(a[2]), int(a[3])) if len(a) >= 4 else (1, 12, 256, 1)
    norm = int(a[4]) if len(a) > 4 else 0
    sparse = bool(int(a[5])) if len(a) > 5 else False
    seed = int(a[6]) if len(a) > 6 else 0
    linear = int(a[7]) if len(a) > 7 else None
    try:
        run_case(B, H, W, n_wg, norm, sparse, seed, linear=linear)
    except EmuError as ex:
        print("EMU ERROR:", ex)
        sys.exit(1)
