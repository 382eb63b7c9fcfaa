"""tools/tswgen/run_emu3.py -- run the round-3 loop (kernel3.py) in the CPU emulator against the oracle.
usage: python -m tools.tswgen.run_emu3 [B H W n_wg norm sparse hin seed]"""
import sys
import time

import numpy as np

from . import kernel3 as K
from tools.tswgen.emu import Emu, EmuError
from tools.tswgen.plan import plan_bands
from .plan3 import build_plan
from tools.tswgen.run_emu import ref_hin


def run_case(B, H, W, n_wg, norm=0, sparse=False, hin=False, seed=0, zero_patch=False, verbose=True, sched=True, cfg=None):
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
        sp[:, :, ::5, 3::17] *= -1   # negative sparse values: m = -1 (cspn.py:64 sign())
    if zero_patch:
        g[:, :, H // 2:H // 2 + 3, 40:48] = 0
    n_iter = 24
    hinv = None
    if hin:  # emulate a second pass: level-0 values differ from blur
        hinv = (rng.random((B, 1, H, W)) * 10).astype(np.float32)
    prog = K.build(dict(norm=norm, sparse=sparse, hin=hin, **(cfg or {})), sched=sched)
    nb = len(plan_bands(W, n_iter))
    n_wg = -(-n_wg // nb) * nb   # whole groups of nb workgroups
    hdr, geom, tab = build_plan(B, H, W, n_iter, n_wg)

    def al(n):
        return (n + 4095) // 4096 * 4096
    off, cur = {}, 8192
    for name, arr in (("gd", g), ("blur", blur), ("hin", hinv), ("sp", sp), ("out", np.zeros_like(blur))):
        if arr is None:
            off[name] = 4096
            continue
        off[name] = cur
        cur += al(arr.nbytes) + 4096
    mem = np.zeros(cur + 4096, np.uint8)
    mem.view(np.float32)[:] = np.nan
    for name, arr in (("gd", g), ("blur", blur), ("hin", hinv), ("sp", sp)):
        if arr is not None:
            mem[off[name]:off[name] + arr.nbytes] = arr.view(np.uint8).ravel()
    t0 = time.time()
    tot = 0
    icount = {}
    for wg in range(n_wg):
        if hdr[wg, 0] == 0:
            continue
        emu = Emu(prog, mem, K.LDS_BYTES)
        flat = np.ascontiguousarray(tab[wg], np.uint32)
        emu.lds[K.LDS_TAB // 4:K.LDS_TAB // 4 + flat.size] = flat
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
            w.s[K.S_LOHI.i] = int(hdr[wg, 2])
            w.s[K.S_P04.i] = int(hdr[wg, 3])
            w.s[K.S_GEOM.i] = int(geom[wg])
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
    if hin:
        ref = ref_hin(g, blur, sp, hinv, n_iter, norm)
    else:
        ref = O.cspn2d_oracle(g, blur, sp, n_iter, ["8sum", "8sum_abs", "none"][norm])
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
