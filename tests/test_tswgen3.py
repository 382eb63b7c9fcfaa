"""The round-3 main loop (tools/tswgen/kernel3.py -> cspn_amd/csrc/cspn2d_tsw3_gen.inc: LDS-DMA row slots, in-place cooking,
4-byte row descriptors) executed instruction by instruction in the CPU emulator (tools/tswgen/emu.py) against the oracle:
register allocation, schedule, waitcnt placement, LDS races, LDS-DMA data touched before its wait or before the barrier behind
it, slot reuse.  Also: the committed include is what the generator emits, the compact planner owns every pixel once."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.tswgen import kernel3 as K  # noqa: E402
from tools.tswgen.isa import check_hazards  # noqa: E402
from tools.tswgen.plan import plan_bands  # noqa: E402
from tools.tswgen.plan3 import build_plan, plan_geo, ybits_of  # noqa: E402
from tools.tswgen.run_emu3 import run_case  # noqa: E402

CASES = [
    # B, H, W, n_wg, norm, sparse, hin, zero_patch
    (1, 12, 256, 1, 0, False, False, False),
    (2, 17, 304, 5, 0, True, False, True),    # two bands, shares that start / end mid-image, NaN patch, negative sparse values
    (1, 20, 512, 2, 1, True, False, False),
    (1, 14, 256, 2, 2, False, True, False),   # pre-normalised gates, continuation pass (H_t0 != H_0)
    (2, 40, 304, 2, 0, False, False, True),   # streams long enough for every slot of the pool to be reused several times
    (3, 9, 256, 2, 2, True, False, False),    # images shorter than the ring: several separator rows resident at once
]


@pytest.mark.parametrize("B,H,W,n_wg,norm,sparse,hin,zp", CASES)
def test_emulated_round3_loop_vs_oracle(B, H, W, n_wg, norm, sparse, hin, zp):
    os.chdir(ROOT)
    err, nanmis, out, ref = run_case(B, H, W, n_wg, norm, sparse, hin, seed=B + H + W, zero_patch=zp, verbose=False)
    assert nanmis == 0
    assert err <= 1e-4, err
    if zp:
        assert np.isnan(ref).any()


def test_emulator_flags_lds_dma_misuse():
    """what the emulator must catch about LDS-DMA: data read before the issuing wave's vmcnt wait, data read by another wave
    without a barrier behind that wait, a landing zone someone else touched in the same barrier epoch"""
    from tools.tswgen.isa import Prog, V, S, M0
    from tools.tswgen.emu import Emu, EmuError

    def prog(wait, barrier):
        p = Prog()
        p.emit("v_lshlrev_b32", V(1), [4, V(0)])
        p.emit("s_cmp_eq_u32", (), [S(31), 0])
        p.emit("s_cbranch_scc0", (), [".Lother"])
        p.emit("s_mov_b32", M0, [1024])
        p.emit("s_nop", (), [0])
        p.emit("global_load_lds_dwordx4", (), [V(1), S(2, 2), M0])
        if wait:
            p.waitcnt(vm=0)
        p.label(".Lother")
        if barrier:
            p.emit("s_barrier")
        p.emit("v_add_u32", V(2), [1024, V(1)])
        p.emit("ds_read_b32", V(3), [V(2)])
        p.waitcnt(lgkm=0)
        if not barrier:
            p.emit("s_barrier")
        p.waitcnt(vm=0)
        return p

    def run(p):
        mem = np.zeros(1 << 16, np.uint8)
        mem.view(np.float32)[:] = np.arange(mem.size // 4)
        e = Emu(p, mem, 8192, nwaves=2)
        for w in e.waves:
            w.v[0] = np.arange(64, dtype=np.uint32)
            w.s[31], w.s[2], w.s[3] = w.wid, 8192, 0
        e.run()
        return e
    e = run(prog(True, True))
    assert np.array_equal(e.waves[1].v[3].view(np.float32), 2048 + 4 * np.arange(64))   # dword 0 of every lane's 16 bytes
    with pytest.raises(EmuError, match="LDS-DMA load in flight"):
        run(prog(False, True))
    with pytest.raises(EmuError, match="race"):
        run(prog(True, False))


def test_generated_include_is_current_and_hazard_free():
    inc = open(os.path.join(ROOT, "cspn_amd", "csrc", "cspn2d_tsw3_gen.inc")).read()
    assert "#define TSW3_LDS_BYTES %d\n" % K.LDS_BYTES in inc and "#define TSW3_TAB_MAX_ROWS %d\n" % K.TAB_MAX_ROWS in inc
    for norm, sparse, hin in ((0, 0, 0), (1, 1, 0), (2, 0, 1)):
        p = K.build(dict(norm=norm, sparse=bool(sparse), hin=bool(hin)))
        assert not check_hazards(p)
        assert ("#define TSW3_ASM_%d_%d_%d R\"ASM(\n%s\n)ASM\"" % (norm, sparse, hin, p.text())) in inc


def test_dma_schedule_is_consistent():
    """the static LDS-DMA schedule: every row of every group requested exactly once, by a wave that is not in an event step,
    at least three steps before the cooking tasks read it, not before its slot's previous row was injected; a wave waits for a
    row before it requests the next one"""
    LV, NW = K.LV, K.NW
    seen = {}
    for w in range(NW):
        for c, (jj, dg) in K.DMA_ISSUE.items():
            assert c >= 4                                    # counters 0..3 carry the wave's events
            for tau in range(0, 48):                         # tau: step number; the wave's counter is (tau - 3 w) mod 24
                if (tau - 3 * w) % LV != c:
                    continue
                g = tau // 3 + dg
                seen.setdefault((g, jj), []).append((w, tau))
                read_step = 3 * g - 2
                assert read_step - tau >= 4                  # landed (wait at the top of step 3g - 3) before the barrier of 3g - 3
                prev_event = 3 * (g - 3) + (0, 0, 1, 2)[jj]  # row 12 earlier in the stream = previous tenant of the slot
                assert tau > prev_event
                waits = [t for t in range(tau + 1, tau + 8) if (t - 3 * w) % LV in K.DMA_WAIT]
                assert waits and waits[0] == 3 * g - 3       # this wave's next wait (vmcnt 0: all it has in flight) is the row's deadline
    full = [k for k in seen if 3 <= k[0] <= 12]
    assert len(full) == 40 and all(len(seen[k]) == 1 for k in full)
    # at most two waves request in any step, and never two on one SIMD (waves w and w + 4 share one)
    for tau in range(24, 48):
        ws = [w for w in range(NW) if (tau - 3 * w) % LV in K.DMA_ISSUE]
        assert len(ws) <= 2 and len({w % 4 for w in ws}) == len(ws)


PLAN_CASES = [(1, 7, 256, 256, False), (3, 33, 304, 256, False), (2, 100, 1216, 256, False), (64, 304, 1216, 256, True),
              (16, 228, 304, 256, True), (5, 19, 516, 64, False), (1, 1, 260, 256, False), (40, 50, 772, 256, True)]


@pytest.mark.parametrize("B,H,W,max_wg,xcd", PLAN_CASES)
def test_compact_plan_owns_every_pixel_once(B, H, W, max_wg, xcd):
    n_iter = 24
    nb = len(plan_bands(W, n_iter))
    n_wg, stride = plan_geo(B, H, W, n_iter, max_wg)
    xcd_geo = None
    if xcd and n_wg == (max_wg // nb) * nb and (max_wg // 8) // nb >= 1:
        per_xcd = max_wg // 8
        gpx = per_xcd // nb
        xcd_geo = (gpx, (8 * (per_xcd - gpx * nb)) // nb, per_xcd)
        n_wg = 8 * per_xcd
    hdr, geom, tab = build_plan(B, H, W, n_iter, n_wg, xcd_geo)
    assert tab.shape[1] <= K.TAB_MAX_ROWS
    yb = ybits_of(H)
    owned = np.zeros((B, H, W), np.int32)
    for g in range(n_wg):
        Q = int(hdr[g, 0])
        lo, hi, p0 = int(hdr[g, 2]) & 0xffff, int(hdr[g, 2]) >> 16, int(hdr[g, 3]) // 4
        assert not tab[g, :K.PADF].any() and not tab[g, K.PADF + Q:].any()   # padding rows are inactive
        assert int(geom[g]) & 0xff == yb and bool(geom[g] >> K.G_FIRST & 1) == (p0 == 0) and bool(geom[g] >> K.G_LAST & 1) == (p0 + 256 == W)
        for d in tab[g, K.PADF:K.PADF + Q]:
            d = int(d)
            if not d & 1:
                assert d == 0   # separator
                continue
            y, b = (d >> 4) & ((1 << yb) - 1), d >> (4 + yb)
            assert b < B and y < H
            assert bool(d >> K.F_UP & 1) == (y + 1 < H) and bool(d >> K.F_DN & 1) == (y >= 1)
            if d >> K.F_OWNED & 1:
                owned[b, y, p0 + lo:p0 + hi] += 1
    assert owned.min() == 1 and owned.max() == 1
