"""The generated gfx950 main loop (tools/tswgen -> cspn_amd/csrc/cspn2d_tsw_gen.inc) executed instruction by instruction in
the CPU emulator (tools/tswgen/emu.py) against the oracle: register allocation, schedule, waitcnt placement, LDS races,
addresses.  Also: the committed include is what the generator emits, and the static hazard rules hold."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.tswgen import kernel as K  # noqa: E402
from tools.tswgen.isa import check_hazards  # noqa: E402
from tools.tswgen.run_emu import fill_table, run_case  # noqa: E402

CASES = [
    # B, H, W, n_wg, norm, sparse, hin, zero_patch
    (1, 12, 256, 1, 0, False, False, False),
    (2, 17, 304, 5, 0, True, False, True),    # two bands, shares that start / end mid-image, NaN patch
    (1, 20, 512, 2, 1, True, False, False),
    (1, 14, 256, 2, 2, True, True, False),    # pre-normalised gates, continuation pass (H_t0 != H_0)
    # norm 3 = CSPN_NORM_PRENORM (round 5): the kernel reads the reference's gate_wb (numpy twin of cspn2d_normalize_f32
    # applied to the raw guidance), the oracle sees the RAW tensors with '8sum'
    (2, 17, 304, 5, 3, True, False, True),    # two bands, mid-image share ends, NaN weights from a zero-guidance patch, mask
    (2, 14, 256, 2, 3, False, True, False),   # continuation pass
]


@pytest.mark.parametrize("B,H,W,n_wg,norm,sparse,hin,zp", CASES)
def test_emulated_asm_loop_vs_oracle(B, H, W, n_wg, norm, sparse, hin, zp):
    os.chdir(ROOT)
    err, nanmis, out, ref = run_case(B, H, W, n_wg, norm, sparse, hin, seed=B + H + W, zero_patch=zp, verbose=False)
    assert nanmis == 0
    assert err <= 1e-4, err
    if zp:
        assert np.isnan(ref).any()


@pytest.mark.parametrize("B,H,W,ncu,norm,sparse", [(2, 21, 304, 3, 0, True), (1, 150, 516, 4, 1, False), (2, 60, 304, 3, 2, True),
                                                   (2, 30, 304, 3, 3, True)])
def test_emulated_asm_loop_on_linear_plan_pieces_that_change_band(B, H, W, ncu, norm, sparse):
    """round 4: the forward passes' linear plan -- a workgroup's piece may end one band's rows and continue with the next band's
    (the retirement re-derives the owned-lane mask from the row's descriptor): every pixel still comes out right"""
    from tools.tswgen.plan import LinearPlan
    lp = LinearPlan(B, H, W, 24, ncu, xcd=False)
    assert any(len({r[0] for r in lp.runs(lp.cut[p], lp.cut[p + 1])}) > 1 for p in range(lp.n_wg)), "no piece changes band: pick another case"
    os.chdir(ROOT)
    err, nanmis, out, ref = run_case(B, H, W, 0, norm, sparse, False, seed=ncu, zero_patch=(norm != 2), verbose=False, linear=ncu)
    assert nanmis == 0
    assert err <= 1e-4, err


@pytest.mark.parametrize("norm,sparse", [(0, True), (1, False), (2, True)])
def test_emulated_sited8_input_variant(norm, sparse):
    """cfg s8 (SURVEY 8f-2 experiment): the guidance arrives pre-sited and pair-interleaved, four aligned 16-byte loads per task"""
    os.chdir(ROOT)
    err, nanmis, _, ref = run_case(2, 15, 304, 4, norm, sparse, False, seed=11, zero_patch=(norm != 2), verbose=False, s8=True)   # (stub form of the inactive-row checks)
    assert nanmis == 0 and err <= 1e-4


@pytest.mark.parametrize("every", [1, K.HIST_EVERY])
def test_emulated_history_variant_writes_its_levels(every):
    """cfg hist (used by the backward pass): the levels every, 2 every .. < 24 of every owned pixel (the product keeps every
    fourth), checked against the oracle level by level"""
    os.chdir(ROOT)
    err, nanmis, _, _ = run_case(2, 11, 304, 4, 0, True, False, seed=3, verbose=False, hist=True, hist_every=every)
    assert nanmis == 0 and err <= 1e-4


@pytest.mark.parametrize("sparse", [False, True])
def test_emulated_history_variant_of_the_prenorm_contract(sparse):
    """cfg hist with norm 3 (round 6: the history sweep of the backward of CSPN_NORM_PRENORM): checkpoints and folded planes against the oracle"""
    os.chdir(ROOT)
    err, nanmis, _, _ = run_case(2, 11, 304, 4, 3, sparse, False, seed=3, verbose=False, hist=True, hist_every=K.HIST_EVERY, zero_patch=True)
    assert nanmis == 0 and err <= 1e-4


def test_emulated_adjoint_variant_vs_numpy_adjoint_recursion():
    """cfg adj: the backward's adjoint sweep A_t(p) = sum_k w'_k(p - off_k) A_{t+1}(p - off_k) run by the ring kernel over the
    folded planes (neighbour-sited, channel order reversed), every level checked"""
    from tools.tswgen.emu import Emu
    from tools.tswgen.plan import build_plan, plan_bands
    from oracle.backward import _shift, DY, DX
    B, H, W, n_wg = 2, 13, 304, 4
    rng = np.random.default_rng(1)
    total = B * H * W
    wp = (rng.standard_normal((8, B, H, W)) * 0.3).astype(np.float32)
    a = rng.standard_normal((B, H, W)).astype(np.float32)
    levels, cur = [], a.copy()
    for _ in range(24):
        nxt = np.zeros_like(cur)
        for k in range(8):
            nxt += _shift(wp[k] * cur, -DY[k], -DX[k])
        cur = nxt.astype(np.float32)
        levels.append(cur)
    every = K.HIST_EVERY
    npl = 24 // every - 1
    prog = K.build(dict(norm=2, adj=True, hist=True, hist_every=every))
    assert not check_hazards(prog)
    n_wg = -(-n_wg // len(plan_bands(W, 24))) * len(plan_bands(W, 24))
    hdr, tab = build_plan(B, H, W, 24, n_wg)
    bufs = {"gd": wp, "blur": a, "out": np.zeros(total, np.float32), "plan": tab, "hist": np.full(23 * total, np.nan, np.float32)}
    off, end = {}, 8192 + 65536   # the variant reads up to 4*W + 16 bytes in front of the planes
    for n, arr in bufs.items():
        off[n] = end
        end += (arr.nbytes + 4095) // 4096 * 4096 + 12288
    mem = np.zeros(end, np.uint8)
    mem.view(np.float32)[:] = np.nan
    for n, arr in bufs.items():
        mem[off[n]:off[n] + arr.nbytes] = arr.view(np.uint8).ravel()
    for wg in range(n_wg):
        if hdr[wg, 0] == 0:
            continue
        emu = Emu(prog, mem, K.LDS_BYTES)
        fill_table(emu, tab[wg])
        for w in emu.waves:
            w.v[0] = np.arange(64, dtype=np.uint32)
            for r, v in ((K.S_GD, off["gd"]), (K.S_BLUR, off["blur"]), (K.S_HIN, 4096), (K.S_SP, 4096), (K.S_OUT, off["out"]),
                         (K.S_PLAN, off["plan"] + wg * tab.shape[1] * 16), (K.S_HIST, off["hist"]), (K.S_HSTRIDE, total * 4)):
                w.s[r.i], w.s[r.i + 1] = v & 0xffffffff, v >> 32
            w.s[K.S_W4.i], w.s[K.S_HW4.i], w.s[K.S_LAST.i], w.s[K.S_WV.i] = 4 * W, 4 * total, int(hdr[wg, 1]), w.wid
            w.s[K.S_NROWS.i], w.s[K.S_LOHI.i] = tab.shape[1], int(hdr[wg, 2])
        emu.run()
    out = mem[off["out"]:off["out"] + total * 4].view(np.float32).reshape(B, H, W)
    hb = mem[off["hist"]:off["hist"] + npl * total * 4].view(np.float32).reshape(npl, B, H, W // 4, 4)[..., [0, 2, 3, 1]].reshape(npl, B, H, W)
    assert np.abs(out - levels[23]).max() <= 1e-5 * np.abs(levels[23]).max()
    for i in range(npl):   # plane i = level (i + 1) * every of the sweep = A_{24 - (i + 1) every}
        n = (i + 1) * every
        assert np.abs(hb[i] - levels[n - 1]).max() <= 1e-5 * np.abs(levels[n - 1]).max(), n
    assert np.isnan(mem[off["hist"] + npl * total * 4:off["hist"] + 23 * total * 4].view(np.float32)).all()   # nothing behind them


def test_generated_include_is_current_and_hazard_free(tmp_path, monkeypatch):
    """the committed cspn2d_tsw_gen.inc is byte for byte what the generator emits -- ALL 33 variants (16 forward: 4 norms incl. round 5's
    prenorm x mask x continuation pass, 8 short first passes (n < 24 iterations), 8 history (round 6: prenorm too), the adjoint sweep) -- and every one of them passes the static hazard rules (K.build
    raises on a hazard)"""
    from tools.tswgen import emit
    out = tmp_path / "gen.inc"
    monkeypatch.setattr(sys, "argv", ["emit", str(out)])
    emit.main()
    new = out.read_text()
    old = open(os.path.join(ROOT, "cspn_amd", "csrc", "cspn2d_tsw_gen.inc")).read()
    assert new.count("#define TSW_ASM_") == 33
    assert new == old, "cspn_amd/csrc/cspn2d_tsw_gen.inc is stale: python -m tools.tswgen.emit"


def test_scheduler_respects_hazards_and_emulator_flags_misuse():
    """the tooling the generated loop relies on: VALU->DPP distance, waitcnt tracking, LDS race detection"""
    from tools.tswgen.isa import Prog, V, S, schedule
    from tools.tswgen.emu import Emu, EmuError
    # 1. a DPP read right behind the VALU write of its source is separated by the scheduler (or s_nop) and accepted
    p = Prog()
    p.emit("v_mov_b32", V(1), [1.0])
    p.emit("v_mov_b32", V(2), V(1), dpp="wave_shr:1")
    p.emit("v_mov_b32", V(3), [2.0])
    assert check_hazards(p), "unscheduled: must be reported"
    schedule(p)
    assert not check_hazards(p)
    # 2. reading a register whose LDS load has not been waited for is an emulator error
    p = Prog()
    p.emit("v_mov_b32", V(1), [0])
    p.emit("ds_read_b32", V(2), [V(1)])
    p.emit("v_mov_b32", V(3), V(2))
    with pytest.raises(EmuError, match="outstanding"):
        Emu(p, np.zeros(8192, np.uint8), 1024, nwaves=1).run()
    # 3. two waves touching the same LDS dword inside one barrier epoch is a race; across a barrier it is not
    for with_barrier in (False, True):
        p = Prog()
        p.emit("v_mov_b32", V(1), [0])
        p.emit("s_cmp_eq_u32", (), [S(31), 0])
        p.emit("s_cbranch_scc0", (), [".Lreader"])
        p.emit("ds_write_b32", (), [V(1), V(1)])
        p.waitcnt(lgkm=0)
        p.emit("s_barrier")
        p.emit("s_branch", (), [".Lend"])
        p.label(".Lreader")
        if with_barrier:
            p.emit("s_barrier")
        p.emit("ds_read_b32", V(2), [V(1)])
        p.waitcnt(lgkm=0)
        if not with_barrier:
            p.emit("s_barrier")
        p.label(".Lend")
        e = Emu(p, np.zeros(8192, np.uint8), 1024, nwaves=2)
        for w in e.waves:
            w.s[31] = w.wid
        if with_barrier:
            e.run()
        else:
            with pytest.raises(EmuError, match="race"):
                e.run()


@pytest.mark.parametrize("kw", [
    dict(B=3, H=30, W=304, n_wg=4, norm=0, sparse=True, early_n=7),                       # slots 2 / 3 deliver (saved copy / LDS boundary row)
    dict(B=2, H=37, W=260, n_wg=3, norm=1, sparse=False, early_n=22, zero_patch=True),    # the stored level wraps into the next ring cycle; NaN patch
    dict(B=2, H=30, W=304, n_wg=2, norm=3, sparse=False, early_n=13, linear=3),           # linear plan: pieces that change band (mask re-derived in the stub)
    dict(B=1, H=5, W=256, n_wg=1, norm=2, sparse=True, early_n=1),                        # one iteration, fewer rows than a ring cycle
])
def test_emulated_short_pass_vs_oracle(kw):
    """generator cfg `early` (round 5): a first pass of n < 24 iterations -- the ring runs its 24 levels, a row is stored at the end of the
    step in which it completed level n (out-of-line stub behind one scalar test per step), nothing at its retirement"""
    os.chdir(ROOT)
    err, nanmis, out, ref = run_case(verbose=False, **kw)
    assert nanmis == 0 and err <= 1e-5, kw
    assert not kw.get("zero_patch") or np.isnan(ref).any()


@pytest.mark.parametrize("sched_seed", [None, 7])
def test_emulated_elastic_loop_vs_oracle(sched_seed):
    """cfg elastic (no barrier in the loop: boundary rows and ring slots are validated by tags, profiles/r02_perf_notes.md)
    under the emulator's most skewed schedule and under a random one with stalls"""
    os.chdir(ROOT)
    err, nanmis, out, ref = run_case(2, 17, 304, 5, 0, True, False, seed=5, zero_patch=True, verbose=False, elastic=True,
                                     sched_seed=sched_seed)
    assert nanmis == 0 and err <= 1e-4 and np.isnan(ref).any()


def test_emulated_staggered_cooking_option():
    """generator option stagger (round 4: the two waves of a SIMD cook in different steps; two flavours of the loop): measured no
    faster and not in the product (profiles/r04_stagger_ab.md), but kept working"""
    os.chdir(ROOT)
    err, nanmis, _, ref = run_case(2, 17, 304, 5, 0, True, False, seed=9, zero_patch=True, verbose=False, cfg_extra=dict(stagger=True))
    assert nanmis == 0 and err <= 1e-4 and np.isnan(ref).any()


def test_emulated_ring_with_odd_record_stride_and_write2_stores(monkeypatch):
    """round 5 experiment switch (TSW_RING_REC / TSW_RING_W2, profiles/r05_lds_conflicts_and_sq.md): the cooked-row ring at 41 dwords per record with
    its 8-byte stores issued as ds_write2_b32 -- conflict-free event reads; measured +-0 on the GPU, kept as a generator option: it must stay correct"""
    os.chdir(ROOT)
    monkeypatch.setattr(K, "RING_REC", 164)
    monkeypatch.setattr(K, "RING_SLOT", 64 * 164)
    monkeypatch.setattr(K, "LDS_TAGS", K.LDS_RING + 8 * 64 * 164)
    monkeypatch.setattr(K, "RING_W2", True)
    try:
        err, nanmis, _, ref = run_case(2, 17, 304, 5, 0, True, False, seed=5, zero_patch=True, verbose=False)
        assert nanmis == 0 and err <= 1e-4 and np.isnan(ref).any()
    finally:
        monkeypatch.undo()
        K.configure(False)
