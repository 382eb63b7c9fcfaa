"""The round-6 loop (tools/tswgen/kernel4.py -> cspn_amd/csrc/cspn2d_tsw4_gen.inc: a ring of 12 waves x 3 rows at 168 VGPRs, LDS-DMA row slots,
one wave per row computing scale / c', events that read the raw planes) executed instruction by instruction in the CPU emulator against the oracle
(reference cspn_pytorch/models/cspn.py:42-172): register allocation (36 working registers, heavy aliasing), schedule, waitcnt / vmcnt placement, LDS races
incl. LDS-DMA in flight, addresses.  Also: the committed include is what the generator emits, and the static rules hold."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.tswgen import kernel4 as K  # noqa: E402
from tools.tswgen.isa import check_hazards  # noqa: E402
from tools.tswgen.run_emu4 import run_case  # noqa: E402

CASES = [
    # B, H, W, n_wg, norm, sparse, zero_patch
    (1, 12, 256, 1, 0, False, False),
    (2, 17, 304, 5, 0, True, True),     # two bands (first / last band edge columns), shares that start / end mid-image, NaN patch, negative sparse
    (1, 20, 512, 2, 1, True, False),    # '8sum_abs': |G| through the source modifier of the event's multiply
    (1, 14, 256, 2, 2, True, False),    # gates used as given: no sums, scale = 1 - m only
    (2, 17, 304, 5, 3, True, True),     # prenorm: the kernel reads the reference's gate_wb, the oracle the raw tensors
]


@pytest.mark.parametrize("B,H,W,n_wg,norm,sparse,zp", CASES)
def test_emulated_ring12x3_vs_oracle(B, H, W, n_wg, norm, sparse, zp):
    os.chdir(ROOT)
    err, nanmis, out, ref = run_case(B, H, W, n_wg, norm, sparse, seed=B + H + W, zero_patch=zp, verbose=False)
    assert nanmis == 0
    assert err <= 1e-5, err
    if zp:
        assert np.isnan(ref).any()


@pytest.mark.parametrize("B,H,W,ncu,norm,sparse", [(2, 21, 304, 3, 0, True), (1, 150, 516, 4, 1, False)])
def test_emulated_ring12x3_on_linear_plan_pieces_that_change_band(B, H, W, ncu, norm, sparse):
    """the forward passes' linear plan: a workgroup's piece may end one band's rows and continue with the next band's (the retirement re-derives
    the owned-lane mask from the descriptor dword it kept in an SGPR since the row's injection)"""
    from tools.tswgen.plan import LinearPlan
    from tools.tswgen.plan4 import _consts
    with _consts():
        lp = LinearPlan(B, H, W, 24, ncu, xcd=False)
    assert any(len({r[0] for r in lp.runs(lp.cut[p], lp.cut[p + 1])}) > 1 for p in range(lp.n_wg)), "no piece changes band: pick another case"
    os.chdir(ROOT)
    err, nanmis, out, ref = run_case(B, H, W, 0, norm, sparse, seed=ncu, zero_patch=True, verbose=False, linear=ncu)
    assert nanmis == 0
    assert err <= 1e-5, err


def test_register_budget_roles_and_pipeline_constants():
    """what the design rests on: 168 VGPRs (three waves per SIMD), 9 slots x 11 KiB + 48 KiB of boundary rows + the table = 160 KiB, 36 = 0 (mod 9) (static slot
    addressing), one heavy job per SIMD and step, no DMA request whose wait would fall into an event step"""
    p = K.build(dict(norm=0, sparse=True))
    assert K.vgprs_used(p) == 168
    assert not check_hazards(p)
    assert K.LDS_BYTES == 163840 and K.LDS_TAB == 2 * 12 * 2048 + 9 * 11264 and (K.NW * K.NSLOT) % K.NR == 0
    g = K.Gen(dict(norm=0))
    for parity in (0, 1):
        simd_jobs = {}
        for c in range(parity, 24, 2):
            jobs = int(c < 3) + int(g.roles[c]["cook"] is not None) + int(g.roles[c]["dma"] is not None)
            simd_jobs.setdefault(c % 8, 0)
            simd_jobs[c % 8] += jobs
        assert sorted(simd_jobs.values()) == ([1, 1, 1, 1] if parity == 0 else [1, 1, 1, 2]), simd_jobs   # odd steps: two rows to request, on one SIMD
    for c in range(24):
        if g.roles[c]["dma"] is not None:
            assert (c + 3) % 24 > 2 and c > 2                 # waited for three steps later: never in an event step (their stores share vmcnt)
        if g.roles[c]["cook"] is not None:
            assert c > 3                                      # steps 1 .. 3 finish the late planes of the events with SQX / SQY
    # rows per step, pipeline: requested at phi - 5, waited for at the end of phi - 2, scale / c' at phi - 1, event at phi
    assert K.last_step(1) == 24 and K.last_step(4) == 26 and K.last_step(481) == 2 * 160 + 0 + 24


def test_generated_include_is_current(tmp_path, monkeypatch):
    """the committed cspn2d_tsw4_gen.inc is byte for byte what the generator emits (8 variants: 4 norms x mask)"""
    from tools.tswgen import emit4
    out = tmp_path / "gen4.inc"
    monkeypatch.setattr(sys, "argv", ["emit4", str(out)])
    monkeypatch.delenv("TSW_CFG", raising=False)
    emit4.main()
    new = out.read_text()
    old = open(os.path.join(ROOT, "cspn_amd", "csrc", "cspn2d_tsw4_gen.inc")).read()
    assert new.count("#define TSW4_ASM_") == 8
    assert new == old, "cspn_amd/csrc/cspn2d_tsw4_gen.inc is stale: python -m tools.tswgen.emit4"
