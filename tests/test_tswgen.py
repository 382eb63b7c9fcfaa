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
from tools.tswgen.run_emu import run_case  # noqa: E402

CASES = [
    # B, H, W, n_wg, norm, sparse, hin, zero_patch
    (1, 12, 256, 1, 0, False, False, False),
    (2, 17, 304, 5, 0, True, False, True),    # two bands, shares that start / end mid-image, NaN patch
    (1, 20, 512, 2, 1, True, False, False),
    (1, 14, 256, 2, 2, True, True, False),    # pre-normalised gates, continuation pass (H_t0 != H_0)
]


@pytest.mark.parametrize("B,H,W,n_wg,norm,sparse,hin,zp", CASES)
def test_emulated_asm_loop_vs_oracle(B, H, W, n_wg, norm, sparse, hin, zp):
    os.chdir(ROOT)
    err, nanmis, out, ref = run_case(B, H, W, n_wg, norm, sparse, hin, seed=B + H + W, zero_patch=zp, verbose=False)
    assert nanmis == 0
    assert err <= 1e-4, err
    if zp:
        assert np.isnan(ref).any()


def test_generated_include_is_current_and_hazard_free():
    inc = open(os.path.join(ROOT, "cspn_amd", "csrc", "cspn2d_tsw_gen.inc")).read()
    for norm, sparse, hin in ((0, 0, 0), (1, 1, 1), (2, 1, 0)):
        p = K.build(dict(norm=norm, sparse=bool(sparse), hin=bool(hin)))
        assert not check_hazards(p)
        assert ("#define TSW_ASM_%d_%d_%d R\"ASM(\n%s\n)ASM\"" % (norm, sparse, hin, p.text())) in inc
