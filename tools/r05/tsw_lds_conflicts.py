#!/usr/bin/env python
"""tools/r05/tsw_lds_conflicts.py -- which LDS instructions of the generated 2D ring loop run into bank conflicts?
The hardware counters say a third of the loop's LDS-active cycles are conflict cycles (profiles/r05_pmc_sq.md, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
This runs the generated loop in the CPU emulator (tools/tswgen/emu.py: per-lane LDS addresses are exact) and prices every LDS access with the lane groups and
bank rule of MI355X_MICROARCH.md (LDS): conflict cycles = sum over lane groups of (most distinct addresses on one bank - 1).
    python -m tools.r05.tsw_lds_conflicts [norm] [RING_REC bytes] [1: ring stores as ds_write2_b32]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.tswgen import emu as E  # noqa: E402
from tools.tswgen import kernel as K  # noqa: E402

G128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128 = G128 + [[l + 32 for l in g] for g in G128]
G32x2 = [list(range(32)), list(range(32, 64))]
G16x4 = [list(range(16 * i, 16 * i + 16)) for i in range(4)]
G8x8 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]
RULE = {  # op, ndw -> (lane groups, banks)
    ("ds_read_b32", 1): (G32x2, 32), ("ds_read2_b32", 1): (G32x2, 32), ("ds_read2st64_b32", 1): (G32x2, 32), ("ds_read_b64", 2): (G32x2, 64),
    ("ds_read_b128", 4): (G128, 64), ("ds_write_b32", 1): (G32x2, 32), ("ds_write2_b32", 1): (G32x2, 32), ("ds_write2st64_b32", 1): (G32x2, 32),
    ("ds_write_b64", 2): (G16x4, 32), ("ds_write_b128", 4): (G8x8, 32),
}
stats = {}


def price(op, addr, ndw, lanes):
    groups, nb = RULE[(op, ndw)]
    base = extra = 0
    for g in groups:
        bank = {}
        for l in g:
            if not lanes[l]:
                continue
            a = int(addr[l]) // 4
            for w in range(ndw):
                bank.setdefault((a + w) % nb, set()).add(a + w)
        if bank:
            base += 1
            extra += max(len(v) for v in bank.values()) - 1
    return base, extra


orig = E.Emu.lds_access


def patched(self, w, addr, ndw, write, lanes, align):
    ins = self.ins[w.pc]
    b, x = price(ins.op, addr, ndw, lanes)
    key = (ins.op, ins.text().split(";")[0].strip()[:70])
    s = stats.setdefault(key, [0, 0, 0])
    s[0] += 1; s[1] += b; s[2] += x
    return orig(self, w, addr, ndw, write, lanes, align)


def main():
    norm = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    if len(sys.argv) > 3:
        K.RING_W2 = sys.argv[3] == "1"
    if len(sys.argv) > 2:
        K.RING_REC = int(sys.argv[2])
        K.RING_SLOT = 64 * K.RING_REC
        K.LDS_TAGS = K.LDS_RING + 8 * K.RING_SLOT
        K.configure(False)
    E.Emu.lds_access = patched
    from tools.tswgen.run_emu import run_case
    err, nanmis, _, _ = run_case(2, 60, 304, 3, norm, True, False, seed=3, verbose=False, linear=3)
    assert nanmis == 0 and err <= 1e-4, (err, nanmis)
    tot_b = sum(v[1] for v in stats.values()); tot_x = sum(v[2] for v in stats.values())
    byop = {}
    for (op, txt), (n, b, x) in stats.items():
        o = byop.setdefault(op, [0, 0, 0]); o[0] += n; o[1] += b; o[2] += x
    print("RING_REC %d bytes, norm %d: LDS cycles conflict-free %d, conflict cycles %d = %.1f %% of LDS-active cycles" % (K.RING_REC, norm, tot_b, tot_x, 100.0 * tot_x / (tot_b + tot_x)))
    for op, (n, b, x) in sorted(byop.items(), key=lambda kv: -kv[1][2]):
        print("  %-18s %8d accesses  base %8d  conflict %8d (%.2f per access)" % (op, n, b, x, x / max(n, 1)))
    print("  worst instructions:")
    for (op, txt), (n, b, x) in sorted(stats.items(), key=lambda kv: -kv[1][2])[:14]:
        print("    %7d x  +%5.2f per access  %s" % (n, x / n, txt))


if __name__ == "__main__":
    main()
