#!/usr/bin/env python
"""tools/r05/ab_vol3d_placement.py -- A/B on one box: the persistent 3D kernel with the XCD-aware tile placement (blocks of the tile grid
on workgroup ids that share an XCD, rows with same-XCD readers stored L2-resident) against the same kernel with the tiles in plain
workgroup order (every neighbour on another XCD: all rows write-through), BASELINE config 5.  Through the hook library (the plain order
is not reachable through the ABI).  One JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cspn_amd  # noqa: E402
from cspn_amd import _lib  # noqa: E402


def main():
    B, D, H, W, N = 4, 32, 160, 608, 12
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(5000)
    g = torch.rand(B, 26, D, H, W, generator=gen, device=dev); g /= g.sum(1, keepdim=True)
    h = torch.rand(B, 1, D, H, W, generator=gen, device=dev)
    hooks, lib = _lib.load_hooks(), cspn_amd.load()
    ws = torch.empty(lib.cspn3d_workspace_bytes_ex(B, D, H, W, N, 2, 0), dtype=torch.uint8, device=dev)
    outs = {0: torch.empty_like(h), 2: torch.empty_like(h)}
    st = torch.cuda.current_stream()

    def run(mode):
        rc = hooks.cspn_debug_3d_persistent_forward(g.data_ptr(), h.data_ptr(), outs[mode].data_ptr(), B, D, H, W, N, -1, mode, ws.data_ptr(), st.cuda_stream)
        assert rc == 0

    for _ in range(20):
        run(0); run(2)
    torch.cuda.synchronize()
    res = {}
    for rnd in range(3):
        for mode, name in ((0, "xcd_placement"), (2, "plain_order")):
            evs = []
            for _ in range(30):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); run(mode); e1.record(st)
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in evs)
            res.setdefault(name, []).append({"avg_ms": round(sum(ms) / len(ms), 4), "min_ms": round(ms[0], 4)})
    cspn_amd.cspn3d_check_status()
    vox = B * D * H * W
    out = {"workload": "config 5: 4 x 32x160x608, 12 steps, persistent kernel", "bit_identical": bool(torch.equal(outs[0], outs[2])), "runs": res}
    for name in res:
        best = min(r["avg_ms"] for r in res[name])
        out[name + "_ms"] = best
        out[name + "_frac"] = round(vox * 112 / (best * 1e-3) / 8e12, 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
