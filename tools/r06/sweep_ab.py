#!/usr/bin/env python
"""tools/r06/sweep_ab.py -- device time of the fused forward (24 iterations) on the 12 x 3 ring (mode 0) and on the 8 x 4 ring (mode 8) over batch sizes and
shapes: where does the new loop pay?  One JSON line per shape."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.fuzz_parity import forward2d_plan  # noqa: E402


def time_mode(g, h, s, mode, reps):
    st = torch.cuda.current_stream()
    for _ in range(15):
        forward2d_plan(g, h, s, 24, "8sum", mode)
    torch.cuda.synchronize()
    ev = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        forward2d_plan(g, h, s, 24, "8sum", mode)
        e1.record(st)
        ev.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ms) / len(ms)


def main():
    gen = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(B, 304, 1216, False) for B in (4, 8, 12, 16, 24, 32, 48, 64, 80, 96)] + [(B, 304, 1216, True) for B in (16, 32, 64)] + \
             [(B, 228, 304, False) for B in (16, 64, 256)] + [(16, 480, 640, False), (64, 480, 640, False), (8, 1024, 2048, False)]
    for (B, H, W, sparse) in shapes:
        g = torch.randn(B, 8, H, W, generator=gen, device="cuda")
        h = torch.rand(B, 1, H, W, generator=gen, device="cuda") * 80
        s = None
        if sparse:
            s = (torch.rand(B, 1, H, W, generator=gen, device="cuda") < 500.0 / (H * W)).float() * 30
        row = {"shape": [B, H, W], "sparse": sparse}
        a = [time_mode(g, h, s, 0, 30), time_mode(g, h, s, 8, 30), time_mode(g, h, s, 0, 30), time_mode(g, h, s, 8, 30)]
        row["new_ms"] = round((a[0] + a[2]) / 2, 4)
        row["old_ms"] = round((a[1] + a[3]) / 2, 4)
        row["new_over_old"] = round(row["new_ms"] / row["old_ms"], 4)
        print(json.dumps(row), flush=True)
        del g, h, s


if __name__ == "__main__":
    main()
