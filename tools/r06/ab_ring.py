#!/usr/bin/env python
"""tools/r06/ab_ring.py -- the round-6 ring (12 waves x 3 rows, cspn2d_tsw4.hip) against the 8 x 4 ring of rounds 1-5 (plan_mode + 8) on one box:
parity against each other and against the CPU oracle (image 0 of each shape), then device time per launch (HIP events, alternating).
usage: python tools/r06/ab_ring.py [reps]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cspn_amd  # noqa: E402
from tools.fuzz_parity import forward2d_plan  # noqa: E402


def oracle_img0(g, h, s, norm):
    from oracle import oracle as O
    return O.cspn2d_oracle(g[:1].cpu().numpy(), h[:1].cpu().numpy(), None if s is None else s[:1].cpu().numpy(), 24, norm)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(7)
    shapes = [(64, 304, 1216, False, "8sum"), (32, 304, 1216, True, "8sum"), (16, 228, 304, False, "8sum"), (8, 304, 1216, False, "8sum_abs"),
              (3, 100, 260, True, "none"), (2, 57, 1000, True, "8sum")]
    st = torch.cuda.current_stream()
    for (B, H, W, sparse, norm) in shapes:
        g = torch.randn(B, 8, H, W, generator=gen, device=dev)
        if norm == "none":
            g = g.abs() / (g.abs().sum(1, keepdim=True) + 0.3)
        h = torch.rand(B, 1, H, W, generator=gen, device=dev) * 80
        s = None
        if sparse:
            m = (torch.rand(B, 1, H, W, generator=gen, device=dev) < 500.0 / (H * W)).float()
            s = m * (torch.rand(B, 1, H, W, generator=gen, device=dev) * 80 + 0.1)
            s[:, :, ::7, 5::31] *= -1
        new = forward2d_plan(g, h, s, 24, norm, 0)
        old = forward2d_plan(g, h, s, 24, norm, 8)
        torch.cuda.synchronize()
        ref = oracle_img0(g, h, s, norm)
        den = float(np.nanmax(np.abs(ref)))
        row = {"shape": [B, H, W], "sparse": sparse, "norm": norm,
               "new_vs_old": float((new - old).abs().max() / old.abs().max()),
               "new_vs_oracle_img0": float(np.nanmax(np.abs(new[:1].cpu().numpy() - ref)) / den),
               "old_vs_oracle_img0": float(np.nanmax(np.abs(old[:1].cpu().numpy() - ref)) / den)}
        for _ in range(10):
            forward2d_plan(g, h, s, 24, norm, 0)
            forward2d_plan(g, h, s, 24, norm, 8)
        torch.cuda.synchronize()
        t = {0: [], 8: []}
        for _ in range(reps):
            for mode in (0, 8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                forward2d_plan(g, h, s, 24, norm, mode)
                e1.record(st)
                t[mode].append((e0, e1))
        torch.cuda.synchronize()
        for mode, name in ((0, "new"), (8, "old")):
            ms = sorted(a.elapsed_time(b) for a, b in t[mode])
            row[name + "_ms_mean"] = round(sum(ms) / len(ms), 4)
            row[name + "_ms_min"] = round(ms[0], 4)
        bpp = 44 if sparse else 40
        row["new_frac"] = round(B * H * W * bpp / (row["new_ms_mean"] * 1e-3) / 8e12, 4)
        row["old_frac"] = round(B * H * W * bpp / (row["old_ms_mean"] * 1e-3) / 8e12, 4)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
