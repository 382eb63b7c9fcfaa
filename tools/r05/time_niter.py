#!/usr/bin/env python
"""tools/r05/time_niter.py -- device time of cspn2d_forward_f32 at KITTI 304x1216 x 64 for a range of iteration counts: the assembly loop (round 5:
a short first pass of n_iter % 24 iterations + full passes) against the compiler-generated ring kernel (algo fused_cxx).  One JSON line per count."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cspn_amd  # noqa: E402


def main():
    B, H, W = 64, 304, 1216
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(11)
    g = torch.randn(B, 8, H, W, generator=gen, device=dev)
    h = torch.rand(B, 1, H, W, generator=gen, device=dev) * 80
    st = torch.cuda.current_stream()
    for n in [int(x) for x in (sys.argv[1:] or "1 4 8 12 16 20 23 24 25 30 36 47 48".split())]:
        row = {"n_iter": n}
        outs = {}
        for algo in ("fused", "fused_cxx"):
            for _ in range(5):
                outs[algo] = cspn_amd.cspn2d_forward(g, h, None, n, "8sum", algo)
            torch.cuda.synchronize()
            evs = []
            for _ in range(30):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); cspn_amd.cspn2d_forward(g, h, None, n, "8sum", algo); e1.record(st)
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in evs)
            row[algo + "_ms"] = round(sum(ms) / len(ms), 4)
        d = (outs["fused"] - outs["fused_cxx"]).abs().max().item() / outs["fused_cxx"].abs().max().item()
        row["max_rel_diff"] = d
        row["frac_of_8TBps_at_40B_per_px"] = round(B * H * W * 40 / (row["fused_ms"] * 1e-3) / 8e12, 4)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
