#!/usr/bin/env python
"""tools/r06/time_off_fast_path.py -- shapes off the assembly rings' fast path, for the record: images narrower than one 256-column band (the compiler-generated ring
kernel cspn2d_fused_kernel), widths that are not a multiple of 4 (rows padded in the workspace), against one launch per iteration.  One JSON line per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cspn_amd  # noqa: E402
from tools.r06.bench_head import timeit  # noqa: E402

for (B, H, W, N) in [(64, 228, 152, 24), (64, 120, 160, 24), (16, 240, 252, 24), (64, 304, 1218, 24), (64, 304, 1216, 24), (16, 228, 302, 12)]:
    gen = torch.Generator(device="cuda").manual_seed(3)
    g = torch.randn(B, 8, H, W, generator=gen, device="cuda")
    h = torch.rand(B, 1, H, W, generator=gen, device="cuda") * 10
    row = {"shape": [B, H, W], "n_iter": N, "algo": int(cspn_amd.load().cspn2d_auto_algo(B, H, W, N))}
    a = cspn_amd.cspn2d_forward(g, h, None, N, "8sum")
    b = cspn_amd.cspn2d_forward(g, h, None, N, "8sum", "stepwise")
    row["max_rel_diff_vs_stepwise"] = float((a - b).abs().max() / b.abs().max())
    for name, algo in (("auto_ms", "auto"), ("stepwise_ms", "stepwise")):
        for r in range(2):
            avg, mn = timeit(lambda: cspn_amd.cspn2d_forward(g, h, None, N, "8sum", algo), reps=20, warm=5)
        row[name] = round(avg, 4)
    alg = B * H * W * 40
    row["frac_of_8TBps_at_40B_per_px"] = round(alg / (row["auto_ms"] * 1e-3) / 8e12, 4)
    print(json.dumps(row), flush=True)
