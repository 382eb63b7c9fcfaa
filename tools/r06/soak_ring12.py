#!/usr/bin/env python
"""tools/r06/soak_ring12.py -- bitwise repeatability of the 12 x 3 ring (an LDS race or a DMA landing late would show as run-to-run differences): every norm with and
without a mask at KITTI x 64 / x 96, 150 launches each against the first one; and against one launch per iteration (<= 1e-5)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cspn_amd  # noqa: E402
from cspn_amd import _lib  # noqa: E402

hooks = _lib.load_hooks()
bad = 0
for B, norm, sp in [(64, "8sum", False), (64, "8sum_abs", False), (64, "none", False), (96, "8sum", True), (96, "8sum_abs", True), (64, "prenorm", False), (96, "prenorm", True)]:
    H, W = 304, 1216
    gen = torch.Generator(device="cuda").manual_seed(B + len(norm))
    g = torch.randn(B, 8, H, W, generator=gen, device="cuda")
    if norm == "none":
        g = g.abs() / (g.abs().sum(1, keepdim=True) + 0.2)
    h = torch.rand(B, 1, H, W, generator=gen, device="cuda") * 80
    s = (torch.rand(B, 1, H, W, generator=gen, device="cuda") < 0.002).float() * (h + 0.1) if sp else None
    gin = cspn_amd.cspn2d_normalize(g, "8sum") if norm == "prenorm" else g
    ring = int(hooks.cspn_debug_fused2d_ring(B, H, W, 1 if sp else 0))
    first = cspn_amd.cspn2d_forward(gin, h, s, 24, norm)
    ref = cspn_amd.cspn2d_forward(g, h, s, 24, "8sum" if norm == "prenorm" else norm, "stepwise")
    err = float((first - ref).abs().max() / ref.abs().max())
    diff = 0
    for _ in range(150):
        o = cspn_amd.cspn2d_forward(gin, h, s, 24, norm)
        diff += int(not torch.equal(o, first))
    print("B%d %s mask=%s ring %d: vs stepwise %.2e, %d of 150 launches differ from the first" % (B, norm, sp, ring, err, diff), flush=True)
    bad += diff + (err > 1e-5)
print("SOAK OK" if bad == 0 else "SOAK FAILED")
