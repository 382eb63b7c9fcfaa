#!/usr/bin/env python
"""tools/r06/soak_head.py -- bitwise repeatability of the guidance head kernels (a counted vmcnt wait that is one too weak would show as run-to-run differences):
forward raw / gate_wb, gradient, 100 launches each against the first, at [64,64,152,608] and at a narrowed odd shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cspn_amd.train_utils import guidance_heads, guidance_heads_backward  # noqa: E402

bad = 0
for (B, C, h, w, oh, ow) in [(64, 64, 152, 608, 0, 0), (7, 37, 45, 211, 89, 421)]:
    gen = torch.Generator(device="cuda").manual_seed(B)
    x = torch.randn(B, C, h, w, generator=gen, device="cuda")
    w6 = torch.randn(8, C, 3, 3, generator=gen, device="cuda") / 24
    w5 = torch.randn(1, C, 3, 3, generator=gen, device="cuda") / 24
    H, W = (oh, ow) if oh else (2 * h, 2 * w)
    gg = torch.randn(B, 8, H, W, generator=gen, device="cuda")
    gb = torch.randn(B, 1, H, W, generator=gen, device="cuda")
    for name, fn in (("forward raw", lambda: guidance_heads(x, w6, w5, oh, ow)), ("forward gate_wb", lambda: guidance_heads(x, w6, w5, oh, ow, norm_type="8sum")),
                     ("gradient", lambda: guidance_heads_backward(x, w6, w5, gg, gb))):
        first = fn()
        diff = 0
        for _ in range(100):
            o = fn()
            diff += int(not all(torch.equal(a.nan_to_num(), b.nan_to_num()) for a, b in zip(o, first)))
        print("[%d,%d,%d,%d] -> %dx%d %s: %d of 100 launches differ from the first" % (B, C, h, w, H, W, name, diff), flush=True)
        bad += diff
print("SOAK OK" if bad == 0 else "SOAK FAILED")
