#!/usr/bin/env python
"""tools/stress_asm.py -- repeated launches of the assembly paths must be bit-identical run to run (an LDS race or a missing
wait would show up as flicker) and must match the compiler-generated kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cspn_amd  # noqa: E402


def main():
    torch.manual_seed(0)
    bad = 0
    from tools.fuzz_parity import forward2d_plan
    for (B, H, W, sp, loop) in [(8, 304, 1216, True, 0), (64, 304, 1216, False, 0), (64, 304, 1216, True, 2), (16, 228, 304, True, 0),
                                (3, 57, 260, False, 0), (16, 228, 304, True, 1), (7, 150, 516, True, 0)]:
        # loop = plan of the assembly passes: 0 the product's linear plan, 1 without XCD-aware placement, 2 band groups
        gen = torch.Generator(device="cuda").manual_seed(B + W)
        g = torch.randn(B, 8, H, W, generator=gen, device="cuda")
        h = torch.rand(B, 1, H, W, generator=gen, device="cuda") * 80
        s = (torch.rand(B, 1, H, W, generator=gen, device="cuda") < 0.01).float() * (h + 0.1) if sp else None
        go = torch.randn(B, 1, H, W, generator=gen, device="cuda")
        ref = forward2d_plan(g, h, s, 24, "8sum", loop)
        cxx = cspn_amd.cspn2d_forward(g, h, s, 24, "8sum", "fused_cxx")
        d = float((ref - cxx).abs().max() / cxx.abs().max())
        gg0, gh0 = cspn_amd.cspn2d_backward(g, h, s, go, 24, "8sum")
        flick = 0
        for i in range(40):
            o = forward2d_plan(g, h, s, 24, "8sum", loop)
            flick += int(not torch.equal(o, ref))
            if i % 8 == 0:
                gg, gh = cspn_amd.cspn2d_backward(g, h, s, go, 24, "8sum")
                flick += int(not torch.equal(gg, gg0)) + int(not torch.equal(gh, gh0))
        torch.cuda.synchronize()
        print("B%d %dx%d sparse=%s plan mode %d: asm vs compiled %.2e, non-identical repeats %d" % (B, H, W, sp, loop, d, flick), flush=True)
        bad += flick + int(d > 1e-5)
    print("STRESS", "OK" if bad == 0 else "FAILED")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
