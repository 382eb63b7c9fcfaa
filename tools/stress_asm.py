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
    lib = cspn_amd.load()
    for (B, H, W, sp, loop) in [(8, 304, 1216, True, 3), (64, 304, 1216, False, 3), (64, 304, 1216, True, 2), (16, 228, 304, True, 3),
                                (3, 57, 260, False, 3), (16, 228, 304, True, 2)]:
        lib.cspn_debug_tsw_loop(loop)   # 2: round-2 loop, 3: round-3 loop (LDS-DMA: a missing wait / barrier would flicker)
        gen = torch.Generator(device="cuda").manual_seed(B + W)
        g = torch.randn(B, 8, H, W, generator=gen, device="cuda")
        h = torch.rand(B, 1, H, W, generator=gen, device="cuda") * 80
        s = (torch.rand(B, 1, H, W, generator=gen, device="cuda") < 0.01).float() * (h + 0.1) if sp else None
        go = torch.randn(B, 1, H, W, generator=gen, device="cuda")
        ref = cspn_amd.cspn2d_forward(g, h, s, 24, "8sum", "fused")
        cxx = cspn_amd.cspn2d_forward(g, h, s, 24, "8sum", "fused_cxx")
        d = float((ref - cxx).abs().max() / cxx.abs().max())
        gg0, gh0 = cspn_amd.cspn2d_backward(g, h, s, go, 24, "8sum")
        flick = 0
        for i in range(40):
            o = cspn_amd.cspn2d_forward(g, h, s, 24, "8sum", "fused")
            flick += int(not torch.equal(o, ref))
            if i % 8 == 0:
                gg, gh = cspn_amd.cspn2d_backward(g, h, s, go, 24, "8sum")
                flick += int(not torch.equal(gg, gg0)) + int(not torch.equal(gh, gh0))
        torch.cuda.synchronize()
        lib.cspn_debug_tsw_loop(0)
        print("B%d %dx%d sparse=%s loop %d: asm vs compiled %.2e, non-identical repeats %d" % (B, H, W, sp, loop, d, flick), flush=True)
        bad += flick + int(d > 1e-5)
    print("STRESS", "OK" if bad == 0 else "FAILED")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
