#!/usr/bin/env python
"""tools/r05/soak_short_pass.py -- run-to-run flicker hunt for the short first pass of the assembly loop: every n_iter = 1 .. 23 (and 25 .. 47 step 3), three shapes,
with and without the mask, 40 repetitions each, every repetition bitwise equal to the first; between repetitions a different n runs (other LDS contents, other
phase of the clock).  One line."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cspn_amd  # noqa: E402


def main():
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(123)
    launches = 0
    for B, H, W in ((8, 304, 1216), (16, 228, 304), (3, 61, 516)):
        g = torch.randn(B, 8, H, W, generator=gen, device=dev)
        h = torch.rand(B, 1, H, W, generator=gen, device=dev) * 80
        s = (torch.rand(B, 1, H, W, generator=gen, device=dev) < 0.01).float() * (h + 0.1)
        ns = list(range(1, 24)) + list(range(25, 48, 3))
        for sp in (None, s):
            first = {n: cspn_amd.cspn2d_forward(g, h, sp, n, "8sum", "fused") for n in ns}
            for rep in range(40):
                for n in ns:
                    o = cspn_amd.cspn2d_forward(g, h, sp, n, "8sum", "fused")
                    launches += 1
                    if not torch.equal(o, first[n]):
                        print("FLICKER: shape %s n_iter %d rep %d sparse %s" % ((B, H, W), n, rep, sp is not None))
                        sys.exit(1)
    torch.cuda.synchronize()
    print("SOAK OK: %d forwards of 1 .. 47 iterations, every repetition bitwise equal to its first run" % launches)


if __name__ == "__main__":
    main()
