#!/usr/bin/env python
"""tools/r05/stress_vol3d.py -- repeated launches of the persistent 3D kernel (XCD-aware placement, L2-resident rows) must be bit-identical run to run and
equal to the per-step kernel, also while another stream keeps the fabric and the L2s busy (a stale L2-resident quad, a lost tag or a misplaced workgroup would
show as a differing voxel, NaN or CSPN_E_ASYNC).  Prints one line per shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cspn_amd  # noqa: E402


def run(B, D, H, W, N, reps, C=1):
    gen = torch.Generator(device="cuda").manual_seed(B * 7 + D + W)
    g = torch.rand(B, 26, D, H, W, generator=gen, device="cuda"); g /= g.sum(1, keepdim=True)
    x = torch.rand(B, C, D, H, W, generator=gen, device="cuda")
    ref = torch.cat([cspn_amd.cspn3d_forward(g, x[:, c:c + 1].contiguous(), None, N, "none", algo="stepwise") for c in range(C)], 1)
    noise_a = torch.empty(64 << 20, dtype=torch.float32, device="cuda").normal_()
    noise_b = torch.empty_like(noise_a)
    side = torch.cuda.Stream()
    bad = 0
    for r in range(reps):
        if r % 2 == 1:   # every other launch runs beside a 256 MB copy + an elementwise kernel on another stream
            with torch.cuda.stream(side):
                noise_b.copy_(noise_a)
                noise_b.mul_(1.0001)
        out = cspn_amd.cspn3d_forward(g, x, None, N, "none", algo="persistent") if C == 1 else cspn_amd.cspn3d_forward_multi(g, x, N)
        if not torch.equal(out, ref):
            bad += 1
    torch.cuda.synchronize()
    cspn_amd.cspn3d_check_status()
    print("3D (%d, %d, %d, %d, %d) C=%d N=%d: %d launches, differing from the per-step kernel: %d" % (B, C, D, H, W, C, N, reps, bad))
    return bad


if __name__ == "__main__":
    reps = int(os.environ.get("STRESS_REPS", "200"))
    bad = run(4, 32, 160, 608, 12, reps) + run(1, 32, 160, 152, 6, reps) + run(2, 16, 64, 200, 5, reps) + run(8, 32, 160, 608, 4, reps // 4, C=3)
    print("STRESS %s" % ("OK" if bad == 0 else "FAILED: %d" % bad))
    sys.exit(1 if bad else 0)
