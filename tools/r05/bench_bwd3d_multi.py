#!/usr/bin/env python
"""tools/r05/bench_bwd3d_multi.py -- C channels on shared gates, training path (reference cspn_paddle/README.md:56, demo.py:65-75):
cspn3d_backward_multi_f32 (one level-keeping forward launch + one transposed launch of the persistent kernel for ALL channels, gate
planes written once) against C calls of cspn3d_backward_f32, at BASELINE config 5's volume.  One JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cspn_amd  # noqa: E402


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    B, C, D, H, W, N = 4, 3, 32, 160, 608, 12
    gen = torch.Generator(device="cuda").manual_seed(3)
    g = torch.rand(B, 26, D, H, W, generator=gen, device="cuda"); g /= g.sum(1, keepdim=True)
    x = torch.rand(B, C, D, H, W, generator=gen, device="cuda")
    go = torch.randn(B, C, D, H, W, generator=gen, device="cuda")
    xs = [x[:, c:c + 1].contiguous() for c in range(C)]
    gos = [go[:, c:c + 1].contiguous() for c in range(C)]
    res = {"op": "cspn3d_backward_multi_f32", "B": B, "C": C, "D": D, "H": H, "W": W, "n_iter": N}
    for n in (1, N):
        multi = timed(lambda: cspn_amd.cspn3d_backward_multi(g, x, go, n))
        loop = timed(lambda: [cspn_amd.cspn3d_backward(g, xs[c], gos[c], n) for c in range(C)])
        a = cspn_amd.cspn3d_backward_multi(g, x, go, n)
        b = [cspn_amd.cspn3d_backward(g, xs[c], gos[c], n) for c in range(C)]
        gsum = sum(p[0] for p in b)
        res["n_iter_%d" % n] = {"multi_ms": round(multi, 3), "per_channel_loop_ms": round(loop, 3), "speedup": round(loop / multi, 3),
                                "grad_feat_equal": bool(torch.equal(a[1], torch.cat([p[1] for p in b], 1))),
                                "grad_gate_rel_diff_vs_sum_of_calls": float((a[0] - gsum).abs().max() / gsum.abs().max())}
    fwd_multi = timed(lambda: cspn_amd.cspn3d_forward_multi(g, x, N))
    res["forward_multi_ms"] = round(fwd_multi, 3)
    cspn_amd.cspn3d_check_status()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
