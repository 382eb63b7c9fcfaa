#!/usr/bin/env python
"""tools/bench_backward.py -- time the 2D backward (cspn2d_backward_f32) at BASELINE config 3's image size.
One JSON line: ms per backward call, pix*iters/s, and the fraction of the 8 TB/s roofline priced with the algorithmic
bytes of a gradient computation that touches every tensor once (guidance 32 + blur 4 + grad_out 4 in, grad_guidance 32 +
grad_blur 4 out = 76 B/pixel, + 4 with a sparse mask)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cspn_amd  # noqa: E402


def vol3d(a):
    """cspn3d_backward_f32 at config 5 (4 x 32x160x608): the single chained call (n_iter 1, how the Paddle graph uses the op)
    and the fused 12-step op.  Algorithmic bytes of a gradient that touches every tensor once: gate 104 + feat 4 + grad_out 4
    in, grad_gate 104 + grad_feat 4 out = 220 B/voxel."""
    B, D, H, W = (a.batch if a.batch != 16 else 4), 32, 160, 608
    gen = torch.Generator(device="cuda").manual_seed(1)
    g = torch.rand(B, 26, D, H, W, generator=gen, device="cuda"); g /= g.sum(1, keepdim=True)
    h = torch.rand(B, 1, D, H, W, generator=gen, device="cuda")
    go = torch.randn(B, 1, D, H, W, generator=gen, device="cuda")
    vox = B * D * H * W
    for N in (1, 12):
        for _ in range(2):
            cspn_amd.cspn3d_backward(g, h, go, N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            cspn_amd.cspn3d_backward(g, h, go, N)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        print(json.dumps({"op": "cspn3d_backward_f32", "B": B, "D": D, "H": H, "W": W, "n_iter": N, "ms_per_call": round(ms, 3),
                          "Mvox_iters_per_s": round(vox * N / ms / 1e3, 1), "algorithmic_bytes": vox * 220,
                          "roofline_frac": round(vox * 220 / (ms * 1e-3) / 8e12, 4),
                          "note": "n_iter - 1 forward steps keeping the levels + n_iter adjoint steps (112 B/voxel each) + one gate-gradient pass"}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--sparse", action="store_true")
    ap.add_argument("--vol3d", action="store_true", help="the 3D backward (Paddle contract) at BASELINE config 5's volume")
    ap.add_argument("--n-iter", type=int, default=24, help="2D: iterations (multiples of 4 up to 24 take the checkpointed ring path)")
    ap.add_argument("--norm", default="8sum", choices=["8sum", "8sum_abs", "none", "prenorm"],
                    help="prenorm (round 6): the guidance is normalised once, outside the timed region (cspn2d_normalize_f32); the timed calls take the reference's gate_wb")
    a = ap.parse_args()
    if a.vol3d:
        return vol3d(a)
    B, H, W, N = a.batch, 304, 1216, a.n_iter
    gen = torch.Generator(device="cuda").manual_seed(1)
    g = torch.randn(B, 8, H, W, generator=gen, device="cuda")
    h = torch.rand(B, 1, H, W, generator=gen, device="cuda") * 80
    s = None
    if a.sparse:
        s = (torch.rand(B, 1, H, W, generator=gen, device="cuda") < 500.0 / (H * W)).float() * (h + 0.1)
    go = torch.randn(B, 1, H, W, generator=gen, device="cuda")
    if a.norm == "prenorm":
        g = cspn_amd.cspn2d_normalize(g, "8sum")
    elif a.norm == "none":
        g = g.abs() / g.abs().sum(1, keepdim=True)
    for _ in range(3):
        cspn_amd.cspn2d_backward(g, h, s, go, N, a.norm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        cspn_amd.cspn2d_backward(g, h, s, go, N, a.norm)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    # a training step through the module (forward + backward), with and without the level history kept by the forward
    train = {}
    for keep in (True, False):
        if a.norm in ("8sum", "8sum_abs"):
            m = cspn_amd.Affinity_Propagate(N, 3, a.norm)
            m.keep_history = keep
        elif a.norm == "prenorm":
            m = lambda gd, hd, sd, keep=keep: cspn_amd.propagate_prenorm(gd, hd, sd, N, keep_history=keep)   # noqa: E731
        else:
            break
        gr, hr = g.clone().requires_grad_(True), h.clone().requires_grad_(True)
        for _ in range(3):
            m(gr, hr, s).backward(go)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            gr.grad = hr.grad = None
            m(gr, hr, s).backward(go)
        torch.cuda.synchronize()
        train["keep_history" if keep else "recompute"] = round((time.perf_counter() - t0) / a.steps * 1e3, 3)
    px = B * H * W
    alg = px * (80 if a.sparse else 76)
    print(json.dumps({"op": "cspn2d_backward_f32", "B": B, "H": H, "W": W, "n_iter": N, "sparse": a.sparse, "norm": a.norm,
                      "ms_per_call": round(ms, 3), "Mpix_iters_per_s": round(px * N / ms / 1e3, 1),
                      "algorithmic_bytes": alg, "roofline_frac": round(alg / (ms * 1e-3) / 8e12, 4),
                      "train_step_fwd_bwd_ms": train,
                      "note": "forward and adjoint sweeps each as one launch of the fused ring kernel keeping every fourth of its levels "
                              "(the forward one also the folded coefficients), + the final pass that recomputes the levels in between "
                              "tile by tile; workspace 20 planes"}))


if __name__ == "__main__":
    main()
