#!/usr/bin/env python
"""tools/bench_torch_ops_baseline.py -- "the reference's PyTorch path on this GPU" (SURVEY.md 8d "reference on MI355X"), as far
as it can be had on a box without /root/reference: tools/torch_ops_baseline.py launches the same sequence of torch ops per
iteration as cspn.py:42-83 (pinned to the unmodified reference's golden vectors by tests/test_oracle.py) and is timed here next
to the HIP engine on the same inputs.  Prints one JSON object; run on the GPU box:
    python tools/bench_torch_ops_baseline.py [out.json]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cspn_amd  # noqa: E402
from tools.torch_ops_baseline import affinity_propagate_torch_ops  # noqa: E402

CASES = [("BASELINE config 3: KITTI 304x1216 x 64, 24 iters", 64, 304, 1216, 24, False, 80.0),
         ("BASELINE config 4: KITTI 304x1216 x 32, 24 iters, sparse", 32, 304, 1216, 24, True, 80.0),
         ("BASELINE config 2: NYU 228x304 x 16, 24 iters", 16, 228, 304, 24, False, 10.0)]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def main():
    res = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "cases": []}
    for desc, B, H, W, N, sparse, scale in CASES:
        gen = torch.Generator(device="cuda").manual_seed(B + W)
        g = torch.randn(B, 8, H, W, generator=gen, device="cuda")
        h = torch.rand(B, 1, H, W, generator=gen, device="cuda") * scale
        s = (torch.rand(B, 1, H, W, generator=gen, device="cuda") < 500.0 / (H * W)).float() * (h + 0.1) if sparse else None
        row = {"workload": desc, "mpix_iters": B * H * W * N / 1e6}
        ms_e, out_e = timed(lambda: cspn_amd.cspn2d_forward(g, h, s, N, "8sum"), 50)
        row["hip_engine_ms"] = round(ms_e, 4)
        for mode in ("sum", "conv3d"):
            t0 = time.perf_counter()
            try:
                ms, out = timed(lambda: affinity_propagate_torch_ops(g, h, s, N, "8sum", mode), 3)
            except Exception as exc:  # (a MIOpen without a usable 1x1x1 Conv3d must not take the other numbers with it)
                row["torch_ops_%s_error" % mode] = repr(exc)[:200]
                continue
            err = float((out - out_e).abs().max() / out_e.abs().max())
            row["torch_ops_%s_ms" % mode] = round(ms, 3)
            row["torch_ops_%s_mpix_iters_per_s" % mode] = round(row["mpix_iters"] / ms * 1e3, 1)
            row["torch_ops_%s_vs_engine_rel_diff" % mode] = err
            row["engine_speedup_over_torch_ops_%s" % mode] = round(ms / ms_e, 1)
            row["torch_ops_%s_wall_s_incl_first_call" % mode] = round(time.perf_counter() - t0, 1)
            del out
        res["cases"].append(row)
        del g, h, s, out_e
        torch.cuda.empty_cache()
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
