#!/usr/bin/env python
"""tools/bench_aux.py -- time the on-device metrics reduction and Unpool against their byte counts."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cspn_amd import train_utils as T  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    B, H, W = 64, 304, 1216
    gt = torch.rand(B, 1, H, W, device="cuda") * 80
    pred = gt + torch.randn_like(gt)
    t = timeit(lambda: T._metrics(gt, pred))
    nbytes = gt.numel() * 8
    res = {"metrics_64x304x1216": {"ms": round(t * 1e3, 4), "GB/s": round(nbytes / t / 1e9, 1), "frac_of_8TBs": round(nbytes / t / 8e12, 3)}}
    x = torch.randn(16, 64, 152, 608, device="cuda")
    up = T.Unpool(64, 2)
    t = timeit(lambda: up(x))
    nbytes = x.numel() * 4 * 5
    res["unpool_16x64x152x608"] = {"ms": round(t * 1e3, 4), "GB/s": round(nbytes / t / 1e9, 1), "frac_of_8TBs": round(nbytes / t / 8e12, 3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
