#!/usr/bin/env python
"""tools/r06/time_head.py -- device time of the raw guidance head at KITTI x 64 for the library named by CSPN_AMD_LIB (ablation builds of
tools/r06/build_abl_head.sh: wrong results, timing only).  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cspn_amd.train_utils import guidance_heads  # noqa: E402
from tools.r06.bench_head import timeit  # noqa: E402

B, C, h, w = 64, 64, 152, 608
gen = torch.Generator(device="cuda").manual_seed(11)
x = torch.randn(B, C, h, w, generator=gen, device="cuda")
w6 = torch.randn(8, C, 3, 3, generator=gen, device="cuda") / 24.0
w5 = torch.randn(1, C, 3, 3, generator=gen, device="cuda") / 24.0
out = {"lib": os.path.basename(os.environ.get("CSPN_AMD_LIB", "libcspn_amd.so"))}
for r in range(2):
    avg, mn = timeit(lambda: guidance_heads(x, w6, w5), reps=20, warm=5)
    out["ms_%d" % r] = round(avg, 4)
out["tflops"] = round(2.0 * B * h * w * C * 81 / (out["ms_1"] * 1e-3) / 1e12, 1)
print(json.dumps(out), flush=True)
