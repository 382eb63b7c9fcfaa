#!/usr/bin/env python
"""tools/r06/time_fwd.py [B H W reps mode] -- device time per launch (HIP events) of the fused 2D forward (24 iterations, 8sum, no mask) with the library CSPN_AMD_LIB
points at; mode 0: the product's dispatch, 8: the 8 x 4 ring.  One line: mean / min / median ms."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.fuzz_parity import forward2d_plan  # noqa: E402

a = sys.argv[1:]
B, H, W = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (64, 304, 1216)
reps = int(a[3]) if len(a) > 3 else 60
mode = int(a[4]) if len(a) > 4 else 0
gen = torch.Generator(device="cuda").manual_seed(7)
g = torch.randn(B, 8, H, W, generator=gen, device="cuda")
h = torch.rand(B, 1, H, W, generator=gen, device="cuda") * 80
st = torch.cuda.current_stream()
for _ in range(40):
    forward2d_plan(g, h, None, 24, "8sum", mode)
torch.cuda.synchronize()
ev = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    forward2d_plan(g, h, None, 24, "8sum", mode)
    e1.record(st)
    ev.append((e0, e1))
torch.cuda.synchronize()
ms = sorted(x.elapsed_time(y) for x, y in ev)
print("%s mode %d: mean %.4f min %.4f median %.4f ms" % (os.path.basename(os.environ.get("CSPN_AMD_LIB", "product")), mode, sum(ms) / len(ms), ms[0], ms[len(ms) // 2]))
