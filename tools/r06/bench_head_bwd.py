#!/usr/bin/env python
"""tools/r06/bench_head_bwd.py -- the heads' gradient (cspn_guidance_head_backward_f32) at [64,64,152,608]: dL/dx alone, dL/dW alone, both; torch's autograd through the
reference's op sequence (conv_transpose2d + two conv2d) on the same GPU beside it.  One JSON line."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cspn_amd.train_utils import guidance_heads_backward  # noqa: E402
from tools.r06.bench_head import timeit  # noqa: E402

B, C, h, w = (int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (64, 64, 152, 608)
H, W = 2 * h, 2 * w
gen = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(B, C, h, w, generator=gen, device="cuda")
w6 = torch.randn(8, C, 3, 3, generator=gen, device="cuda") / 24
w5 = torch.randn(1, C, 3, 3, generator=gen, device="cuda") / 24
gg = torch.randn(B, 8, H, W, generator=gen, device="cuda")
gb = torch.randn(B, 1, H, W, generator=gen, device="cuda")
flop = 2.0 * 81 * C * B * h * w
row = {"shape": [B, C, h, w], "gflop_each": round(flop / 1e9, 2)}
for name, kw in (("grad_x", dict(need_w=False)), ("grad_w", dict(need_x=False)), ("both", {})):
    for r in range(2):
        avg, mn = timeit(lambda: guidance_heads_backward(x, w6, w5, gg, gb, **kw))
    row[name + "_ms"] = round(avg, 4)
row["grad_x_tflops"] = round(flop / (row["grad_x_ms"] * 1e-3) / 1e12, 1)
row["grad_w_tflops"] = round(flop / (row["grad_w_ms"] * 1e-3) / 1e12, 1)
try:
    up = torch.zeros(C, 1, 2, 2, device="cuda")
    up[:, :, 0, 0] = 1
    xa, w6a, w5a = (t.clone().requires_grad_(True) for t in (x, w6, w5))

    def torch_fb():
        xa.grad = w6a.grad = w5a.grad = None
        U = F.conv_transpose2d(xa, up, stride=2, groups=C)
        ((F.conv2d(U, w6a, padding=1) * gg).sum() + (F.conv2d(U, w5a, padding=1) * gb).sum()).backward()
    row["torch_forward_plus_backward_ms"] = round(timeit(torch_fb, reps=5, warm=2)[0], 2)
    dx, d6, d5 = guidance_heads_backward(x, w6, w5, gg, gb)
    # dL/dW by its definition in fp64 on the GPU: the yardstick for both
    xd, gp = x.double(), F.pad(torch.cat([gg, gb], 1).double(), (1, 1, 1, 1))
    ref = torch.zeros(9, C, 3, 3, dtype=torch.float64, device="cuda")
    for ky in range(3):
        for kx in range(3):
            ref[:, :, ky, kx] = torch.einsum("bcyx,boyx->oc", xd, gp[:, :, 2 - ky:2 - ky + 2 * h:2, 2 - kx:2 - kx + 2 * w:2])
    rel = lambda a: float((a.double() - ref).abs().max() / ref.abs().max())   # noqa: E731
    row["grad_w_max_rel_vs_fp64_definition"] = {"engine": rel(torch.cat([d6, d5], 0)), "torch_fp32_autograd": rel(torch.cat([w6a.grad, w5a.grad], 0))}
    row["grad_x_max_rel_vs_torch"] = float((dx - xa.grad).abs().max() / xa.grad.abs().max())
except Exception as ex:   # noqa: BLE001
    row["torch_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:120])
print(json.dumps(row), flush=True)
