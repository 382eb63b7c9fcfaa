#!/usr/bin/env python
"""tools/r06/bench_train_step.py [B] -- one training step of producer + path on one GPU: feature map -> both guidance heads -> 24 CSPN iterations -> masked L1 loss ->
backward to the feature map and both head weights.  The engine (guidance_heads + Affinity_Propagate in training mode + its loss kernel) against the reference's op
sequence in torch on the same GPU (conv_transpose2d Unpool + two conv2d, tools/torch_path.cspn2d_torch = cspn.py:42-83 as torch ops, torch autograd), same inputs;
gradients compared.  One JSON line."""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cspn_amd  # noqa: E402
from cspn_amd.train_utils import guidance_heads  # noqa: E402
from tools.torch_path import cspn2d_torch  # noqa: E402
from tools.r06.bench_head import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
C, h, w, N = 64, 152, 608, 24
H, W = 2 * h, 2 * w
gen = torch.Generator(device="cuda").manual_seed(21)
x = torch.randn(B, C, h, w, generator=gen, device="cuda")
w6 = torch.randn(8, C, 3, 3, generator=gen, device="cuda") / 24
w5 = torch.randn(1, C, 3, 3, generator=gen, device="cuda") / 24 + 0.02
sp = (torch.rand(B, 1, H, W, generator=gen, device="cuda") < 500.0 / (H * W)).float() * 5.0
gt = torch.rand(B, 1, H, W, generator=gen, device="cuda") * 10 + 0.5
prop = cspn_amd.Affinity_Propagate(N, 3, "8sum")
up = torch.zeros(C, 1, 2, 2, device="cuda")
up[:, :, 0, 0] = 1


def loss_of(pred):
    m = (gt > 0).float()
    return ((pred - gt).abs() * m).sum() / m.sum()


def engine_step(xa, wa, wb):
    g, b = guidance_heads(xa, wa, wb)
    return loss_of(prop(g, b, sp))


def torch_step(xa, wa, wb):
    U = F.conv_transpose2d(xa, up, stride=2, groups=C)
    return loss_of(cspn2d_torch(F.conv2d(U, wa, padding=1), F.conv2d(U, wb, padding=1), sp, N, "8sum"))


row = {"shape": [B, C, h, w], "n_iter": N}
grads = {}
for name, step in (("engine", engine_step), ("torch_reference_ops", torch_step)):
    xa, wa, wb = (t.clone().requires_grad_(True) for t in (x, w6, w5))

    def fb():
        xa.grad = wa.grad = wb.grad = None
        ls = step(xa, wa, wb)
        ls.backward()
        return ls
    try:
        ls = fb()
        row[name + "_loss"] = float(ls)
        row[name + "_ms"] = round(timeit(fb, reps=10 if name == "engine" else 3, warm=2)[0], 3)
        grads[name] = (xa.grad.clone(), wa.grad.clone(), wb.grad.clone())
        row[name + "_peak_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
    except Exception as ex:   # noqa: BLE001
        row[name + "_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:160])
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
if len(grads) == 2:
    a, b = grads["engine"], grads["torch_reference_ops"]
    row["grad_max_rel_diff"] = {k: float((p - q).abs().max() / q.abs().max()) for k, p, q in zip(("x", "w_guidance", "w_blur"), a, b)}
    row["speedup"] = round(row["torch_reference_ops_ms"] / row["engine_ms"], 1)
print(json.dumps(row), flush=True)
