#!/usr/bin/env python
"""tools/r06/bench_head.py -- the guidance head (cspn_guidance_head_f32: Unpool + 3x3 conv 64 -> 8 / 64 -> 1 as one kernel) against the reference's op sequence
in torch on the same GPU (conv_transpose2d Unpool + two conv2d, torch_resnet_cspn_nyu.py:41-54,187-206), and end to end with the propagation:
    head (raw guidance) + forward '8sum'   |   head (gate_wb) + forward 'prenorm'   |   torch heads + forward '8sum'
One JSON line per shape.  Algorithmic work of the head: 9 products per input pixel, input channel and output channel (the other 27 taps are zeros)."""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cspn_amd  # noqa: E402
from cspn_amd.train_utils import guidance_heads  # noqa: E402


def timeit(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    ev = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        fn()
        e1.record(st)
        ev.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ms) / len(ms), ms[0]


def main():
    shapes = [(64, 152, 608), (16, 114, 152)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    C = 64
    for (B, h, w) in shapes:
        H, W = 2 * h, 2 * w
        gen = torch.Generator(device="cuda").manual_seed(11)
        x = torch.randn(B, C, h, w, generator=gen, device="cuda")
        w6 = torch.randn(8, C, 3, 3, generator=gen, device="cuda") / (3.0 * C ** 0.5)
        w5 = torch.randn(1, C, 3, 3, generator=gen, device="cuda") / (3.0 * C ** 0.5) + 0.02
        up = torch.zeros(C, 1, 2, 2, device="cuda")
        up[:, :, 0, 0] = 1

        def torch_heads():
            U = F.conv_transpose2d(x, up, stride=2, groups=C)
            return F.conv2d(U, w6, padding=1), F.conv2d(U, w5, padding=1)
        row = {"shape": [B, C, h, w], "out": [H, W]}
        flop = 2.0 * B * h * w * C * 9 * 9
        row["algorithmic_gflop"] = round(flop / 1e9, 2)
        for name, fn in (("head_raw", lambda: guidance_heads(x, w6, w5)), ("head_gate_wb", lambda: guidance_heads(x, w6, w5, norm_type="8sum")),
                         ("torch_heads", torch_heads)):
            try:
                avg, mn = timeit(fn)
                row[name + "_ms"] = round(avg, 4)
                row[name + "_tflops"] = round(flop / (avg * 1e-3) / 1e12, 1)
            except Exception as ex:   # noqa: BLE001
                row[name + "_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:120])
            torch.cuda.empty_cache()
        g, b = guidance_heads(x, w6, w5)
        tg, tb = torch_heads()
        row["raw_vs_torch_max_rel"] = float((g - tg).abs().max() / tg.abs().max())
        del tg, tb
        depth = b.abs() * 10 + 1
        row["normalize_standalone_ms"] = round(timeit(lambda: cspn_amd.cspn2d_normalize(g, "8sum"))[0], 4)
        row["forward_8sum_ms"] = round(timeit(lambda: cspn_amd.cspn2d_forward(g, depth, None, 24, "8sum"))[0], 4)
        wb, _ = guidance_heads(x, w6, w5, norm_type="8sum")
        row["forward_prenorm_ms"] = round(timeit(lambda: cspn_amd.cspn2d_forward(wb, depth, None, 24, "prenorm"))[0], 4)
        a = cspn_amd.cspn2d_forward(wb, depth, None, 24, "prenorm")
        c = cspn_amd.cspn2d_forward(g, depth, None, 24, "8sum")
        row["prenorm_vs_raw_path_max_rel"] = float((a - c).abs().max() / c.abs().max())
        del a, c, g, b, wb

        def e2e_raw():
            g_, b_ = guidance_heads(x, w6, w5)
            return cspn_amd.cspn2d_forward(g_, depth, None, 24, "8sum")

        def e2e_wb():
            g_, b_ = guidance_heads(x, w6, w5, norm_type="8sum")
            return cspn_amd.cspn2d_forward(g_, depth, None, 24, "prenorm")

        def e2e_torch():
            g_, b_ = torch_heads()
            return cspn_amd.cspn2d_forward(g_, depth, None, 24, "8sum")
        for name, fn in (("e2e_head_raw_plus_forward", e2e_raw), ("e2e_head_gate_wb_plus_prenorm_forward", e2e_wb), ("e2e_torch_heads_plus_forward", e2e_torch)):
            try:
                row[name + "_ms"] = round(timeit(fn, reps=10, warm=3)[0], 4)
            except Exception as ex:   # noqa: BLE001
                row[name + "_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:120])
            torch.cuda.empty_cache()
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
