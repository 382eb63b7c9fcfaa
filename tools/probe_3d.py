"""tools/probe_3d.py -- persistent vs stepwise 3D propagation on config 5 and sub-shapes: ms per forward"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cspn_amd
dev = "cuda:0"
for B, D, H, W, N in ((1, 32, 160, 152, 12), (1, 32, 160, 608, 12), (4, 32, 160, 608, 12), (4, 32, 160, 608, 2)):
    g = torch.rand(B, 26, D, H, W, device=dev); g /= g.sum(1, keepdim=True)
    h = torch.rand(B, 1, D, H, W, device=dev)
    res = {}
    for algo in ("persistent", "stepwise"):
        for _ in range(3):
            o = cspn_amd.cspn3d_forward(g, h, None, N, "none", algo=algo)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            o = cspn_amd.cspn3d_forward(g, h, None, N, "none", algo=algo)
        e1.record(); torch.cuda.synchronize()
        res[algo] = round(e0.elapsed_time(e1) / 5, 4)
    print(B, D, H, W, N, res, flush=True)
