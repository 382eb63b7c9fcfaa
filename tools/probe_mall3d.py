"""tools/probe_mall3d.py -- does a 3D propagation step run faster when its gate tensor fits in the 256 MB Infinity Cache?
Times the one-launch-per-iteration kernel (step3d_direct_kernel, 112 B/voxel) on sub-volumes of BASELINE config 5."""
import sys, os, json, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cspn_amd

dev = "cuda:0"
D, H = 32, 160
res = []
for B, W in ((1, 76), (1, 152), (1, 304), (1, 608), (2, 608), (4, 608)):
    g = torch.rand(B, 26, D, H, W, device=dev)
    g /= g.sum(1, keepdim=True)
    h = torch.rand(B, 1, D, H, W, device=dev)
    for _ in range(3):
        cspn_amd.cspn3d_forward(g, h, None, 12, "none")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        cspn_amd.cspn3d_forward(g, h, None, 12, "none")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps / 12
    vox = B * D * H * W
    res.append({"B": B, "W": W, "gate_MB": round(vox * 104 / 1e6, 1), "us_per_launch": round(ms * 1e3, 2),
                "GBps_algorithmic": round(vox * 112 / (ms * 1e-3) / 1e9, 1)})
    print(res[-1], flush=True)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r2c_mall3d.json"), "w"), indent=1)
