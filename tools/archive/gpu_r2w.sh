#!/bin/bash
# 3D: the normalising / masked modes fused (fold + persistent) against fold + one launch per step; main bench line
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2w.txt
import os, sys, torch
sys.path.insert(0, os.getcwd())
import cspn_amd
B, D, H, W, N = 4, 32, 160, 608, 12
gen = torch.Generator(device="cuda").manual_seed(1)
g = torch.rand(B, 26, D, H, W, generator=gen, device="cuda")
h = torch.rand(B, 1, D, H, W, generator=gen, device="cuda")
s = (torch.rand(B, 1, D, H, W, generator=gen, device="cuda") < 0.02).float() * (h + 0.1)
for norm, sp in (("8sum_abs", None), ("8sum_abs", s), ("none", s)):
    res = {}
    for algo in ("persistent", "stepwise"):
        for _ in range(3): o = cspn_amd.cspn3d_forward(g, h, sp, N, norm, algo=algo)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): o = cspn_amd.cspn3d_forward(g, h, sp, N, norm, algo=algo)
        e1.record(); torch.cuda.synchronize()
        res[algo] = round(e0.elapsed_time(e1) / 10, 3)
    print("config 5 volume, norm %s, mask %s: ms per forward (fold + steps) %s" % (norm, sp is not None, res), flush=True)
PY
timeout 300 python bench.py --workload vol3d --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vol3d bench', d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline']['frac'])" | tee -a gpurun_out/r2w.txt
