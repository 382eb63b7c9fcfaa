#!/bin/bash
# tagged-quad halo exchange of the persistent 3D kernel: parity, repeats, timing, phase trace
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "3d" 2>&1 | tail -3 | tee gpurun_out/r2t.txt
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2t.txt
import os, sys, torch
sys.path.insert(0, os.getcwd())
import cspn_amd
for (B, D, H, W, N) in [(4, 32, 160, 608, 12), (2, 20, 30, 200, 12), (1, 32, 160, 304, 5), (1, 8, 8, 64, 3), (3, 9, 17, 72, 4), (2, 32, 160, 608, 24)]:
    g = torch.rand(B, 26, D, H, W, device="cuda"); g /= g.sum(1, keepdim=True)
    h = torch.rand(B, 1, D, H, W, device="cuda")
    ref = cspn_amd.cspn3d_forward(g, h, None, N, "none", algo="stepwise")
    fl = 0
    for i in range(25):
        o, ws = cspn_amd.cspn3d_forward(g, h, None, N, "none", algo="persistent", _return_ws=True)
        fl += int(not torch.equal(o, ref))
        torch.cuda.synchronize()
        fl += int(cspn_amd.load().cspn_debug_3d_persistent_error(ws.data_ptr(), B, D, H, W) != 0)
    print("3D %s: repeats differing / timeouts: %d  maxdiff %.3g" % ((B, D, H, W, N), fl, (o - ref).abs().max().item()), flush=True)
PY
timeout 300 python bench.py --workload vol3d --steps 50 --warmup 10 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r2t.txt
CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_p3trace.so timeout 300 python tools/probe_3d_trace.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2t.txt
