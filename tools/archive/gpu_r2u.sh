#!/bin/bash
# timing variants of the persistent 3D kernel (tools/build_p3var.sh): config 5 forward per library
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && : > gpurun_out/r2u.txt
for lib in "" $(ls cspn_amd/abl/libcspn_*.so 2>/dev/null); do
  echo "== ${lib:-product}" | tee -a gpurun_out/r2u.txt
  CSPN_AMD_LIB=${lib:+$PWD/$lib} timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2u.txt
import os, sys, torch
sys.path.insert(0, os.getcwd())
import cspn_amd
B, D, H, W = 4, 32, 160, 608
g = torch.rand(B, 26, D, H, W, device="cuda"); g /= g.sum(1, keepdim=True)
h = torch.rand(B, 1, D, H, W, device="cuda")
for N in (12, 2):
    ref = cspn_amd.cspn3d_forward(g, h, None, N, "none", algo="stepwise")
    for _ in range(5): o = cspn_amd.cspn3d_forward(g, h, None, N, "none", algo="persistent")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): o = cspn_amd.cspn3d_forward(g, h, None, N, "none", algo="persistent")
    e1.record(); torch.cuda.synchronize()
    print("n_iter %2d: %.4f ms  equal to stepwise: %s" % (N, e0.elapsed_time(e1) / 20, torch.equal(o, ref)), flush=True)
PY
done
