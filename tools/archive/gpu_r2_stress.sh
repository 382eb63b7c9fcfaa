#!/bin/bash
# repeated launches of every assembly path + the persistent 3D kernel: run-to-run flicker would reveal a race / missing wait
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
for i in 1 2 3; do timeout 300 python tools/stress_asm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_stress.txt; done
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_stress.txt
import os, sys, torch
sys.path.insert(0, os.getcwd())
import cspn_amd
bad = 0
for (B, D, H, W, N) in [(4, 32, 160, 608, 12), (2, 20, 30, 200, 12), (1, 32, 160, 304, 5)]:
    g = torch.rand(B, 26, D, H, W, device="cuda"); g /= g.sum(1, keepdim=True)
    h = torch.rand(B, 1, D, H, W, device="cuda")
    ref = cspn_amd.cspn3d_forward(g, h, None, N, "none", algo="stepwise")
    fl = 0
    for i in range(25):
        o, ws = cspn_amd.cspn3d_forward(g, h, None, N, "none", algo="persistent", _return_ws=True)
        fl += int(not torch.equal(o, ref))
        torch.cuda.synchronize()
        fl += int(cspn_amd.load().cspn_debug_3d_persistent_error(ws.data_ptr(), B, D, H, W) != 0)
    print("3D %s: repeats differing from the per-step kernel or reporting a sync timeout: %d" % ((B, D, H, W, N), fl), flush=True)
    bad += fl
# sited8 + sparse + abs variants of the 2D loop
for norm, sp in (("8sum_abs", True), ("none", False)):
    gen = torch.Generator(device="cuda").manual_seed(5)
    g = torch.randn(8, 8, 304, 1216, generator=gen, device="cuda")
    if norm == "none": g = g.abs() / (g.abs().sum(1, keepdim=True) + 0.25)
    h = torch.rand(8, 1, 304, 1216, generator=gen, device="cuda") * 80
    s = (torch.rand(8, 1, 304, 1216, generator=gen, device="cuda") < 0.01).float() * (h + 0.1) if sp else None
    ref = cspn_amd.cspn2d_forward(g, h, s, 24, norm, "fused")
    g8 = cspn_amd.guidance_to_sited8(g, norm)
    fl = 0
    for i in range(30):
        fl += int(not torch.equal(cspn_amd.cspn2d_forward(g, h, s, 24, norm, "fused"), ref))
        fl += int(not torch.equal(cspn_amd.cspn2d_forward_sited8(g8, h, s, 24, norm), ref))
        fl += int(not torch.equal(cspn_amd.cspn2d_forward(g, h, s, 48, norm, "fused"), cspn_amd.cspn2d_forward(g, h, s, 48, norm, "fused")))
    print("2D %s sparse=%s: non-identical repeats %d" % (norm, sp, fl), flush=True)
    bad += fl
# folded 3D modes (fold + persistent kernel with the constant term) and the fused 3D backward: repeats must be bit-identical
B, D, H, W, N = 2, 32, 160, 304, 12
gen = torch.Generator(device="cuda").manual_seed(9)
g = torch.rand(B, 26, D, H, W, generator=gen, device="cuda")
h = torch.rand(B, 1, D, H, W, generator=gen, device="cuda")
s = (torch.rand(B, 1, D, H, W, generator=gen, device="cuda") < 0.02).float() * (h + 0.1)
ref = cspn_amd.cspn3d_forward(g, h, s, N, "8sum_abs", algo="stepwise")
fl = sum(int(not torch.equal(cspn_amd.cspn3d_forward(g, h, s, N, "8sum_abs"), ref)) for _ in range(20))
print("3D folded (8sum_abs + mask) fused vs per-step: differing repeats %d" % fl, flush=True)
bad += fl
gn = g / g.sum(1, keepdim=True)
go = torch.randn(B, 1, D, H, W, generator=gen, device="cuda")
gg0, gf0 = cspn_amd.cspn3d_backward(gn, h, go, N)
fl = 0
for _ in range(15):
    gg, gf = cspn_amd.cspn3d_backward(gn, h, go, N)
    fl += int(not torch.equal(gg, gg0)) + int(not torch.equal(gf, gf0))
print("3D fused backward: non-identical repeats %d" % fl, flush=True)
bad += fl
g2 = torch.randn(8, 8, 304, 1216, generator=gen, device="cuda")
h2 = torch.rand(8, 1, 304, 1216, generator=gen, device="cuda") * 80
go2 = torch.randn(8, 1, 304, 1216, generator=gen, device="cuda")
a0, b0 = cspn_amd.cspn2d_backward(g2, h2, None, go2, 24, "8sum")
fl = 0
for _ in range(15):
    a, b = cspn_amd.cspn2d_backward(g2, h2, None, go2, 24, "8sum")
    fl += int(not torch.equal(a, a0)) + int(not torch.equal(b, b0))
print("2D backward: non-identical repeats %d" % fl, flush=True)
bad += fl
print("STRESS2", "OK" if bad == 0 else "FAILED")
PY
