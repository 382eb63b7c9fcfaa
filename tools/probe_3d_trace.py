"""per-step phases of the persistent 3D kernel (build with -DP3_TRACE): cycles of workgroup 37 in chunk 1"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cspn_amd
B, D, H, W, N = 4, 32, 160, 608, 12
g = torch.rand(B, 26, D, H, W, device="cuda"); g /= g.sum(1, keepdim=True)
h = torch.rand(B, 1, D, H, W, device="cuda")
for _ in range(3):
    o, ws = cspn_amd.cspn3d_forward(g, h, None, N, "none", algo="persistent", _return_ws=True)
torch.cuda.synchronize()
sync = ws[2 * B * D * H * W * 4 + 2 * 256 * 1664 * 16:].view(torch.int32)
raw = sync[2048:2048 + (N + 1) * 12].cpu().numpy().view(np.uint64).astype(np.int64)
t = raw[:N * 6].reshape(N, 6)
ck = raw[N * 6:N * 6 + 4]
print("chunk: gate loads %d, level-0 fill %d, %d steps %d cycles" % (ck[1] - ck[0], ck[2] - ck[1], N, ck[3] - ck[2]))
print("mean cycles per phase (shader clock):")
print("  compute                          %10.1f" % (t[:N - 1, 1] - t[:N - 1, 0]).mean())
print("  publish + first poll round trip  %10.1f" % (t[:N - 1, 2] - t[:N - 1, 1]).mean())
print("  further polls, halo to LDS, sync %10.1f" % (t[:N - 1, 4] - t[:N - 1, 2]).mean())
print("  step to step                     %10.1f" % np.diff(t[:N - 1, 0]).mean())
