import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cspn_amd
lib = cspn_amd.load()
B,H,W,N = 64,304,1216,24
gen = torch.Generator(device="cuda").manual_seed(1)
g = torch.randn(B,8,H,W,generator=gen,device="cuda"); h = torch.rand(B,1,H,W,generator=gen,device="cuda")*80
for r in range(3):
    o = cspn_amd.cspn2d_forward(g,h,None,N,"8sum","fused"); torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
lib.cspn_debug_timing(buf)
t = np.array(list(buf)).reshape(8,8)
names = ["post-barrier->reads", "cook (consume+issue)", "step fast", "step slow", "barrier wait", "  of cook: consume", "", ""]
tot = t[:, :5].sum(1)
print("per-wave cycle totals (block 0), clock64 ticks; total per wave:", tot.tolist())
for i,n in enumerate(names[:6]):
    print("%-24s" % n, " ".join("%9d" % v for v in t[:, i]), "  avg %.1f%%" % (100.0*t[:, i].mean()/tot.mean()))
