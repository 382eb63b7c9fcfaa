import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cspn_amd
lib = cspn_amd.load()
def trace():
    buf = (ctypes.c_int * (2048*8))(); n = ctypes.c_int(0)
    lib.cspn_debug_trace(buf, ctypes.byref(n))
    return np.array(buf[:min(n.value,2048)*8]).reshape(-1,8)
B,H,W,N = 1,26,280,24
gen = torch.Generator(device="cuda").manual_seed(B+H+W)
g = torch.randn(B,8,H,W,generator=gen,device="cuda"); h = torch.rand(B,1,H,W,generator=gen,device="cuda")*10
s = ((torch.rand(B,1,H,W,generator=gen,device="cuda")<0.01).float()*(h+0.1))
ref = cspn_amd.cspn2d_forward(g,h,s,N,"8sum","stepwise"); torch.cuda.synchronize()
good = None
for r in range(60):
    o = cspn_amd.cspn2d_forward(g,h,s,N,"8sum","fused"); torch.cuda.synchronize()
    t = trace(); t = t[np.lexsort((t[:,2], t[:,1], t[:,0]))]
    dd = (o-ref).abs(); dd[torch.isnan(dd)] = 1e9
    bad = float(dd.max()) > 1e-3
    if good is None and not bad: good = t
    if bad:
        print("rep", r, "BAD; trace rows", len(t))
        if good is not None:
            diff = np.where((t != good).any(1))[0] if len(t)==len(good) else None
            print(" differing trace entries (tau,wv,j,q,act,outoff,lo,hi): mine vs good")
            if diff is not None:
                for i in diff[:30]: print("  ", t[i].tolist(), " | ", good[i].tolist())
        else:
            print(t[:40])
        break
print("done; good trace sample:"); print(good[:12] if good is not None else None)
