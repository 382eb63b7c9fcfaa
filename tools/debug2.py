import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cspn_amd
from oracle import cspn2d_oracle
z = np.load(os.path.join(ROOT, "tests/golden/cspn2d_golden.npz"))
c = {k.split("/")[1]: z[k] for k in z.files if k.startswith("i_multiband_280/")}
g, h, s = torch.from_numpy(c["guidance"]), torch.from_numpy(c["blur"]), torch.from_numpy(c["sparse"])
for rep in range(3):
    out = cspn_amd.cspn2d_forward(g.cuda(), h.cuda(), s.cuda(), 24, "8sum", "fused").cpu().numpy()
    nan = np.isnan(out); err = np.abs(np.nan_to_num(out) - c["out"])
    print("golden i rep", rep, "nan", int(nan.sum()), "maxerr", err.max())
    if nan.sum():
        print(" nan/row", nan[0,0].sum(1).tolist()); print(" nan/col first", np.where(nan[0,0].sum(0)>0)[0][:20])
# sparse negative / special values?
print("sparse stats", float(s.min()), float(s.max()), int((s>0).sum()))
for B in (2, 8, 16, 32, 64):
    gen = torch.Generator(device="cuda").manual_seed(1)
    G = torch.randn(B, 8, 304, 1216, generator=gen, device="cuda"); Hh = torch.rand(B, 1, 304, 1216, generator=gen, device="cuda") * 80
    try:
        o = cspn_amd.cspn2d_forward(G, Hh, None, 24, "8sum", "fused"); torch.cuda.synchronize()
        o2 = cspn_amd.cspn2d_forward(G, Hh, None, 24, "8sum", "stepwise"); torch.cuda.synchronize()
        print("B", B, "ok; fused vs stepwise max diff", float((o - o2).abs().max()), "nan", int(torch.isnan(o).sum()))
    except Exception as e:
        print("B", B, "EXC", e)
