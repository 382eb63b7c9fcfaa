import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cspn_amd
from helpers import make_inputs
from oracle import cspn2d_oracle
np.set_printoptions(linewidth=200, precision=3, suppress=True)
cases = [(1,8,64,1,False),(1,8,64,2,False),(1,8,64,3,False),(1,16,64,24,False),(1,40,64,24,True),(1,26,280,24,True),(2,37,256,24,True),(1,304,1216,24,False)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (B,H,W,N,sp) in cases:
    g,h,s = make_inputs(B,H,W,seed=H+W+N,sparse=bool(sp))
    ref = cspn2d_oracle(g,h,s,N,"8sum")
    try:
        out = cspn_amd.cspn2d_forward(g.cuda(),h.cuda(),None if s is None else s.cuda(),N,"8sum","fused")
        torch.cuda.synchronize()
    except Exception as e:
        print((B,H,W,N,sp), "EXC", e); continue
    o = out.cpu().numpy()
    nan = np.isnan(o)
    err = np.abs(np.nan_to_num(o) - ref)
    print("case", (B,H,W,N,sp), "nan", int(nan.sum()), "maxerr %.3e" % err.max(), "ref max %.2f" % np.abs(ref).max())
    if nan.sum() or err.max() > 1e-3:
        for b in range(B):
            rows_nan = nan[b,0].sum(1); rows_err = err[b,0].max(1)
            print("  img",b,"nan/row:", rows_nan[:48].tolist())
            print("  err/row:", np.round(rows_err[:48],3).tolist())
            cols_err = err[b,0].max(0)
            bad = np.where(cols_err > 1e-3)[0]
            print("  bad cols:", bad[:40].tolist(), "... n=", len(bad))
