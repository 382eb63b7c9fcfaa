#!/bin/bash
# single-variant experiment lib: parity of the plain variant, then timings of the named builds
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/tests
CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_$1.so timeout 120 python - <<'PY' 2>&1 | tee gpurun_out/asm2_first.log
import torch, numpy as np, cspn_amd
from helpers import make_inputs, rel_err
from oracle import cspn2d_oracle
for (B,H,W) in [(1,12,256),(2,17,304),(1,40,512),(3,304,1216)]:
    g,h,s = make_inputs(B,H,W,seed=3,sparse=False)
    ref = cspn2d_oracle(g,h,None,24,"8sum")
    o = cspn_amd.cspn2d_forward(g.cuda(),h.cuda(),None,24,"8sum","fused")
    torch.cuda.synchronize()
    o = o.cpu().numpy()
    bad = np.isnan(o) != np.isnan(ref)
    print(B,H,W,"nan-mismatch",int(bad.sum()),"err",float(np.nanmax(np.abs(o-ref))/np.nanmax(np.abs(ref))), flush=True)
PY
bash tools/gpu_abl_asm.sh "$@"
