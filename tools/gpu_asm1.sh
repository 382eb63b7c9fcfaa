#!/bin/bash
# first contact of the assembly main loop with a real GPU: tiny case under a short timeout, then the test-suite, then bench
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/tests
timeout 120 python - <<'PY' 2>&1 | tee gpurun_out/asm_first.log
import torch, numpy as np, cspn_amd
from helpers import make_inputs, rel_err
from oracle import cspn2d_oracle
for (B,H,W,sp,norm) in [(1,12,256,False,"8sum"),(2,17,304,True,"8sum"),(1,40,512,True,"8sum_abs"),(1,30,1216,True,"none")]:
    g,h,s = make_inputs(B,H,W,seed=3,sparse=sp)
    if norm == "none":
        g = g.abs() / (g.abs().sum(1, keepdim=True) + 0.25)
    ref = cspn2d_oracle(g,h,s,24,norm)
    for algo in ("fused_cxx","fused"):
        o = cspn_amd.cspn2d_forward(g.cuda(),h.cuda(),None if s is None else s.cuda(),24,norm,algo)
        torch.cuda.synchronize()
        o = o.cpu().numpy()
        bad = np.isnan(o) != np.isnan(ref)
        print(B,H,W,sp,norm,algo,"nan-mismatch",int(bad.sum()),"err",float(np.nanmax(np.abs(o-ref))/np.nanmax(np.abs(ref))), flush=True)
PY
echo "== first contact exit: $?"
if grep -q "fused nan-mismatch 0" gpurun_out/asm_first.log; then
  timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/asm_pytest.log
  for algo in fused fused_cxx; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --algo $algo > gpurun_out/asm_bench_$algo.json 2> gpurun_out/asm_bench_$algo.err
    cat gpurun_out/asm_bench_$algo.json
  done
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload kitti_sparse > gpurun_out/asm_bench_sparse.json 2>/dev/null; cat gpurun_out/asm_bench_sparse.json
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload nyu > gpurun_out/asm_bench_nyu.json 2>/dev/null; cat gpurun_out/asm_bench_nyu.json
fi
