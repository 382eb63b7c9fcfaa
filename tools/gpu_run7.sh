#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
# 1. two ranks on the single GPU: exercises the torchrun / nccl / barrier / all_reduce path of bench.py
(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --batch-per-gpu 8 2>&1 | tail -15) > gpurun_out/bench_2rank.log 2>&1
tail -8 gpurun_out/bench_2rank.log
# 2. kernel-trace stats + PMC traffic for the committed fused kernel
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fused2 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline) > gpurun_out/rocprof2.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_fused2/*/*.db gpurun_out/prof_fused2.md | head -6 | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc2_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline) > gpurun_out/pmc2_$c.log 2>&1
python tools/rocpd_summary.py gpurun_out/pmc2_$c/*/*.db gpurun_out/pmc2_$c.md | grep -E "fused.*SIZE" | cut -c1-160
done
# 3. 3D config 5 timing (stepwise 3D path)
python - <<'PY'
import torch, time, sys
sys.path.insert(0, '.')
import cspn_amd
B,D,H,W = 4,32,160,608
gen = torch.Generator(device="cuda").manual_seed(5)
g = torch.rand(B,26,D,H,W,generator=gen,device="cuda"); h = torch.rand(B,1,D,H,W,generator=gen,device="cuda")
for _ in range(2): o = cspn_amd.cspn3d_forward(g,h,None,12,"8sum_abs")
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5): o = cspn_amd.cspn3d_forward(g,h,None,12,"8sum_abs")
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
vox=B*D*H*W
print("3D config5: %.3f ms/forward, %.1f Mvox*iters/s, alg %.1f GB/s (%.3f of 8 TB/s)" % (dt*1e3, vox*12/dt/1e6, vox*112/dt/1e9, vox*112/dt/8e12))
PY
