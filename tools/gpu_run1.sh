#!/bin/bash
# first GPU pass: ubench + gpu tests + bench + rocprof of the bench
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 120 tools/ubench) > gpurun_out/ubench.log 2>&1
(timeout 1200 python -m pytest tests -x -q -m gpu) > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
(timeout 600 python bench.py --steps 10 --warmup 3) > gpurun_out/bench.log 2>&1
tail -3 gpurun_out/bench.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_stepwise -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline) > gpurun_out/rocprof.log 2>&1
ls -R gpurun_out | head -40
cat gpurun_out/ubench.log
