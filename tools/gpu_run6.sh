#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(CSPN_AMD_LIB=$PWD/gpurun_dbg_CHECK.so timeout 600 python tools/stress_check.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/stress_check.log 2>&1
tail -8 gpurun_out/stress_check.log
(timeout 600 python tools/stress_fused.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/stress.log 2>&1
tail -5 gpurun_out/stress.log
for i in 1 2; do (timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4) > gpurun_out/pytest_gpu_$i.log 2>&1; tail -2 gpurun_out/pytest_gpu_$i.log; done
(timeout 300 python bench.py --steps 20 --warmup 5) > gpurun_out/bench_fused.log 2>&1; tail -1 gpurun_out/bench_fused.log
(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload kitti_sparse --batch-per-gpu 32) > gpurun_out/bench_fused_c4.log 2>&1; tail -1 gpurun_out/bench_fused_c4.log
(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload nyu --batch-per-gpu 16) > gpurun_out/bench_fused_c2.log 2>&1; tail -1 gpurun_out/bench_fused_c2.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fused -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline) > gpurun_out/rocprof.log 2>&1
ls gpurun_out/prof_fused/*
