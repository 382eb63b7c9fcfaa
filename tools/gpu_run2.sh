#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -x -q -m gpu) > gpurun_out/pytest_gpu.log 2>&1
tail -15 gpurun_out/pytest_gpu.log
(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline) > gpurun_out/bench_fused.log 2>&1
tail -2 gpurun_out/bench_fused.log
(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload kitti_sparse --batch-per-gpu 32) > gpurun_out/bench_fused_c4.log 2>&1
tail -1 gpurun_out/bench_fused_c4.log
