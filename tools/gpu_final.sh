#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3) > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json; cut -c1-400 gpurun_out/bench_default.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fused5 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline) > gpurun_out/rocprof5.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_fused5/*/*.db gpurun_out/prof_fused5.md | head -4 | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc5_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline) > gpurun_out/pmc5_$c.log 2>&1
python tools/rocpd_summary.py gpurun_out/pmc5_$c/*/*.db gpurun_out/pmc5_$c.md | grep -E "SIZE" | cut -c1-160
done
for cfg in "kitti_sparse 32" "nyu 16"; do set -- $cfg; python bench.py --no-cpu-baseline --workload $1 --batch-per-gpu $2 2>&1 | tail -1 > gpurun_out/bench_$1.json; cut -c1-200 gpurun_out/bench_$1.json; done
