#!/bin/bash
# kernel trace + PMC passes for the assembly build (separate rocprofv3 runs; no tracing domains besides kernel-trace)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_tsw -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 30 --no-cpu-baseline) > gpurun_out/prof_tsw.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_tsw/*/*.db gpurun_out/prof_tsw.md | head -12 | cut -c1-200
BENCH_ARGS="--steps 10 --warmup 3 --no-cpu-baseline" bash tools/gpu_pmc.sh 2>&1 | grep -v "^$" | cut -c1-220
