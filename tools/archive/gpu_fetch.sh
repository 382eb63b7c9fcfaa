#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for c in FETCH_SIZE; do
 (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/f3_pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline) > gpurun_out/f3_pmc_$c.log 2>&1
 python tools/rocpd_summary.py gpurun_out/f3_pmc_$c/*/*.db gpurun_out/f3_pmc_$c.md | grep -E "tsw|counter" | cut -c1-200
done
