#!/bin/bash
# round 3: memory-side traffic of the three backward kernels (FETCH_SIZE / WRITE_SIZE, separate passes)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
 (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_bwd_$c -- python $R/tools/bench_backward.py --batch 64 --steps 3) > gpurun_out/pmc_bwd_$c.log 2>&1
 python tools/rocpd_summary.py gpurun_out/pmc_bwd_$c/*/*.db gpurun_out/r3j_pmc_bwd_$c.md | grep -E "final_ck|tsw_kernel|counter" | cut -c1-170
 rm -rf gpurun_out/pmc_bwd_$c
done
