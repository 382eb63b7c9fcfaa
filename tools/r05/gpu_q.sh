#!/bin/bash
# round 5, run Q: memory-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the two new legs: 12 iterations (short pass of the assembly loop) and 304 x 1218 (padded path)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r5q
for wl in kitti_n12 kitti_w1218; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/${O}_pmc_${wl}_$c -- python $R/bench.py --workload $wl --steps 10 --warmup 3 --prewarm-s 0 --no-cpu-baseline --no-parity-check --pmc-calib) > ${O}_pmc_${wl}_$c.log 2>&1
    python tools/rocpd_summary.py ${O}_pmc_${wl}_$c/*/*.db ${O}_pmc_${wl}_$c.md | grep -v "^| kernel\|^|---" | cut -c1-200 | head -12; rm -rf ${O}_pmc_${wl}_$c
  done
done
