#!/bin/bash
# round 2, GPU call I: step-phase trace of the loop + rocprofv3 kernel stats + HBM traffic counters of the committed build
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_trace.so timeout 300 python tools/tsw_trace.py gpurun_out/r2i_trace.json > gpurun_out/r2i_trace.log 2>&1
tail -40 gpurun_out/r2i_trace.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2i_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline) > gpurun_out/r2i_prof.log 2>&1
python tools/rocpd_summary.py gpurun_out/r2i_prof/*/*.db gpurun_out/r2i_prof.md | head -8 | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
 (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/r2i_pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --prewarm-s 0 --no-cpu-baseline --pmc-calib) > gpurun_out/r2i_pmc_$c.log 2>&1
 python tools/rocpd_summary.py gpurun_out/r2i_pmc_$c/*/*.db gpurun_out/r2i_pmc_$c.md | grep -E "tsw|elementwise|counter" | cut -c1-200
done
rm -rf gpurun_out/r2i_prof gpurun_out/r2i_pmc_FETCH_SIZE gpurun_out/r2i_pmc_WRITE_SIZE
