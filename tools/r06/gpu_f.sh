#!/bin/bash
# round 6, run F: HBM fetch volume of the two rings with / without XCD-aware placement (FETCH_SIZE passes)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r6f
HL="--steps 10 --warmup 3 --prewarm-s 0 --no-cpu-baseline --no-extra-configs --no-parity-check --pmc-calib"
pmc() { name=$1; args=$2; shift 2; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $R/${O}_pmc_$name -- python $R/bench.py $args) > ${O}_pmc_$name.log 2>&1; echo "== $name"; python tools/rocpd_summary.py ${O}_pmc_$name/*/*.db ${O}_pmc_$name.md | grep -E "tsw|elementwise" | grep -v "^| kernel" | cut -c1-200; rm -rf ${O}_pmc_$name; }
for pm in 16 17 8 9; do
  pmc fetch_pm$pm "$HL --plan-mode $pm" FETCH_SIZE
done
