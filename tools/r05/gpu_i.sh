#!/bin/bash
# round 5, run I: SQ counters of the 2D backward's kernels and of the persistent 3D kernel (what are they waiting for?)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r5i
pmc() { name=$1; cmd=$2; shift 2; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $R/${O}_pmc_$name -- $cmd) > ${O}_pmc_$name.log 2>&1; python tools/rocpd_summary.py ${O}_pmc_$name/*/*.db ${O}_pmc_$name.md | grep -E "final_mx|persistent|tsw_kernel<0, 0, 0, 1>|tsw_kernel<3" | grep -v "^| kernel" | cut -c1-200; rm -rf ${O}_pmc_$name; }
BW="python $R/tools/bench_backward.py --batch 64 --steps 5"
V3="python $R/bench.py --workload vol3d --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check"
for w in bw v3; do
  if [ $w = bw ]; then C="$BW"; else C="$V3"; fi
  pmc ${w}_sq1 "$C" SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU
  pmc ${w}_sq2 "$C" SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
  pmc ${w}_sq3 "$C" GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD
  pmc ${w}_sq4 "$C" SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE
done
