#!/bin/bash
# round 6, run E: the 12 x 3 ring as shipped: rocprofv3 kernel stats of the headline-only command, SQ counters (four --pmc passes), HBM traffic (FETCH_SIZE / WRITE_SIZE passes)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r6e
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/${O}_prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs) > ${O}_prof.log 2>&1
python tools/rocpd_summary.py ${O}_prof/*/*.db ${O}_kernel_stats_headline_only.md | head -4 | cut -c1-200; rm -rf ${O}_prof
HL="--steps 10 --warmup 3 --prewarm-s 0 --no-cpu-baseline --no-extra-configs --no-parity-check"
pmc() { name=$1; args=$2; shift 2; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $R/${O}_pmc_$name -- python $R/bench.py $args) > ${O}_pmc_$name.log 2>&1; python tools/rocpd_summary.py ${O}_pmc_$name/*/*.db ${O}_pmc_$name.md | grep -E "tsw|elementwise" | grep -v "^| kernel" | cut -c1-200; rm -rf ${O}_pmc_$name; }
pmc sq1 "$HL" SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU
pmc sq2 "$HL" SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
pmc sq3 "$HL" GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD
pmc sq4 "$HL" SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_SMEM
pmc sq5 "$HL" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM
for c in FETCH_SIZE WRITE_SIZE; do
  pmc hl_$c "$HL --pmc-calib" $c
done
ls gpurun_out | grep r6e | head -30
