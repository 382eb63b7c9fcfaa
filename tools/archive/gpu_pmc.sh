#!/bin/bash
# PMC passes over the bench (each in its own rocprofv3 run, no tracing domains besides kernel-trace)
mkdir -p gpurun_out
export TMPDIR=/tmp
ARGS="${BENCH_ARGS:---steps 5 --warmup 2 --no-cpu-baseline}"
run() { name=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$name -- python $GRAFT_REPO_ROOT/bench.py $ARGS) > gpurun_out/pmc_$name.log 2>&1; python tools/rocpd_summary.py gpurun_out/pmc_$name/*/*.db gpurun_out/pmc_$name.md | grep -E "fused|tsw|step2d|counter" | cut -c1-160; }
run sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
