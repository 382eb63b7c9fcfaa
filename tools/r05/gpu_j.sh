#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r5j
HL="python $R/bench.py --steps 10 --warmup 3 --prewarm-s 0 --no-cpu-baseline --no-extra-configs --no-parity-check"
pmc() { name=$1; cmd=$2; shift 2; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $R/${O}_pmc_$name -- $cmd) > ${O}_pmc_$name.log 2>&1; python tools/rocpd_summary.py ${O}_pmc_$name/*/*.db ${O}_pmc_$name.md | grep -E "tsw_kernel" | grep -v "^| kernel" | cut -c1-200; rm -rf ${O}_pmc_$name; }
pmc fwd_lds "$HL" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
pmc fwd_lds2 "$HL" SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN
pmc pre_lds "$HL --layout prenorm" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
