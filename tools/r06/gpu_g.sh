#!/bin/bash
# round 6, run G: the guidance head -- kernel stats, SQ counters and HBM bytes of head_raw_kernel (tools/r06/time_head.py), then the driver's command
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r6g
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/${O}_stats -- python $R/tools/r06/time_head.py) > ${O}_stats.log 2>&1
python tools/rocpd_summary.py ${O}_stats/*/*.db ${O}_head_kernel_stats.md | grep -E "head|kernel" | cut -c1-200; rm -rf ${O}_stats
pmc() { name=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $R/${O}_pmc_$name -- python $R/tools/r06/time_head.py) > ${O}_pmc_$name.log 2>&1; echo "== $name"; python tools/rocpd_summary.py ${O}_pmc_$name/*/*.db ${O}_pmc_$name.md | grep -E "head_raw" | cut -c1-240; rm -rf ${O}_pmc_$name; }
pmc sq SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES
pmc sq2 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver.json 2> gpurun_out/r06_bench_driver.err
wc -c gpurun_out/r06_bench_driver.json
