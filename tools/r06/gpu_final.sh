#!/bin/bash
# round 6, final evidence on the final tree: the GPU suite, smoke, the driver's command (-> profiles/r06_bench_driver.json), and rocprofv3 kernel stats of that same
# command (-> profiles/r06_kernel_stats_driver_cmd.md) and of the headline-only command (-> profiles/r06_kernel_stats_headline_only.md)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r06_final_pytest.txt; cat gpurun_out/r06_final_pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver.json 2> gpurun_out/r06_bench_driver.err; wc -c gpurun_out/r06_bench_driver.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6fin_drv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline) > gpurun_out/r6fin_drv.log 2>&1
python tools/rocpd_summary.py gpurun_out/r6fin_drv/*/*.db gpurun_out/r06_kernel_stats_driver_cmd.md | head -14 | cut -c1-180; rm -rf gpurun_out/r6fin_drv
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6fin_hl -- python $R/bench.py --gpus 1 --steps 300 --warmup 100 --no-cpu-baseline --no-extra-configs) > gpurun_out/r6fin_hl.log 2>&1
python tools/rocpd_summary.py gpurun_out/r6fin_hl/*/*.db gpurun_out/r06_kernel_stats_headline_only.md | head -6 | cut -c1-180; rm -rf gpurun_out/r6fin_hl
grep -h '"metric"' gpurun_out/r6fin_hl.log | tail -1 | cut -c1-400
