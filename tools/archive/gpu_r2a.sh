#!/bin/bash
# round 2, GPU call A: full -m gpu suite, the driver's bench command, the 2-rank launch path, grid-sync microbenchmark
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2a_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r2a_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a_bench_driver.json 2> gpurun_out/r2a_bench_driver.err
timeout 300 python bench.py --steps 20 --warmup 5 --prewarm-s 0 --no-cpu-baseline > gpurun_out/r2a_bench_driver_noprewarm.json 2>> gpurun_out/r2a_bench_driver.err
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2a_bench_default.json 2>> gpurun_out/r2a_bench_driver.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2a_bench_2rank.json 2> gpurun_out/r2a_bench_2rank.err
timeout 120 tools/ubench_gridsync > gpurun_out/r2a_gridsync.txt 2>&1
tail -3 gpurun_out/r2a_pytest.log; cat gpurun_out/r2a_bench_driver.json | head -c 600; echo; cat gpurun_out/r2a_gridsync.txt
