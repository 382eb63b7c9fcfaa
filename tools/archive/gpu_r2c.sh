#!/bin/bash
# round 2, GPU call C: parity suite on the new register pairing + in-kernel planning, benches, sync + MALL probes
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r2c_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r2c_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2c_bench_driver.json 2> gpurun_out/r2c_bench.err
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2c_bench_default.json 2>> gpurun_out/r2c_bench.err
timeout 300 python bench.py --no-cpu-baseline --workload kitti_sparse --batch-per-gpu 32 > gpurun_out/r2c_bench_c4.json 2>> gpurun_out/r2c_bench.err
timeout 300 python bench.py --no-cpu-baseline --workload nyu --batch-per-gpu 16 > gpurun_out/r2c_bench_c2.json 2>> gpurun_out/r2c_bench.err
timeout 200 python tools/probe_mall3d.py > gpurun_out/r2c_mall3d.txt 2>&1
timeout 100 tools/ubench_gridsync > gpurun_out/r2c_gridsync.txt 2>&1
tail -5 gpurun_out/r2c_pytest.log; cat gpurun_out/r2c_gridsync.txt; cat gpurun_out/r2c_mall3d.txt | grep "^{"
for f in driver default c4 c2; do python -c "import json;d=json.load(open('gpurun_out/r2c_bench_$f.json'));print('$f',d['ms_per_step'],d['roofline']['device_ms_per_launch'],d['roofline']['device_ms_min'],d['roofline']['frac'],d['parity_checked']['ok'])"; done
