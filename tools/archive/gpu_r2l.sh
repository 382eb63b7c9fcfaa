#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "3d or paddle" > gpurun_out/r2l_pytest3d.log 2>&1
echo "rc $?" >> gpurun_out/r2l_pytest3d.log
tail -15 gpurun_out/r2l_pytest3d.log
timeout 300 python bench.py --workload vol3d --no-cpu-baseline > gpurun_out/r2l_vol3d.json 2> gpurun_out/r2l_vol3d.err; tail -3 gpurun_out/r2l_vol3d.err
timeout 300 python bench.py --workload vol3d --no-cpu-baseline --algo stepwise > gpurun_out/r2l_vol3d_stepwise.json 2>> gpurun_out/r2l_vol3d.err
for f in vol3d vol3d_stepwise; do python -c "import json;d=json.load(open('gpurun_out/r2l_$f.json'));print('$f',d['ms_per_step'],d['roofline']['device_ms_per_launch'],d['roofline']['whole_forward_frac'],d['parity_checked'])"; done
timeout 600 python -m pytest tests/test_train_utils.py -m gpu -x -q 2>&1 | tail -3
