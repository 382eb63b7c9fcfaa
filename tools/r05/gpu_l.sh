#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
for w in kitti_n12 kitti_w1218; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --workload $w --steps 50 --warmup 10 > gpurun_out/r5l_bench_$w.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/r5l_bench_$w.json'));print('$w',d['config']['algo'],d['ms_per_step'],d['roofline']['device_ms_per_launch'],d['roofline']['frac'],d['parity_checked']['ok'])"
done
