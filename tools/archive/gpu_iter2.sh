#!/bin/bash
# standard iteration for the assembly build: GPU test-suite, then bench lines for the three 2D configs (+ the C++ kernel for A/B)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/tests
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/it2_pytest.log
for w in kitti kitti_sparse nyu; do
  timeout 300 python bench.py --no-cpu-baseline --workload $w > gpurun_out/it2_bench_$w.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/it2_bench_$w.json')); print('$w', d['config']['algo'], d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline']['frac'])"
done
timeout 300 python bench.py --no-cpu-baseline --algo fused_cxx > gpurun_out/it2_bench_cxx.json 2>/dev/null
python -c "import json; d=json.load(open('gpurun_out/it2_bench_cxx.json')); print('kitti', d['config']['algo'], d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline']['frac'])"
