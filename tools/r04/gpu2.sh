#!/bin/bash
# r04 second GPU run: 3D (spill fixes, status sequence numbers, shared-gate channels) + the dist tests + vol3d bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_backward3d.py tests/test_dist_gpu.py -m gpu -x -q -k "3d or two_ranks or bench" 2>&1 | tail -15 > gpurun_out/r04c/pytest_3d_dist.txt
cat gpurun_out/r04c/pytest_3d_dist.txt
timeout 600 python tools/r04/bench_multi3d.py gpurun_out/r04c/multi3d.json 2>&1 | tail -12
for i in 1 2; do timeout 300 python bench.py --workload vol3d --batch-per-gpu 4 --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04c/bench_vol3d_$i.json; done
grep -h -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*' gpurun_out/r04c/bench_vol3d_*.json | paste - - 
