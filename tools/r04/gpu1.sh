#!/bin/bash
# r04 first GPU run: the linear plan + band-switching retirement on hardware
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_gpu.py -m gpu -x -q -k "not 3d and not vol3d" 2>&1 | tail -15 > gpurun_out/r04a/pytest_2d.txt
for m in 0 2 1 0 2; do
  timeout 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --plan-mode $m 2>&1 | tail -1 > gpurun_out/r04a/bench_mode${m}_$RANDOM.json
done
for b in 32 16 8; do
  for m in 0 2; do
    timeout 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --batch-per-gpu $b --plan-mode $m 2>&1 | tail -1 > gpurun_out/r04a/bench_b${b}_mode${m}.json
  done
done
for w in kitti_sparse nyu; do
  bb=32; [ $w = nyu ] && bb=16
  for m in 0 2; do
    timeout 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --workload $w --batch-per-gpu $bb --plan-mode $m 2>&1 | tail -1 > gpurun_out/r04a/bench_${w}_mode${m}.json
  done
done
cat gpurun_out/r04a/pytest_2d.txt
grep -h -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"plan_mode": [0-9]\|"B_per_gpu": [0-9]*\|"device_ms_per_launch": [0-9.]*' gpurun_out/r04a/bench_*.json | paste - - - - - 2>/dev/null | head -40
