#!/bin/bash
# round 3: the per-GPU shapes of a strong-scaling run of BASELINE config 3 (batch 64 over 1 / 2 / 4 / 8 GPUs = 64 / 32 / 16 / 8 images
# per GPU) measured on one GPU, round-3 loop and round-2 loop; config 2 and config 4 lines
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
out=gpurun_out/r3c_shapes.txt; : > $out
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); r=d['roofline']; print('$1', d['config']['B_per_gpu'], d['ms_per_step'], r['device_ms_per_launch'], r['frac'], d['value'])" | tee -a $out; }
for b in 64; do
  timeout 200 python bench.py --batch-per-gpu $b --steps 300 --warmup 20 --no-cpu-baseline --no-parity-check 2>/dev/null | line kitti_v3
  CSPN_TSW_V2=1 timeout 200 python bench.py --batch-per-gpu $b --steps 300 --warmup 20 --no-cpu-baseline --no-parity-check 2>/dev/null | line kitti_v2
done
timeout 200 python bench.py --workload kitti_sparse --batch-per-gpu 32 --steps 300 --warmup 20 --no-cpu-baseline --no-parity-check 2>/dev/null | line c4_v3
CSPN_TSW_V2=1 timeout 200 python bench.py --workload kitti_sparse --batch-per-gpu 32 --steps 300 --warmup 20 --no-cpu-baseline --no-parity-check 2>/dev/null | line c4_v2
timeout 200 python bench.py --workload nyu --batch-per-gpu 16 --steps 300 --warmup 20 --no-cpu-baseline --no-parity-check 2>/dev/null | line c2_v3
CSPN_TSW_V2=1 timeout 200 python bench.py --workload nyu --batch-per-gpu 16 --steps 300 --warmup 20 --no-cpu-baseline --no-parity-check 2>/dev/null | line c2_v2

