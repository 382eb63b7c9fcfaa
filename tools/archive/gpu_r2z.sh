#!/bin/bash
# how long does a fresh box need before the 2D figure settles?  the driver's command with different clock pre-warm times
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && : > gpurun_out/r2z.txt
for pw in 1 1 1 4 1 8 1; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prewarm-s $pw 2>/dev/null | tail -1 | python -c "import sys,json,time; d=json.loads(sys.stdin.read()); print('prewarm $pw:', d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline']['frac'])" | tee -a gpurun_out/r2z.txt
done
