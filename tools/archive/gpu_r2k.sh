#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
for d in 0 3 2 5 0; do
CSPN_TSW_DESYNC=$d timeout 200 python bench.py --no-cpu-baseline --no-parity-check --prewarm-s 0.7 --steps 300 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('desync $d', d['roofline']['device_ms_per_launch'], d['roofline']['device_ms_min'])" | tee -a gpurun_out/r2k_desync.txt
done
