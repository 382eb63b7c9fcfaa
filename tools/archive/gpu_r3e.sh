#!/bin/bash
# round 3: 3D persistent kernel with parked gate quads (LDS-DMA) -- 3D tests, fuzz, bench
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/tests
timeout 900 python -m pytest tests -m gpu -q -x -k "3d or persistent or vol or paddle or affinity" 2>&1 | tail -5
FUZZ_CASES=25 FUZZ_SEED=7 timeout 600 python tools/fuzz_parity.py 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --workload vol3d --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('vol3d', d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline'].get('whole_forward_frac'), d['parity_checked']['ok'])"; done
