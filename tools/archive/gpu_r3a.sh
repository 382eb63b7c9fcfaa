#!/bin/bash
# round 3, first run of the LDS-DMA loop: smoke, 2D parity tests, bench (v3 vs the round-2 loop)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3a_smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r3a_smoke.txt
tail -5 gpurun_out/r3a_smoke.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3a_parity.txt 2>&1; tail -15 gpurun_out/r3a_parity.txt
timeout 300 python bench.py --steps 100 --warmup 20 > gpurun_out/r3a_bench_v3.json 2> gpurun_out/r3a_bench_v3.err; tail -c 900 gpurun_out/r3a_bench_v3.json
CSPN_TSW_V2=1 timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r3a_bench_v2.json 2>/dev/null; tail -c 600 gpurun_out/r3a_bench_v2.json
