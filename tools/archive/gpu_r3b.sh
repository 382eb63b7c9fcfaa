#!/bin/bash
# round 3: the whole GPU suite with the round-3 loop as the product + bench lines (v3, v2 for comparison)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r3b_pytest.txt 2>&1; tail -6 gpurun_out/r3b_pytest.txt
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3b_bench_driver.json 2> gpurun_out/r3b_bench.err; tail -c 700 gpurun_out/r3b_bench_driver.json | head -c 400; echo
