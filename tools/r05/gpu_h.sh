#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
timeout 900 python tools/r05/stress_vol3d.py 2>&1 | tail -6 | tee gpurun_out/r5h_stress_vol3d.txt
timeout 900 python -m pytest tests/test_bench_gpu.py -m gpu -q 2>&1 | tail -3
