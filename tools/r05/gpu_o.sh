#!/bin/bash
# round 5, run O: every n_iter on the assembly loop (short first pass): parity tests, then timings per iteration count
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
O=gpurun_out/r5o
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "asm or golden or parity_vs_oracle or full_size or benchmarked" > ${O}_pytest.log 2>&1; echo "pytest rc $?" >> ${O}_pytest.log; tail -5 ${O}_pytest.log
timeout 600 python tools/r05/time_niter.py > ${O}_time_niter.jsonl 2>${O}_time.err; cat ${O}_time_niter.jsonl; tail -2 ${O}_time.err
