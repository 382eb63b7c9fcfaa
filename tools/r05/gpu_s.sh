#!/bin/bash
# round 5, run S: the checkpointed ring backward for n_iter = 4, 8 .. 20: tests, then its time at KITTI x 64 per iteration count
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
O=gpurun_out/r5s
timeout 900 python -m pytest tests/test_backward.py tests/test_dropin_host.py -m gpu -q -x > ${O}_pytest.log 2>&1; echo "pytest rc $?" >> ${O}_pytest.log; tail -5 ${O}_pytest.log
for n in 24 12 4 20; do timeout 300 python tools/bench_backward.py --batch 64 --steps 20 --n-iter $n 2>/dev/null | tail -1 | cut -c1-330; done | tee ${O}_bwd_niter.jsonl
for seed in 21; do FUZZ_CASES=30 FUZZ_SEED=$seed timeout 900 python tools/fuzz_parity.py 2>&1 | tail -1; done
