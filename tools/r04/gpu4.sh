#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python tools/stress_asm.py 2>&1 | tail -9 > $O/stress_repeats.txt
FUZZ_CASES=60 FUZZ_SEED=11 timeout 1500 python tools/fuzz_parity.py 2>&1 | tail -3 > $O/fuzz_parity.txt
timeout 600 python tools/bench_backward.py --batch 64 --steps 20 2>&1 | tail -3 > $O/backward.txt
cat $O/stress_repeats.txt $O/fuzz_parity.txt $O/backward.txt
