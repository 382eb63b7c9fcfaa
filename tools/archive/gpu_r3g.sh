#!/bin/bash
# round 3: fuzz (three seeds) + repeat-stress of the assembly loops after the last kernel changes
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/tests
: > gpurun_out/r3g_fuzz.txt
for seed in 11 12 13; do FUZZ_CASES=40 FUZZ_SEED=$seed timeout 900 python tools/fuzz_parity.py 2>&1 | tail -1 | sed "s/^/seed $seed: /" | tee -a gpurun_out/r3g_fuzz.txt; done
timeout 900 python tools/stress_asm.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3g_stress.txt
