#!/bin/bash
# A/B: staggered cooking (waves 0..3 / 4..7 cook in different steps) vs the committed loop
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_gpu.py -m gpu -x -q -k "not 3d and not vol3d and not capture and not threads" 2>&1 | tail -3
for i in 1 2 3; do
  for lib in stag base; do
    if [ $lib = base ]; then export CSPN_AMD_LIB=$PWD/cspn_amd/abl/base/libcspn_amd.so; else unset CSPN_AMD_LIB; fi
    python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline']['frac'], d['parity_checked']['ok'])"
  done
done | tee $O/stagger_ab.txt
for b in 32 8; do for lib in stag base; do
    if [ $lib = base ]; then export CSPN_AMD_LIB=$PWD/cspn_amd/abl/base/libcspn_amd.so; else unset CSPN_AMD_LIB; fi
    python bench.py --steps 200 --warmup 30 --no-cpu-baseline --batch-per-gpu $b 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib B$b', d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline']['frac'])"
done; done | tee -a $O/stagger_ab.txt
