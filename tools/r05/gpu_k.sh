#!/bin/bash
# round 5, run K: does the headline loop's time follow its LDS bank conflicts?  Single-variant builds with the cooked-row ring at record strides 160 / 168 / 176 bytes
# (66.9 / 33.7 / 43.1 % conflict cycles in the emulator's bank model), two alternating rounds, parity checked by bench.py.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
for rnd in 1 2; do
  for v in product ring168 ring164w2 ring172w2; do
    if [ $v = product ]; then unset CSPN_AMD_LIB; else export CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_$v.so; fi
    timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $rnd', d['roofline']['device_ms_per_launch'], d['roofline']['device_ms_min'], d['parity_checked']['ok'])" | tee -a gpurun_out/r5k_ringrec.txt
  done
done
