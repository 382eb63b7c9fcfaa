#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export PYTHONPATH=$PWD
for rnd in 1 2; do
for v in product rows64; do
  if [ $v = product ]; then unset CSPN_AMD_LIB; else export CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_$v.so; fi
  timeout 300 python tools/bench_backward.py --batch 64 --steps 30 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v $rnd',d['ms_per_call'])"
done
done
unset CSPN_AMD_LIB
CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_rows64.so timeout 600 python -m pytest tests/test_backward.py -m gpu -q 2>&1 | tail -2
