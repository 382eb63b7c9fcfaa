#!/bin/bash
# time single-variant builds of the round-3 loop (tools/build_abl3.sh); ablations give wrong results: timings only
# usage: gpu_abl3.sh OUTFILE NAME...
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
out=gpurun_out/$1; shift
: > $out
for name in "$@"; do
  lib=$PWD/cspn_amd/abl/libcspn_$name.so
  [ "$name" = product ] && lib=$PWD/cspn_amd/libcspn_amd.so
  CSPN_AMD_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-parity-check --algo fused --prewarm-s 0.7 --steps 300 --warmup 20 2>/dev/null \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('$name', d['roofline']['device_ms_per_launch'], d['roofline']['device_ms_min'])" | tee -a $out
done
