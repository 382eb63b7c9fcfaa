#!/bin/bash
# elastic (barrier-free) 2D loop: single-variant builds, parity through bench.py's check + timing
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && : > gpurun_out/r2y.txt
for name in "$@"; do
  CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_$name.so timeout 300 python bench.py --no-cpu-baseline --algo fused --prewarm-s 0.7 --steps 300 --warmup 20 2>gpurun_out/r2y_$name.err \
    | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('$name', d['roofline']['device_ms_per_launch'], d['roofline']['device_ms_min'], d['roofline']['frac'], d['parity_checked'])" | tee -a gpurun_out/r2y.txt
  tail -2 gpurun_out/r2y_$name.err | grep -v amdgpu.ids
done
