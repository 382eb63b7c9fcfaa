#!/bin/bash
# time the ablation builds of the assembly loop (tools/build_abl.sh); results are only timings
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
: > gpurun_out/abl_asm.txt
for name in "$@"; do
  CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_$name.so timeout 200 python bench.py --no-cpu-baseline --algo fused 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['roofline']['device_ms_per_launch'], d['roofline']['device_ms_min'])" | tee -a gpurun_out/abl_asm.txt
done
