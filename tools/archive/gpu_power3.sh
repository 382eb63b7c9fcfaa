#!/bin/bash
# power / clock of the device while a build of the loop runs back to back (rocm-smi samples during a long bench run)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
for name in "$@"; do
  lib=$PWD/cspn_amd/abl/libcspn_$name.so
  [ "$name" = product ] && lib=$PWD/cspn_amd/libcspn_amd.so
  CSPN_AMD_LIB=$lib timeout 120 python bench.py --no-cpu-baseline --no-parity-check --algo fused --prewarm-s 0.5 --steps 20000 --warmup 20 > /tmp/pw_$name.json 2>/dev/null &
  pid=$!
  sleep 4.5
  for i in 1 2 3; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo
    sleep 0.7
  done
  wait $pid
  python -c "import sys,json; d=json.loads([l for l in open('/tmp/pw_$name.json').read().splitlines() if l.startswith('{')][0]); print('$name', d['roofline']['device_ms_per_launch'], d['roofline']['device_ms_min'])"
done
