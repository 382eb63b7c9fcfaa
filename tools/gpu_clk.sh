#!/bin/bash
# sample shader / memory clocks and power while a build of the kernel runs in a loop
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/tests
for name in "$@"; do
  CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_$name.so timeout 60 python bench.py --steps 12000 --warmup 5 --no-cpu-baseline --algo fused > gpurun_out/clk_$name.json 2>/dev/null &
  pid=$!
  sleep 9
  for i in 1 2 3; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" | tr '\n' ' '; echo
    sleep 0.7
  done | sed "s/^/$name: /" | tee -a gpurun_out/clk.txt
  wait $pid
  python -c "import json; d=json.load(open('gpurun_out/clk_$name.json')); print('$name', d['roofline']['device_ms_per_launch'])" | tee -a gpurun_out/clk.txt
done
