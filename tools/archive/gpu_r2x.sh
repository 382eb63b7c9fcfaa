#!/bin/bash
# the persistent 3D kernel with two half-width workgroups per CU (-DP3_XG=4) against the product library
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && : > gpurun_out/r2x.txt
for lib in "" cspn_amd/abl/libcspn_xg4.so; do
  export CSPN_AMD_LIB=${lib:+$PWD/$lib}
  [ -z "$lib" ] && unset CSPN_AMD_LIB
  echo "== ${lib:-product}" | tee -a gpurun_out/r2x.txt
  timeout 800 python -m pytest tests/test_gpu_parity.py tests/test_backward3d.py -m gpu -x -q -k "3d" 2>&1 | tail -3 | tee -a gpurun_out/r2x.txt
  timeout 300 python bench.py --workload vol3d --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vol3d bench', d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline']['frac'], d['parity_checked']['ok'])" | tee -a gpurun_out/r2x.txt
  timeout 300 python tools/bench_backward.py --vol3d --steps 5 2>&1 | grep -v amdgpu.ids | cut -c1-130 | tee -a gpurun_out/r2x.txt
done
