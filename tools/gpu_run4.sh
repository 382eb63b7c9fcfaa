#!/bin/bash
mkdir -p gpurun_out
for v in SYNCTHREADS FENCE_BARRIER DEFAULT; do
  if [ $v = DEFAULT ]; then unset CSPN_AMD_LIB; else export CSPN_AMD_LIB=$PWD/gpurun_dbg_$v.so; fi
  for i in 1 2 3; do
    (timeout 300 python tools/stress_fused.py 2>&1 | grep -v amdgpu.ids | tail -12) > gpurun_out/stress_${v}_$i.log 2>&1
    echo "== $v run $i"; tail -6 gpurun_out/stress_${v}_$i.log
  done
done
