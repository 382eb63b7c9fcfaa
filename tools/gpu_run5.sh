#!/bin/bash
mkdir -p gpurun_out
(CSPN_AMD_LIB=$PWD/gpurun_dbg_CHECK.so timeout 600 python tools/stress_check.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/stress_check.log 2>&1
tail -20 gpurun_out/stress_check.log
(timeout 600 python tools/stress_fused.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/stress.log 2>&1
tail -12 gpurun_out/stress.log
