#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_p3trace.so timeout 300 python tools/probe_3d_trace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2v.txt
