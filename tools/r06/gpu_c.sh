#!/bin/bash
# round 6, run C: s_memtime step traces of the 12 x 3 ring: full, without DMA, bare ring
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
for v in ${VARIANTS:-trace tracenodma tracecore}; do
  echo "== $v"
  CSPN_AMD_LIB=$R/cspn_amd/abl/libcspn_t4_$v.so timeout 300 python tools/r06/tsw4_trace.py gpurun_out/r6c_$v.json 2>&1 | grep -v amdgpu.ids
done
