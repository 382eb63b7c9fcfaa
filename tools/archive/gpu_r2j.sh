#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
bash tools/gpu_abl_r2.sh r2j_abl.txt base nopw rw08 rw025 late07 base
CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_trace.so timeout 300 python tools/tsw_trace.py gpurun_out/r2j_trace.json > gpurun_out/r2j_trace.log 2>&1
python -c "
import json
d=json.load(open('gpurun_out/r2j_trace.json'))
print(d['forward_ms_instrumented'], d['mean_cycles_per_step_incl_flush'])
print('%-12s'%'kind', d['phases'])
for k,v in d['kinds'].items(): print('%-12s'%k, v['cycles'], v['total'])
"
