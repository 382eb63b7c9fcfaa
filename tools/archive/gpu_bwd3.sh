#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_backward3d.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/bwd3d.txt
python tools/bench_backward.py --vol3d --steps 5 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/bwd3d.txt
CSPN_3D_BWD_STEPWISE=1 python tools/bench_backward.py --vol3d --steps 5 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/bwd3d.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bwd3 -- python $GRAFT_REPO_ROOT/tools/bench_backward.py --vol3d --steps 3) > gpurun_out/prof_bwd3.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_bwd3/*/*.db gpurun_out/prof_bwd3.md | head -8 | cut -c1-200 | tee -a gpurun_out/bwd3d.txt
