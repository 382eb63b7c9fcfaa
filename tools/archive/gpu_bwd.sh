#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/bench_backward.py --batch 16 --sparse | tee gpurun_out/bwd_b16.json
python tools/bench_backward.py --batch 64 | tee gpurun_out/bwd_b64.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bwd -- python $GRAFT_REPO_ROOT/tools/bench_backward.py --batch 64 --steps 5) > gpurun_out/prof_bwd.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_bwd/*/*.db gpurun_out/prof_bwd.md | head -8 | cut -c1-200
