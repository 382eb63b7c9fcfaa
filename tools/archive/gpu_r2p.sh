#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_backward.py tests/test_gpu_parity.py -m gpu -x -q -k "backward or training or deterministic" 2>&1 | tail -4
python tools/bench_backward.py --batch 64 | tee gpurun_out/r2p_bwd_b64.json
CSPN_BWD_FINAL1=1 python tools/bench_backward.py --batch 64 | tee gpurun_out/r2p_bwd_b64_final1.json
python tools/bench_backward.py --batch 16 --sparse | tee gpurun_out/r2p_bwd_b16.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2p_prof_bwd -- python $GRAFT_REPO_ROOT/tools/bench_backward.py --batch 64 --steps 5) > gpurun_out/r2p_prof_bwd.log 2>&1
python tools/rocpd_summary.py gpurun_out/r2p_prof_bwd/*/*.db gpurun_out/r2p_prof_bwd.md | head -8 | cut -c1-200
rm -rf gpurun_out/r2p_prof_bwd
