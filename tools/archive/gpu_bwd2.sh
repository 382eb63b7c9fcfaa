#!/bin/bash
# backward: parity tests + timing, both lane mappings of the final pass
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/bwd2.txt
for rl in 16 64; do
  export CSPN_BWD_FINAL_RL=$rl
  echo "== final pass RL=$rl" | tee -a gpurun_out/bwd2.txt
  timeout 900 python -m pytest tests -m gpu -q -x -k "backward or grad or bwd" 2>&1 | tail -2 | tee -a gpurun_out/bwd2.txt
  python tools/bench_backward.py --batch 64 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a gpurun_out/bwd2.txt
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bwd2_$rl -- python $GRAFT_REPO_ROOT/tools/bench_backward.py --batch 64 --steps 5) > gpurun_out/prof_bwd2.log 2>&1
  python tools/rocpd_summary.py gpurun_out/prof_bwd2_$rl/*/*.db gpurun_out/prof_bwd2_$rl.md | head -5 | cut -c1-200 | tee -a gpurun_out/bwd2.txt
done
