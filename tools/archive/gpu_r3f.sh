#!/bin/bash
# round 3: 2D backward final pass with the rows above / below shared between lanes (ds_bpermute) -- tests, timing, kernel stats
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
for nb in 2 3; do
export CSPN_BWD_FINAL_NB=$nb
echo "== NB $nb"
timeout 900 python -m pytest tests/test_backward.py -m gpu -x -q 2>&1 | tail -1
timeout 300 python tools/bench_backward.py --batch 64 2>/dev/null | tail -1 | cut -c1-330
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bwd -- python $GRAFT_REPO_ROOT/tools/bench_backward.py --batch 64) > gpurun_out/prof_bwd.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_bwd/*/*.db gpurun_out/r3f_backward_kernel_stats_nb$nb.md | head -5 | cut -c1-200
rm -rf gpurun_out/prof_bwd
done
