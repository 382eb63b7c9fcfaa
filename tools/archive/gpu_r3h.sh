#!/bin/bash
# round 3: 2D backward from checkpoints (every 4th level kept, the rest recomputed by bwd_final_ck_kernel) -- tests, timing, kernel stats
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
for rows in ${CK_ROWS_LIST:-48}; do
export CSPN_BWD_CK_ROWS=$rows
echo "== region rows $rows"
timeout 900 python -m pytest tests/test_backward.py tests/test_dropin_host.py -m gpu -x -q 2>&1 | tail -${TAILN:-3}
timeout 300 python tools/bench_backward.py --batch 64 2>/dev/null | tail -1 | cut -c1-330
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bwd -- python $GRAFT_REPO_ROOT/tools/bench_backward.py --batch 64) > gpurun_out/prof_bwd.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_bwd/*/*.db gpurun_out/r3h_backward_kernel_stats_rows$rows.md | head -5 | cut -c1-200
rm -rf gpurun_out/prof_bwd
done
