#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "3d or paddle" 2>&1 | tail -4
timeout 600 python bench.py --workload vol3d > gpurun_out/bench_vol3d.json 2> gpurun_out/bench_vol3d.err; cat gpurun_out/bench_vol3d.json; tail -3 gpurun_out/bench_vol3d.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_vol3d -- python $GRAFT_REPO_ROOT/bench.py --workload vol3d --no-cpu-baseline) > gpurun_out/prof_vol3d.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_vol3d/*/*.db gpurun_out/prof_vol3d.md | head -6 | cut -c1-200
