#!/bin/bash
# round-end evidence for the assembly build: GPU tests, bench lines (configs 2,3,4 + C++ kernel A/B + CPU baseline), kernel trace, HBM PMC
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/f2_pytest.log
timeout 600 python bench.py > gpurun_out/f2_bench_default.json 2> gpurun_out/f2_bench_default.err; cat gpurun_out/f2_bench_default.json
for w in kitti_sparse nyu; do timeout 300 python bench.py --no-cpu-baseline --workload $w > gpurun_out/f2_bench_$w.json 2>/dev/null; done
timeout 300 python bench.py --no-cpu-baseline --workload kitti_sparse --batch-per-gpu 32 > gpurun_out/f2_bench_kitti_sparse_b32.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --workload nyu --batch-per-gpu 16 > gpurun_out/f2_bench_nyu_b16.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --algo fused_cxx > gpurun_out/f2_bench_cxx.json 2>/dev/null
for f in kitti_sparse kitti_sparse_b32 nyu nyu_b16 cxx; do python -c "import json; d=json.load(open('gpurun_out/f2_bench_$f.json')); print('$f', d['config']['algo'], d['config']['B_per_gpu'], d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline']['frac'])"; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/f2_prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline) > gpurun_out/f2_prof.log 2>&1
python tools/rocpd_summary.py gpurun_out/f2_prof/*/*.db gpurun_out/f2_prof.md | head -8 | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
 (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/f2_pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline) > gpurun_out/f2_pmc_$c.log 2>&1
 python tools/rocpd_summary.py gpurun_out/f2_pmc_$c/*/*.db gpurun_out/f2_pmc_$c.md | grep -E "tsw|elementwise|counter" | cut -c1-200
done
