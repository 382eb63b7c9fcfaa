#!/bin/bash
# r04 evidence run: whole GPU suite, smoke, the driver's bench command, rocprofv3 kernel stats of the same command, PMC passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
export PYTHONPATH=$PWD TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/pytest_gpu_tail.txt 2>&1
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 > $O/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > $O/bench_driver.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_driver_2.json
timeout 300 python bench.py --workload kitti_sparse --batch-per-gpu 32 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_c4.json
timeout 300 python bench.py --workload nyu --batch-per-gpu 16 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_c2.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline) > $O/prof.log 2>&1
python tools/rocpd_summary.py $O/prof/*/*.db $O/kernel_stats_driver_cmd.md | head -8 | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/$O/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --pmc-calib --prewarm-s 0.2) > $O/pmc_$c.log 2>&1
  python tools/rocpd_summary.py $O/pmc_$c/*/*.db $O/pmc_$c.md | grep -E "tsw|elementwise|counter" | cut -c1-200
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/$O/pmc3d_$c -- python $GRAFT_REPO_ROOT/bench.py --workload vol3d --batch-per-gpu 4 --steps 5 --warmup 2 --no-cpu-baseline --prewarm-s 0.2) > $O/pmc3d_$c.log 2>&1
  python tools/rocpd_summary.py $O/pmc3d_$c/*/*.db $O/pmc3d_$c.md | grep -E "persistent|counter" | cut -c1-200
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof3d -- python $GRAFT_REPO_ROOT/bench.py --workload vol3d --batch-per-gpu 4 --steps 50 --warmup 10 --no-cpu-baseline) > $O/prof3d.log 2>&1
python tools/rocpd_summary.py $O/prof3d/*/*.db $O/vol3d_kernel_stats.md | head -6 | cut -c1-200
rm -rf $O/prof $O/prof3d $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc3d_FETCH_SIZE $O/pmc3d_WRITE_SIZE
cat $O/pytest_gpu_tail.txt $O/smoke.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('value'))
PY
done
