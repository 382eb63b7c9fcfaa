#!/bin/bash
# round 2 evidence run: full -m gpu suite, every bench line, rocprofv3 kernel stats (same commands), HBM traffic counters
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
O=gpurun_out/r2f
timeout 1800 python -m pytest tests -m gpu -x -q > ${O}_pytest.log 2>&1; echo "pytest rc $?" >> ${O}_pytest.log; tail -3 ${O}_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > ${O}_bench_driver.json 2> ${O}_bench.err
timeout 300 python bench.py > ${O}_bench_default.json 2>> ${O}_bench.err
timeout 300 python bench.py --no-cpu-baseline --workload kitti_sparse --batch-per-gpu 32 > ${O}_bench_c4.json 2>> ${O}_bench.err
timeout 300 python bench.py --no-cpu-baseline --workload nyu --batch-per-gpu 16 > ${O}_bench_c2.json 2>> ${O}_bench.err
timeout 300 python bench.py --no-cpu-baseline --algo fused_cxx > ${O}_bench_cxx.json 2>> ${O}_bench.err
timeout 300 python bench.py --workload vol3d > ${O}_bench_vol3d.json 2>> ${O}_bench.err
timeout 300 python bench.py --workload vol3d --algo stepwise --no-cpu-baseline > ${O}_bench_vol3d_stepwise.json 2>> ${O}_bench.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2> ${O}_bench_2rank.err | grep "^{" > ${O}_bench_2rank.json
for f in driver default c4 c2 cxx vol3d vol3d_stepwise 2rank; do python -c "import json;d=json.load(open('${O}_bench_$f.json'));print('$f',d['n_gpus'],d['ms_per_step'],d['roofline']['device_ms_per_launch'],d['roofline']['frac'],d['roofline'].get('whole_forward_frac'),d['parity_checked']['ok'])"; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/${O}_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline) > ${O}_prof.log 2>&1
python tools/rocpd_summary.py ${O}_prof/*/*.db ${O}_prof.md | head -6 | cut -c1-180
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/${O}_prof3d -- python $GRAFT_REPO_ROOT/bench.py --workload vol3d --steps 20 --warmup 5 --no-cpu-baseline) > ${O}_prof3d.log 2>&1
python tools/rocpd_summary.py ${O}_prof3d/*/*.db ${O}_prof3d.md | head -6 | cut -c1-180
for c in FETCH_SIZE WRITE_SIZE; do
 (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/${O}_pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --prewarm-s 0 --no-cpu-baseline --pmc-calib) > ${O}_pmc_$c.log 2>&1
 python tools/rocpd_summary.py ${O}_pmc_$c/*/*.db ${O}_pmc_$c.md | grep -E "tsw|elementwise|counter" | cut -c1-180
 (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/${O}_pmc3d_$c -- python $GRAFT_REPO_ROOT/bench.py --workload vol3d --steps 5 --warmup 2 --no-cpu-baseline --no-parity-check) > ${O}_pmc3d_$c.log 2>&1
 python tools/rocpd_summary.py ${O}_pmc3d_$c/*/*.db ${O}_pmc3d_$c.md | grep -E "persistent|counter" | cut -c1-180
done
python tools/bench_aux.py > ${O}_aux.json 2>/dev/null; tail -3 ${O}_aux.json | cut -c1-300
rm -rf ${O}_prof ${O}_prof3d ${O}_pmc_FETCH_SIZE ${O}_pmc_WRITE_SIZE ${O}_pmc3d_FETCH_SIZE ${O}_pmc3d_WRITE_SIZE
