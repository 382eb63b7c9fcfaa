#!/bin/bash
# round 5 evidence run on the final tree: the full -m gpu suite, smoke, the driver's command (wall-timed), rocprofv3 kernel stats of the same command and of the
# headline-only command, the fuzz tool on two more seeds, the 2-rank launch path.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r5z
timeout 1500 python -m pytest tests -m gpu -q > ${O}_pytest.log 2>&1; echo "pytest rc $?" >> ${O}_pytest.log; tail -4 ${O}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee ${O}_smoke.txt
T0=$(date +%s.%N)
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > ${O}_bench_driver.json 2> ${O}_bench.err
python -c "import time,sys; print(\"driver command wall s: %.1f\" % (time.time() - float(sys.argv[1])))" $T0 | tee ${O}_bench_driver_wall.txt
python - <<P
import json
d=json.load(open('${O}_bench_driver.json'))
print('headline', d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline']['frac'], d['parity_checked']['ok'])
for k,v in d.get('configs',{}).items():
    print(k, v.get('error') or (v['ms_per_step'], v['roofline']['device_ms_per_launch'], v['roofline']['frac'], v['parity_checked']['ok']))
P
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/${O}_prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline) > ${O}_prof.log 2>&1
python tools/rocpd_summary.py ${O}_prof/*/*.db ${O}_kernel_stats_driver_cmd.md | head -14 | cut -c1-200; rm -rf ${O}_prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/${O}_prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs) > ${O}_prof.log 2>&1
python tools/rocpd_summary.py ${O}_prof/*/*.db ${O}_kernel_stats_headline_only.md | head -4 | cut -c1-200; rm -rf ${O}_prof
timeout 300 python bench.py --workload kitti_n12 --steps 50 --warmup 10 --no-cpu-baseline > ${O}_bench_n12.json 2>> ${O}_bench.err; python -c "import json;d=json.load(open('${O}_bench_n12.json'));print('n12',d['ms_per_step'],d['roofline']['device_ms_per_launch'],d['roofline']['frac'],d['parity_checked']['ok'])"
for seed in 11 12; do FUZZ_CASES=40 FUZZ_SEED=$seed timeout 900 python tools/fuzz_parity.py 2>&1 | tail -1; done | tee ${O}_fuzz.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 2> ${O}_bench_2rank.err | grep "^{" > ${O}_bench_2rank.json
python -c "import json;d=json.load(open('${O}_bench_2rank.json'));print('2rank',d['n_gpus'],d['ms_per_step'],d['roofline']['frac'],d.get('strong',{}).get('roofline_frac_per_gpu'),d['parity_checked']['ok'],d.get('notes'))"
tail -2 ${O}_bench.err
