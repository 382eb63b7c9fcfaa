#!/bin/bash
# round 5, run B: the -m gpu suite with the prenorm contract, the multi-channel 3D backward and the bounded CPU baselines; the driver's
# command timed by the wall clock; kernel stats of the headline-only command; prenorm A/B; 3D multi-channel backward timing.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r5b
timeout 1500 python -m pytest tests -m gpu -q -x > ${O}_pytest.log 2>&1; echo "pytest rc $?" >> ${O}_pytest.log; tail -6 ${O}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T0=$(date +%s.%N)
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > ${O}_bench_driver.json 2> ${O}_bench.err
python -c "import time,sys; print(\"driver command wall s: %.1f\" % (time.time() - float(sys.argv[1])))" $T0 | tee ${O}_bench_driver_wall.txt
python - <<P
import json
d=json.load(open('${O}_bench_driver.json'))
print('headline', d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline']['frac'], d['parity_checked']['ok'])
for k,v in d.get('configs',{}).items():
    print(k, v.get('error') or (v['ms_per_step'], v['roofline']['device_ms_per_launch'], v['roofline']['frac'], v['parity_checked']['ok'], v.get('producer_epilogue_standalone_ms')))
print('cpu', json.dumps(d.get('cpu_baseline'))[:900])
P
tail -3 ${O}_bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/${O}_prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs) > ${O}_prof.log 2>&1
python tools/rocpd_summary.py ${O}_prof/*/*.db ${O}_prof_headline_only.md | head -5 | cut -c1-200
rm -rf ${O}_prof
timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --layout prenorm > ${O}_bench_prenorm.json 2>> ${O}_bench.err
timeout 300 python bench.py --no-cpu-baseline --no-extra-configs > ${O}_bench_planar.json 2>> ${O}_bench.err
timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --layout prenorm --workload kitti_sparse --batch-per-gpu 32 > ${O}_bench_prenorm_c4.json 2>> ${O}_bench.err
timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --layout prenorm --batch-per-gpu 8 > ${O}_bench_prenorm_b8.json 2>> ${O}_bench.err
for f in prenorm planar prenorm_c4 prenorm_b8; do python -c "import json;d=json.load(open('${O}_bench_$f.json'));print('$f',d['ms_per_step'],d['roofline']['device_ms_per_launch'],d['roofline']['frac'],d['parity_checked']['ok'])"; done
timeout 600 python tools/r05/bench_bwd3d_multi.py > ${O}_bwd3d_multi.json 2>> ${O}_bench.err; cat ${O}_bwd3d_multi.json
tail -3 ${O}_bench.err
