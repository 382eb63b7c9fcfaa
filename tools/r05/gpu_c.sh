#!/bin/bash
# round 5, run C: the 3D kernel with the XCD-aware placement -- full -m gpu suite, A/B against the plain order on the same box, HBM traffic
# counters, kernel stats; the driver's command with the child-process CPU baseline.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r5c
timeout 1500 python -m pytest tests -m gpu -q > ${O}_pytest.log 2>&1; echo "pytest rc $?" >> ${O}_pytest.log; tail -8 ${O}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python tools/r05/ab_vol3d_placement.py > ${O}_ab_vol3d.json 2> ${O}_ab.err; cat ${O}_ab_vol3d.json; tail -2 ${O}_ab.err
T0=$(date +%s.%N)
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > ${O}_bench_driver.json 2> ${O}_bench.err
python -c "import time,sys; print(\"driver command wall s: %.1f\" % (time.time() - float(sys.argv[1])))" $T0 | tee ${O}_bench_driver_wall.txt
python - <<P
import json
d=json.load(open('${O}_bench_driver.json'))
print('headline', d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline']['frac'], d['parity_checked']['ok'])
for k,v in d.get('configs',{}).items():
    print(k, v.get('error') or (v['ms_per_step'], v['roofline']['device_ms_per_launch'], v['roofline']['frac'], v['parity_checked']['ok']))
c=d.get('cpu_baseline'); print('cpu port', c['value'], c['cores'], c.get('port_on_64_threads')); print('refops', json.dumps(c.get('reference_op_sequence'))[:900])
P
tail -3 ${O}_bench.err
V3="--workload vol3d --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/${O}_prof3d -- python $R/bench.py $V3) > ${O}_prof3d.log 2>&1
python tools/rocpd_summary.py ${O}_prof3d/*/*.db ${O}_vol3d_kernel_stats.md | head -5 | cut -c1-200; rm -rf ${O}_prof3d
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/${O}_pmc3d_$c -- python $R/bench.py $V3 --steps 10 --warmup 3) > ${O}_pmc3d_$c.log 2>&1
  python tools/rocpd_summary.py ${O}_pmc3d_$c/*/*.db ${O}_pmc3d_$c.md | grep -E "persistent|distribution|elementwise" | grep -v "^| kernel" | cut -c1-220; rm -rf ${O}_pmc3d_$c
done
