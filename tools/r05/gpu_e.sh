#!/bin/bash
# round 5, runs D / E: the 3D kernel's row assignment (row_of): full -m gpu suite, then A/B of the product (conflict-free 16-byte LDS reads, plane pitch
# 12) against variant builds (a wave = one z plane; boundary rows first; the new assignment at plane pitch 10) -- separate processes, alternating, same box.
# Then the XCD hand-off micro-benchmark and the driver's command.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r5e
timeout 1500 python -m pytest tests -m gpu -q > ${O}_pytest.log 2>&1; echo "pytest rc $?" >> ${O}_pytest.log; tail -4 ${O}_pytest.log
V3="--workload vol3d --steps 60 --warmup 20 --no-cpu-baseline"
for rnd in 1 2; do
  for v in product rowsplain boundary cflyp10; do
    if [ $v = product ]; then unset CSPN_AMD_LIB; else export CSPN_AMD_LIB=$R/cspn_amd/abl/libcspn_$v.so; fi
    timeout 300 python bench.py $V3 > ${O}_vol3d_${v}_$rnd.json 2>> ${O}_bench.err
    python -c "import json;d=json.load(open('${O}_vol3d_${v}_$rnd.json'));print('$v $rnd',d['ms_per_step'],d['roofline']['device_ms_per_launch'],d['roofline']['device_ms_min'],d['roofline']['frac'],d['parity_checked']['ok'], d['parity_checked']['oracle_full_volume']['max_rel_err'])"
  done
done
unset CSPN_AMD_LIB
./tools/r05/ubench_xcd_rtt > ${O}_ubench_xcd_rtt.txt 2>&1; cat ${O}_ubench_xcd_rtt.txt
timeout 300 python tools/r05/bench_bwd3d_multi.py > ${O}_bwd3d_multi.json 2>> ${O}_bench.err; cat ${O}_bwd3d_multi.json
timeout 300 python tools/bench_backward.py --vol3d > ${O}_bwd3d.json 2>> ${O}_bench.err; cat ${O}_bwd3d.json
T0=$(date +%s.%N)
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > ${O}_bench_driver.json 2>> ${O}_bench.err
python -c "import time,sys; print(\"driver command wall s: %.1f\" % (time.time() - float(sys.argv[1])))" $T0 | tee ${O}_bench_driver_wall.txt
python - <<P
import json
d=json.load(open('${O}_bench_driver.json'))
print('headline', d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline']['frac'], d['parity_checked']['ok'])
for k,v in d.get('configs',{}).items():
    print(k, v.get('error') or (v['ms_per_step'], v['roofline']['device_ms_per_launch'], v['roofline']['frac'], v['parity_checked']['ok']), json.dumps(v.get('cpu_reference_op_sequence'))[:300] if 'cpu_reference_op_sequence' in v else '')
P
tail -3 ${O}_bench.err
