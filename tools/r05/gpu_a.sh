#!/bin/bash
# round 5, run A: the -m gpu suite, the driver's bench command (now with every BASELINE config + backward in the ONE line),
# rocprofv3 kernel stats of the same command, SQ counters of the shipped headline kernel, HBM traffic of configs 4 / 2 / 3-share.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r5a
timeout 1500 python -m pytest tests -m gpu -q -x > ${O}_pytest.log 2>&1; echo "pytest rc $?" >> ${O}_pytest.log; tail -4 ${O}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T0=$(date +%s.%N)
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > ${O}_bench_driver.json 2> ${O}_bench.err
python -c "import time,sys; print(\"driver command wall s: %.1f\" % (time.time() - float(sys.argv[1])))" $T0 | tee ${O}_bench_driver_wall.txt
python - <<P
import json
d=json.load(open('${O}_bench_driver.json'))
print('headline', d['ms_per_step'], d['roofline']['device_ms_per_launch'], d['roofline']['frac'], d['parity_checked']['ok'])
for k,v in d.get('configs',{}).items():
    print(k, v.get('error') or (v['ms_per_step'], v['roofline']['device_ms_per_launch'], v['roofline']['frac'], v['parity_checked']['ok']))
print('cpu', json.dumps(d.get('cpu_baseline'))[:600])
P
tail -5 ${O}_bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/${O}_prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline) > ${O}_prof.log 2>&1
python tools/rocpd_summary.py ${O}_prof/*/*.db ${O}_prof.md | head -14 | cut -c1-200
HL="--steps 10 --warmup 3 --prewarm-s 0 --no-cpu-baseline --no-extra-configs --no-parity-check"
pmc() { name=$1; args=$2; shift 2; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $R/${O}_pmc_$name -- python $R/bench.py $args) > ${O}_pmc_$name.log 2>&1; python tools/rocpd_summary.py ${O}_pmc_$name/*/*.db ${O}_pmc_$name.md | grep -E "tsw|elementwise" | grep -v "^| kernel" | cut -c1-200; rm -rf ${O}_pmc_$name; }
pmc sq1 "$HL" SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU
pmc sq2 "$HL" SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
pmc sq3 "$HL" GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD
pmc sq4 "$HL" SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_SMEM
for c in FETCH_SIZE WRITE_SIZE; do
  pmc c4_$c "$HL --pmc-calib --workload kitti_sparse --batch-per-gpu 32" $c
  pmc c2_$c "$HL --pmc-calib --workload nyu --batch-per-gpu 16" $c
  pmc b8_$c "$HL --pmc-calib --batch-per-gpu 8" $c
done
rm -rf ${O}_prof
ls -la gpurun_out | grep r5a | head -40
