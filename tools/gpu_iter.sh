#!/bin/bash
# standard iteration: stress + gpu tests + bench(3 configs) + one PMC pass
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 600 python tools/stress_fused.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/stress.log 2>&1; tail -4 gpurun_out/stress.log
(timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4) > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
for cfg in "kitti 64" "kitti_sparse 32" "nyu 16"; do set -- $cfg
(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $1 --batch-per-gpu $2 2>&1 | grep -v amdgpu.ids) > gpurun_out/bench_$1.log 2>&1
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_$1.log").read().strip().splitlines()[-1])
    print("$1 B=$2", d["config"]["algo"], "ms/launch", d["roofline"]["device_ms_per_launch"], "min", d["roofline"]["device_ms_min"], "frac", d["roofline"]["frac"], "value", d["value"])
except Exception as e: print("bench $1 failed", e, open("gpurun_out/bench_$1.log").read()[-500:])
PY
done
if [ -n "$PMC" ]; then
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq1 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline) > gpurun_out/pmc_sq1.log 2>&1
python tools/rocpd_summary.py gpurun_out/pmc_sq1/*/*.db gpurun_out/pmc_sq1.md | grep -E "fused" | cut -c1-150
fi
