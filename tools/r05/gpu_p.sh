#!/bin/bash
# round 5, run P: W % 4 != 0 on the padded fused path: parity tests, then the bench leg that priced the old path (304 x 1218 x 64)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
O=gpurun_out/r5p
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_gpu.py -m gpu -q -x -k "padded or parity_vs_oracle or fuzz or golden or noncontig" > ${O}_pytest.log 2>&1; echo "pytest rc $?" >> ${O}_pytest.log; tail -5 ${O}_pytest.log
timeout 300 python bench.py --workload kitti_w1218 --steps 30 --warmup 5 --no-cpu-baseline > ${O}_bench_w1218.json 2> ${O}_bench.err; python -c "import json;d=json.load(open('${O}_bench_w1218.json'));print('w1218',d['ms_per_step'],d['roofline']['device_ms_per_launch'],d['roofline']['frac'],d['parity_checked']['ok'],d['config'].get('algo'))"
tail -2 ${O}_bench.err
