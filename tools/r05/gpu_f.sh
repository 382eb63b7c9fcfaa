#!/bin/bash
# round 5, run F: the 2D backward's final pass with mixed pixel pairs: backward tests + fuzz, then A/B against the round-3 final pass (variant build).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r5f
timeout 900 python -m pytest tests/test_backward.py tests/test_fuzz_gpu.py tests/test_dropin_host.py -m gpu -q > ${O}_pytest_bwd.log 2>&1; echo "pytest rc $?" >> ${O}_pytest_bwd.log; tail -5 ${O}_pytest_bwd.log
for rnd in 1 2; do
  for v in product finalck; do
    if [ $v = product ]; then unset CSPN_AMD_LIB; else export CSPN_AMD_LIB=$R/cspn_amd/abl/libcspn_$v.so; fi
    timeout 300 python tools/bench_backward.py --batch 64 --steps 30 > ${O}_bwd_${v}_$rnd.json 2>> ${O}_bench.err
    python -c "import json;d=json.loads(open('${O}_bwd_${v}_$rnd.json').read().strip().splitlines()[-1]);print('$v $rnd',d['ms_per_call'],d['roofline_frac'],d['train_step_fwd_bwd_ms'])"
  done
done
unset CSPN_AMD_LIB
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/${O}_prof -- python $R/tools/bench_backward.py --batch 64 --steps 20) > ${O}_prof.log 2>&1
python tools/rocpd_summary.py ${O}_prof/*/*.db ${O}_backward_kernel_stats.md | head -8 | cut -c1-200; rm -rf ${O}_prof
tail -2 ${O}_bench.err
