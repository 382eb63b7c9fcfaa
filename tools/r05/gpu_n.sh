#!/bin/bash
# round 5, run N: the persistent 3D kernel with its tiles started a fixed delay apart along y (P3_STAGGER x 64 cycles per tile row), so that the
# chunk prologues' gate bursts spread over the neighbours' steps.  A/B against the product, separate processes, alternating, same box.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r5n
V3="--workload vol3d --steps 60 --warmup 20 --no-cpu-baseline"
for rnd in 1 2; do
  for v in product ${VARIANTS:-stg16 stg32 stg64 stg127}; do
    if [ $v = product ]; then unset CSPN_AMD_LIB; else export CSPN_AMD_LIB=$R/cspn_amd/abl/libcspn_$v.so; fi
    timeout 300 python bench.py $V3 > ${O}_vol3d_${v}_$rnd.json 2>> ${O}_bench.err
    python -c "import json;d=json.load(open('${O}_vol3d_${v}_$rnd.json'));print('$v $rnd',d['ms_per_step'],d['roofline']['device_ms_per_launch'],d['roofline']['device_ms_min'],d['roofline']['frac'],d['parity_checked']['ok'], d['parity_checked']['oracle_full_volume']['max_rel_err'])"
  done
done
tail -3 ${O}_bench.err
