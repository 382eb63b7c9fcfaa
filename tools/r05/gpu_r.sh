#!/bin/bash
# round 5, run R: the 3D kernel's poll loop with a pause between two polls of quads that were not there yet (P3_BACKOFF x 64 cycles): time (alternating
# rounds) and memory-side read traffic (FETCH_SIZE) against the product (no pause)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r5r
V3="--workload vol3d --steps 60 --warmup 20 --no-cpu-baseline"
for rnd in 1 2; do
  for v in product bo1 bo2 bo4; do
    if [ $v = product ]; then unset CSPN_AMD_LIB; else export CSPN_AMD_LIB=$R/cspn_amd/abl/libcspn_$v.so; fi
    timeout 300 python bench.py $V3 > ${O}_vol3d_${v}_$rnd.json 2>> ${O}_bench.err
    python -c "import json;d=json.load(open('${O}_vol3d_${v}_$rnd.json'));print('$v $rnd',d['ms_per_step'],d['roofline']['device_ms_per_launch'],d['roofline']['device_ms_min'],d['roofline']['frac'],d['parity_checked']['ok'])"
  done
done
for v in product bo1 bo2 bo4; do
  if [ $v = product ]; then unset CSPN_AMD_LIB; else export CSPN_AMD_LIB=$R/cspn_amd/abl/libcspn_$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/${O}_pmc_${v}_$c -- python $R/bench.py --workload vol3d --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check) > ${O}_pmc_${v}_$c.log 2>&1
    echo "$v $c: $(python tools/rocpd_summary.py ${O}_pmc_${v}_$c/*/*.db ${O}_pmc_${v}_$c.md | grep persistent | grep $c | cut -c1-200)"; rm -rf ${O}_pmc_${v}_$c
  done
done
