#!/bin/bash
# round 6, run B: ablations of the 12 x 3 ring (timing-only builds, tools/r06/build_abl4.sh), KITTI x 64, separate processes, two rounds
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
R=$PWD
O=gpurun_out/r6b_abl.txt
: > $O
for rnd in 1 2; do
  unset CSPN_AMD_LIB
  timeout 120 python tools/r06/time_fwd.py 64 304 1216 60 8 >> $O 2>/dev/null
  timeout 120 python tools/r06/time_fwd.py 64 304 1216 60 0 >> $O 2>/dev/null
  for v in ${VARIANTS:-full nocook nodma nodmaload noev core corenobar corenolds nocookmath nocookwrite nocookread nodmawait}; do
    CSPN_AMD_LIB=$R/cspn_amd/abl/libcspn_t4_$v.so timeout 120 python tools/r06/time_fwd.py 64 304 1216 60 0 >> $O 2>/dev/null
  done
done
cat $O
