#!/bin/bash
# time variants of the backward (tools/build_bwdvar.sh): usage gpu_bwdvar.sh NAME...   (product = the shipped library)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for name in "$@"; do
  lib=$PWD/cspn_amd/abl/libcspn_$name.so
  [ "$name" = product ] && lib=$PWD/cspn_amd/libcspn_amd.so
  (cd /tmp && CSPN_AMD_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bv -- python $GRAFT_REPO_ROOT/tools/bench_backward.py --batch 64) > gpurun_out/prof_bv.log 2>&1
  python tools/rocpd_summary.py gpurun_out/prof_bv/*/*.db gpurun_out/bv_$name.md | grep -E "final_ck" | cut -c1-60,95-170 | sed "s/^/$name /"
  rm -rf gpurun_out/prof_bv
done
