#!/bin/bash
# tools/r06/build_abl_head.sh N ["-Dextra flags" TAG] -- timing build of libcspn_amd whose raw guidance-head kernel is compiled with -DHEAD_ABL=N (cspn_head.hip) -> cspn_amd/abl/libcspn_head_N.so.
# Results are wrong by design: timing only.  Use: CSPN_AMD_LIB=cspn_amd/abl/libcspn_head_N.so python tools/r06/bench_head.py
set -e
cd "$(dirname "$0")/../.."
n=$1; extra=$2; tag=${3:-$1}
mkdir -p cspn_amd/abl
cd cspn_amd/csrc
OBJS=$(ls build/*.o | grep -v -e cspn_head -e cspn_test_hooks -e '/t4_' -e head_abl)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DHEAD_ABL=$n $extra -I. -x hip -c cspn_head.hip -o build/head_abl_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/libcspn_head_$tag.so $OBJS build/head_abl_$tag.o
echo built cspn_amd/abl/libcspn_head_$tag.so
