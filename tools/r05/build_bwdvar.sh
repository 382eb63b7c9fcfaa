#!/bin/bash
# tools/r05/build_bwdvar.sh NAME "-DFLAG ..."  -- libcspn_amd with a variant of cspn2d_backward.hip -> cspn_amd/abl/libcspn_NAME.so (A/B timing through
# CSPN_AMD_LIB=...).  -DBWD_FINAL_CK: the round-3 final pass (image-order pixel pairs) instead of round 5's mixed pairs.
set -e
cd "$(dirname "$0")/../../cspn_amd/csrc"
mkdir -p ../abl build
make -s -j8 2>/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize $2 -x hip -c cspn2d_backward.hip -o build/bwdvar_$1.o
OBJS="build/cspn_abi.cpp.o build/cspn2d_stepwise.hip.o build/cspn3d_stepwise.hip.o build/cspn3d_persistent.hip.o build/cspn3d_backward.hip.o build/cspn2d_fused.hip.o build/cspn_aux.hip.o build/cspn2d_tsw.p0.o build/cspn2d_tsw.p3.o build/cspn2d_tsw.p4.o build/cspn2d_tsw.p6.o build/cspn2d_tsw.p1.o build/cspn2d_tsw.p5.o build/cspn2d_tsw.p7.o build/cspn2d_tsw.p8.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/libcspn_$1.so $OBJS build/bwdvar_$1.o
echo built cspn_amd/abl/libcspn_$1.so
