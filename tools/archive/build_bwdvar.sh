#!/bin/bash
# tools/build_bwdvar.sh NAME "-DFLAG ..."  -- libcspn_amd with a timing variant of cspn2d_backward.hip (BWD_EXP_* switches give
# WRONG RESULTS) -> cspn_amd/abl/libcspn_NAME.so, selected at run time with CSPN_AMD_LIB
set -e
cd "$(dirname "$0")/../cspn_amd/csrc"
mkdir -p ../abl build
make -s 2>/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DBWD_EXPERIMENT_BUILD $2 -x hip -c cspn2d_backward.hip -o build/bwdvar_$1.o
OBJS="build/cspn_abi.cpp.o build/cspn2d_stepwise.hip.o build/cspn3d_stepwise.hip.o build/cspn3d_persistent.hip.o build/cspn3d_backward.hip.o build/cspn2d_fused.hip.o build/cspn2d_tsw.p0.o build/cspn2d_tsw.p1.o build/cspn2d_tsw.p2.o build/cspn2d_tsw3.hip.o build/cspn_aux.hip.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/libcspn_$1.so $OBJS build/bwdvar_$1.o
rm -f build/bwdvar_$1.o
echo built cspn_amd/abl/libcspn_$1.so
