#!/bin/bash
# libcspn_amd with the persistent 3D kernel's phase stamps compiled in (-DP3_TRACE) -> cspn_amd/abl/libcspn_p3trace.so
set -e
cd "$(dirname "$0")/../cspn_amd/csrc"
mkdir -p ../abl build
make -s 2>/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DP3_TRACE -x hip -c cspn3d_persistent.hip -o build/p3trace.o
OBJS="build/cspn_abi.cpp.o build/cspn2d_stepwise.hip.o build/cspn3d_stepwise.hip.o build/cspn3d_backward.hip.o build/cspn2d_fused.hip.o build/cspn2d_tsw.hip.o build/cspn2d_backward.hip.o build/cspn_aux.hip.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/libcspn_p3trace.so $OBJS build/p3trace.o
echo built cspn_amd/abl/libcspn_p3trace.so
