#!/bin/bash
# tools/build_abl.sh NAME FLAGS  -- timing-experiment build of libcspn_amd with a single generated asm loop variant
# (FLAGS: comma list of nocook,noevents,noact,nobar,nolds, or "" for the full loop) -> cspn_amd/abl/libcspn_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
mkdir -p cspn_amd/abl cspn_amd/csrc/build
inc=$PWD/cspn_amd/csrc/build/abl_$name.inc
python -m tools.tswgen.emit $inc "$flags"
cd cspn_amd/csrc
OBJS="build/cspn_abi.cpp.o build/cspn2d_stepwise.hip.o build/cspn3d_stepwise.hip.o build/cspn3d_persistent.hip.o build/cspn3d_backward.hip.o build/cspn2d_fused.hip.o build/cspn2d_tsw3.hip.o build/cspn2d_backward.hip.o build/cspn_aux.hip.o"
make -s $OBJS 2>/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DTSW_GEN_INC="\"$inc\"" -x hip -c cspn2d_tsw.hip -o build/abl_$name.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/libcspn_$name.so $OBJS build/abl_$name.o 2>/dev/null
echo built cspn_amd/abl/libcspn_$name.so
