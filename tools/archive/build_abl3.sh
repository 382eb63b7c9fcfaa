#!/bin/bash
# tools/build_abl3.sh NAME FLAGS  -- timing-experiment build of libcspn_amd with a single variant of the round-3 loop
# (FLAGS: comma list of kernel3.py ablations, "" for the full loop; generator options via TSW_CFG="dict(...)")
# -> cspn_amd/abl/libcspn_NAME.so, selected at run time with CSPN_AMD_LIB.  Ablations give wrong results: timings only.
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
mkdir -p cspn_amd/abl cspn_amd/csrc/build
inc=$PWD/cspn_amd/csrc/build/abl3_$name.inc
python -m tools.tswgen.emit3 $inc "$flags"
cd cspn_amd/csrc
OBJS="build/cspn_abi.cpp.o build/cspn2d_stepwise.hip.o build/cspn3d_stepwise.hip.o build/cspn3d_persistent.hip.o build/cspn3d_backward.hip.o build/cspn2d_fused.hip.o build/cspn2d_tsw.p0.o build/cspn2d_tsw.p1.o build/cspn2d_tsw.p2.o build/cspn2d_backward.hip.o build/cspn_aux.hip.o"
make -s $OBJS 2>/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DTSW3_GEN_INC="\"$inc\"" -x hip -c cspn2d_tsw3.hip -o build/abl3_$name.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/libcspn_$name.so $OBJS build/abl3_$name.o 2>/dev/null
rm -f build/abl3_$name.o $inc
echo built cspn_amd/abl/libcspn_$name.so
