#!/bin/bash
# tools/r04/build_trace.sh -- timing-instrumented single-variant build of the product loop (six s_memtime stamps per wave and step,
# generator option trace) -> cspn_amd/abl/trace/libcspn_amd.so, for tools/tsw_trace.py (CSPN_AMD_LIB=...).  Not the product.
set -e
cd "$(dirname "$0")/../.."
mkdir -p cspn_amd/abl/trace cspn_amd/csrc/build
inc=$PWD/cspn_amd/csrc/build/abl_trace.inc
TSW_CFG="dict(trace=True)" python -m tools.tswgen.emit $inc ""
cd cspn_amd/csrc
OBJS="build/cspn_abi.cpp.o build/cspn2d_stepwise.hip.o build/cspn3d_stepwise.hip.o build/cspn3d_persistent.hip.o build/cspn3d_backward.hip.o build/cspn2d_fused.hip.o build/cspn2d_backward.hip.o build/cspn_aux.hip.o"
make -s $OBJS
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DTSW_GEN_INC="\"$inc\"" -x hip -c cspn2d_tsw.hip -o build/abl_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/trace/libcspn_amd.so $OBJS build/abl_trace.o
echo built cspn_amd/abl/trace/libcspn_amd.so
