#!/bin/bash
# tools/r06/build_abl4.sh NAME "ablation,flags" ["dict(cfg)"] -- timing build of libcspn_amd whose round-6 loop is ONE generated variant (norm 8sum, no
# mask) with the given ablation flags / generator options -> cspn_amd/abl/libcspn_t4_NAME.so.  Results are wrong with ablation flags: timing only.
set -e
cd "$(dirname "$0")/../.."
n=$1; fl=$2; cfg=${3:-"{}"}
mkdir -p cspn_amd/abl cspn_amd/csrc/build
inc=$PWD/cspn_amd/csrc/build/t4_$n.inc
TSW_CFG="$cfg" python -m tools.tswgen.emit4 $inc "$fl" 2>/dev/null
cd cspn_amd/csrc
OBJS="build/cspn_abi.cpp.o build/cspn2d_stepwise.hip.o build/cspn3d_stepwise.hip.o build/cspn3d_persistent.hip.o build/cspn3d_backward.hip.o build/cspn2d_fused.hip.o build/cspn2d_backward.hip.o build/cspn_aux.hip.o build/cspn2d_tsw.p0.o build/cspn2d_tsw.p3.o build/cspn2d_tsw.p4.o build/cspn2d_tsw.p6.o build/cspn2d_tsw.p1.o build/cspn2d_tsw.p5.o build/cspn2d_tsw.p7.o build/cspn2d_tsw.p8.o"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DTSW4_GEN_INC="\"$inc\"" -x hip -c cspn2d_tsw4.hip -o build/t4_$n.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/libcspn_t4_$n.so $OBJS build/t4_$n.o
echo built cspn_amd/abl/libcspn_t4_$n.so
