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
OBJS=$(ls build/*.o | grep -v -e cspn2d_tsw4 -e cspn_test_hooks -e "/t4_" -e head_abl)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DTSW4_GEN_INC="\"$inc\"" -x hip -c cspn2d_tsw4.hip -o build/t4_$n.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/libcspn_t4_$n.so $OBJS build/t4_$n.o
echo built cspn_amd/abl/libcspn_t4_$n.so
