#!/bin/bash
# tools/r05/build_ringrec.sh BYTES[w2] (w2: ring stores as ds_write2_b32)  -- timing build of libcspn_amd whose forward loop (single variant: norm 8sum, no mask) keeps the cooked-row ring at a record
# stride of BYTES instead of 168 (more / fewer LDS bank conflicts, profiles/r05_lds_conflicts_and_sq.md) -> cspn_amd/abl/libcspn_ring$BYTES.so.  Results are correct
# for that variant; every other variant of the library runs the same loop (TSW_SINGLE_VARIANT), so use it for bench.py's default workload only.
set -e
cd "$(dirname "$0")/../.."
b=$1
mkdir -p cspn_amd/abl cspn_amd/csrc/build
inc=$PWD/cspn_amd/csrc/build/ring_$b.inc
TSW_RING_REC=${b%w2} TSW_RING_W2=$([[ $b == *w2 ]] && echo 1 || echo 0) python -m tools.tswgen.emit $inc "" 2>/dev/null
cd cspn_amd/csrc
OBJS="build/cspn_abi.cpp.o build/cspn2d_stepwise.hip.o build/cspn3d_stepwise.hip.o build/cspn3d_persistent.hip.o build/cspn3d_backward.hip.o build/cspn2d_fused.hip.o build/cspn2d_backward.hip.o build/cspn_aux.hip.o"
make -s -j8 2>/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DTSW_GEN_INC="\"$inc\"" -x hip -c cspn2d_tsw.hip -o build/ring_$b.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/libcspn_ring$b.so $OBJS build/ring_$b.o
echo built cspn_amd/abl/libcspn_ring$b.so
