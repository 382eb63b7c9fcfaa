#!/bin/bash
# the raw guidance head: product and the timing builds (tools/r06/build_abl_head.sh), alternating, one box
for r in 1 2; do
  python tools/r06/time_head.py 2>/dev/null | tail -1
  for f in cspn_amd/abl/libcspn_head_*.so; do CSPN_AMD_LIB=$PWD/$f python tools/r06/time_head.py 2>/dev/null | tail -1; done
done
