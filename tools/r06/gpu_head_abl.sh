#!/bin/bash
# the raw guidance head: product and the timing builds, alternating, one box
for r in 1 2; do
  python tools/r06/time_head.py 2>/dev/null | tail -1
  for n in 1 2 3 7; do CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_head_$n.so python tools/r06/time_head.py 2>/dev/null | tail -1; done
done
