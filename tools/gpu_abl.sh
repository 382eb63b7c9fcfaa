#!/bin/bash
for abl in 1 2 3 6; do echo "== ABL $abl (bit0: no global loads, bit1: no events, bit2: no cook consume)"; CSPN_AMD_LIB=$PWD/gpurun_dbg_ABL$abl.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"device_ms_per_launch": [0-9.]*'; done
