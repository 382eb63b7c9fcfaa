#!/bin/bash
for abl in 2 6 14; do echo "== ABL $abl (2: no events; 6: no events, no cook consume; 14: no events, no consume, no loads)"; CSPN_AMD_LIB=$PWD/gpurun_dbg_ABL$abl.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"device_ms_per_launch": [0-9.]*'; done
