#!/bin/bash
for abl in 1 2 3; do echo "== events off; ABL $abl (1=no global loads, 2=no cook math/LDS writes, 3=both)"; CSPN_AMD_LIB=$PWD/gpurun_dbg_ABL$abl.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"device_ms_per_launch": [0-9.]*'; done
