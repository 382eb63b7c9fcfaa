#!/bin/bash
# round 6, run D: new ring vs old ring over batch sizes / shapes (dispatch rule)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
timeout 900 python tools/r06/sweep_ab.py > gpurun_out/r6d_sweep.jsonl 2> gpurun_out/r6d_sweep.err
cat gpurun_out/r6d_sweep.jsonl; tail -3 gpurun_out/r6d_sweep.err
