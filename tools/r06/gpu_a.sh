#!/bin/bash
# round 6, run A: first contact of the 12 x 3 ring with the GPU: parity + A/B timing against the 8 x 4 ring
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
timeout 600 python tools/r06/ab_ring.py 40 > gpurun_out/r6a_ab.jsonl 2> gpurun_out/r6a_ab.err
cat gpurun_out/r6a_ab.jsonl; tail -5 gpurun_out/r6a_ab.err
