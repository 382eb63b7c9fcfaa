#!/bin/bash
# shader clock during a run = GRBM_GUI_ACTIVE cycles / kernel duration, per ablation build
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for name in "$@"; do
 (cd /tmp && CSPN_AMD_LIB=$GRAFT_REPO_ROOT/cspn_amd/abl/libcspn_$name.so timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/clk2_$name -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 100 --no-cpu-baseline --algo fused) > gpurun_out/clk2_$name.log 2>&1
 python tools/rocpd_summary.py gpurun_out/clk2_$name/*/*.db gpurun_out/clk2_$name.md | grep -E "tsw" | cut -c1-200 | sed "s/^/$name /"
done
