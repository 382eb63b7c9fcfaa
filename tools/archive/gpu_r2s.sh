#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sited8" 2>&1 | tail -4
for l in planar sited8 planar sited8; do
timeout 200 python bench.py --no-cpu-baseline --layout $l 2>/dev/null | python -c "import sys,json; d=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][0]); print('$l', d['roofline']['device_ms_per_launch'], d['roofline']['device_ms_min'], d['parity_checked']['ok'])" | tee -a gpurun_out/r2s_sited8_ab.txt
done
timeout 200 python bench.py --no-cpu-baseline --layout sited8 > gpurun_out/r2s_bench_sited8.json 2>/dev/null
python - <<'PY'
import torch, time, sys, os
sys.path.insert(0, os.getcwd())
import cspn_amd
g = torch.randn(64, 8, 304, 1216, device="cuda")
for _ in range(3): cspn_amd.guidance_to_sited8(g)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): cspn_amd.guidance_to_sited8(g)
e1.record(); torch.cuda.synchronize()
print("relayout kernel ms (64 x 8 x 304 x 1216):", e0.elapsed_time(e1) / 10)
PY
