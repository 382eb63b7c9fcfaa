#!/bin/bash
# effective shader clock of single-variant builds: GRBM_GUI_ACTIVE / kernel duration (rocprofv3, PMC pass on its own)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
for name in "$@"; do
  rm -rf /tmp/clk_$name
  CSPN_AMD_LIB=$R/cspn_amd/abl/libcspn_$name.so timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/clk_$name -o out --output-format csv -- python $R/bench.py --no-cpu-baseline --no-parity-check --algo fused --prewarm-s 0.5 --steps 60 --warmup 10 > /tmp/clk_$name.log 2>&1
  python - "$name" <<'PY'
import csv, glob, sys
name = sys.argv[1]
rows = []
for f in glob.glob("/tmp/clk_%s/**/*counter_collection.csv" % name, recursive=True):
    rows += list(csv.DictReader(open(f)))
kt = {}
for f in glob.glob("/tmp/clk_%s/**/*kernel_trace.csv" % name, recursive=True):
    for r in csv.DictReader(open(f)):
        kt[r.get("Dispatch_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = {}
for r in rows:
    if "tsw3" not in r["Kernel_Name"]:
        continue
    d = kt.get(r["Dispatch_Id"])
    acc.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), d[0] if d else 0))
for c, v in acc.items():
    v = v[len(v) // 2:]
    val = sum(x[0] for x in v) / len(v); ns = sum(x[1] for x in v) / len(v)
    print("%s %s: %.0f per dispatch, %.1f us -> %.3f per ns" % (name, c, val, ns / 1e3, val / ns if ns else 0))
PY
done
