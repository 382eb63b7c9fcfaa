#!/bin/bash
# round 3: poll back-off of the persistent 3D kernel: time and memory-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
R=$PWD; out=$R/gpurun_out/r3d_backoff.txt; : > $out
cd /tmp && export TMPDIR=/tmp
for name in "$@"; do
  lib=$R/cspn_amd/abl/libcspn_$name.so
  t=$(CSPN_AMD_LIB=$lib timeout 200 python $R/bench.py --workload vol3d --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print(d['roofline']['device_ms_per_launch'])")
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_$name_$c
    CSPN_AMD_LIB=$lib timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/pm_${name}_$c -o out --output-format csv -- python $R/bench.py --workload vol3d --steps 10 --warmup 3 --prewarm-s 0 --no-cpu-baseline > /dev/null 2>&1
  done
  python - "$name" "$t" <<'PY' | tee -a $out
import csv, glob, sys
name, t = sys.argv[1], sys.argv[2]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob("/tmp/pm_%s_%s/**/*counter_collection.csv" % (name, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if "persistent" in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"]))
    res[c] = sum(vals) / max(1, len(vals))
print("%s: %s ms/forward, FETCH_SIZE %.0f KiB (x2 = %.3f GB), WRITE_SIZE %.0f KiB (%.3f GB), memory-side total %.3f GB" % (
    name, t, res["FETCH_SIZE"], 2 * res["FETCH_SIZE"] * 1024 / 1e9, res["WRITE_SIZE"], res["WRITE_SIZE"] * 1024 / 1e9,
    (2 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024 / 1e9))
PY
done
