#!/bin/bash
# FETCH_SIZE (HBM/fabric read KiB per launch) of the headline kernel for several tunings,
# one rocprofv3 --pmc run each.   bash tools/fetch_sweep.sh "R=8,D=1" "R=32,D=1" ...
set -u
export TMPDIR=/tmp
OUT=gpurun_out/fetch_sweep
mkdir -p "$OUT"
i=0
for T in "$@"; do
  i=$((i+1))
  D="$OUT/run$i"
  rm -rf "$D"
  MIFC_VORTDIV_TUNE="$T" rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D" -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > "$D.log" 2>&1
  F=$(find "$D" -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$T" "$D.log" <<'EOF'
import csv, sys, json
vals = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        if "vortdiv_rows_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            vals.append(float(r["Counter_Value"]))
ms = None
for ln in open(sys.argv[3]):
    if ln.startswith("{"):
        ms = json.loads(ln)["roofline"]["kernel_ms_avg"]
rd = sum(vals) / len(vals) * 1024 * 2
print("%-40s read %.3f GB (x%.3f of 1.136)  kernel %.4f ms (profiled)" % (sys.argv[2], rd / 1e9, rd / 1.136e9, ms or -1))
EOF
done
