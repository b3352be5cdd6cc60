#!/bin/bash
# SQ-side counters of the two-stage stencil kernels (TFP, Q-vector, Shapiro) over a 137-level batch: where do the cycles go?
#   bash tools/profile_f1_sq.sh   -> gpurun_out/prof_f1/
set -u
OUT=gpurun_out/prof_f1
mkdir -p "$OUT"
export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  NAME=$(echo "$C" | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$NAME" -- python3 tools/bench_f1_levels.py 137 > "$OUT/pmc_$NAME.log" 2>&1
  echo "pmc $C rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mifc" in k:
            vals[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(vals):
    print(k)
    for c in sorted(vals[k]):
        v = vals[k][c]
        print("   %-28s %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
