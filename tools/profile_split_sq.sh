#!/bin/bash
# SQ-side counters of the split-role one-input stencil kernels, per workgroup shape: where do the cycles of the
# instruction-bound operators go?   bash tools/profile_split_sq.sh "<ops>" "<shapes>" [--tested]  -> gpurun_out/prof_split/
set -u
OPS=${1:-gradient3}
SHAPES=${2:-"TR=12,NL=4,PF=2;TR=14,NL=2,PF=2"}
EXTRA=${3:-}
OUT=gpurun_out/prof_split
mkdir -p "$OUT"
export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"; do
  NAME=$(echo "$C" | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$NAME" -- python3 tools/ab_split_ops.py --ops "$OPS" --shapes "$SHAPES" --rounds 2 --burst 3 $EXTRA > "$OUT/pmc_$NAME.log" 2>&1
  echo "pmc $C rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mifc" in k and ("split" in k or "levelwalk" in k):
            vals[k[:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(vals):
    print(k)
    for c in sorted(vals[k]):
        v = vals[k][c]
        print("   %-28s %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
