#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and, in SEPARATE runs, the HBM-traffic counters of
# the fused derived-variable kernel (BASELINE.json config 2 x 137 levels: ff + RH + theta, and with Td).
#   bash tools/profile_derived.sh [tag]   -> gpurun_out/prof_derived_<tag>/
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_derived_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
export DERIVED_SWEEP=""
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/bench_derived.py 137 > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"; do
  NAME=$(echo "$C" | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$NAME" -- python3 tools/bench_derived.py 137 > "$OUT/pmc_$NAME.log" 2>&1
  echo "pmc $C rc=$?"
done
python3 tools/summarize_prof.py "$OUT" derived_levels_kernel > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
