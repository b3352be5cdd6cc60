#!/bin/bash
# SQ-side counters of the headline kernel (issue / wait breakdown), own rocprofv3 run.
#   bash tools/profile_sq.sh [tag]
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_sq" -- $CMD > "$OUT/pmc_sq.log" 2>&1
echo "pmc sq rc=$?"
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MUL_F64 --output-format csv -d "$OUT/pmc_sq2" -- $CMD > "$OUT/pmc_sq2.log" 2>&1
echo "pmc sq2 rc=$?"
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary_sq.txt" 2>&1
cat "$OUT/summary_sq.txt"
