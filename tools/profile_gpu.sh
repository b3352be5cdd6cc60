#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and, in SEPARATE runs,
# the PMC counters for HBM traffic / L2 behaviour of the headline kernel.
#   bash tools/profile_gpu.sh [tag]
# Summaries land under gpurun_out/prof_<tag>/; copy what is to be judged into profiles/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
# The default command with MANY timed steps: rocprof averages over ALL dispatches of the kernel, i.e. also over
# the ~300 probe launches of the placement step (other arrays, 0-4 % slower); with 1000 timed launches on the
# chosen arrays the average is theirs to 1 %.
CMD="python3 bench.py --steps 1000 --warmup 5 --no-cpu-baseline --no-verify --no-check-variant"
PMC_CMD="python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-verify --no-check-variant"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES"; do
  NAME=$(echo "$C" | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$NAME" -- $PMC_CMD > "$OUT/pmc_$NAME.log" 2>&1
  echo "pmc $C rc=$?"
done
# compact summaries
python3 tools/summarize_prof.py "$OUT" vortdiv_levelwalk_kernel > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
