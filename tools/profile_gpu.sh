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
# The default command with MANY timed steps: --stats averages over ALL dispatches of the kernel, i.e. also over
# the probe launches of the placement search (other arrays, 0-12 % slower): the search is cut to 48 probes here
# (~830 dispatches next to the 6 000 of the timed region and the per-launch pairs), and tools/summarize_prof.py
# also averages the LAST dispatches of the kernel trace alone (all on the chosen batch).  The counter passes skip the search: the bytes a launch moves do not depend on where the arrays lie.
CMD="python3 bench.py --steps 3000 --warmup 5 --no-cpu-baseline --no-verify --no-check-variant --no-as-allocated --placement-tries 48"
PMC_CMD="python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-verify --no-check-variant --no-as-allocated --placement-pool 0"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES"; do
  NAME=$(echo "$C" | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$NAME" -- $PMC_CMD > "$OUT/pmc_$NAME.log" 2>&1
  echo "pmc $C rc=$?"
done
# compact summaries
python3 tools/summarize_prof.py "$OUT" vortdiv_split_kernel > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
