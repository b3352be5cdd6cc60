#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/prof_k2
mkdir -p $O
export MIFC_VORTDIV_TUNE="K=2,RB=14,LG=4"
CMD="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-verify --no-check-variant"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  N=$(echo "$C" | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_$N -- $CMD > $O/pmc_$N.log 2>&1
done
python3 tools/summarize_prof.py $O vortdiv_tile_kernel > $O/summary.txt 2>&1
grep -E "vortdiv|FETCH|WRITE|TCC|TCP|HBM|L2 hit|bench.py" $O/summary.txt
