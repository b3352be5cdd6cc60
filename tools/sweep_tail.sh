#!/bin/bash
# How does the fused kernel's time depend on the batch size (tail / quantisation of workgroup rounds)
# and on the unit size (R, WPB)?   tools/sweep_tail.sh > gpurun_out/sweep_tail.txt
export SWEEP_ROUNDS=5
for nlev in 96 104 112 120 128 137 144 152 160; do
  echo "== nlev=$nlev"
  SWEEP_SHAPE=1440,720,$nlev python tools/sweep_vortdiv.py "R=8" "R=6" 2>&1 | grep -E "^R=|nt-ld\+st, 1 lane"
done
echo "== unit size, nlev=137"
python tools/sweep_vortdiv.py "R=3" "R=4" "R=5" "R=6" "R=7" "R=8" "R=4,WPB=2" "R=6,WPB=2" "R=8,WPB=2" "R=4,WPB=8" "R=6,WPB=8" 2>&1 | grep -E "^R=|nt-ld\+st, 1 lane"
