#!/bin/bash
# Which schedule is robust against the level stride?
export SWEEP_ROUNDS=5
for pad in 0 5 8; do
  echo "== padrows=$pad (level stride $(( (720+pad)*5760 )) B)"
  P="PADROWS=$pad"
  SWEEP_SHAPE=1440,$((720+pad)),137 python tools/sweep_vortdiv.py "R=8,$P" "R=8,WPB=1,$P" "R=8,WPB=2,$P" "R=8,ORDER=0,$P" "R=8,ORDER=2,$P" "R=8,ZZ=0,$P" "R=8,XCD=0,$P" "R=8,V=1,$P" "R=4,$P" "R=8,XS=1,$P" 2>&1 | grep -E "^R=|nt-ld\+st, 1 lane"
done
