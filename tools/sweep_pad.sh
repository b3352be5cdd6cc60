#!/bin/bash
# Does the level stride (4,147,200 B = 2^11 * 2025 for 1440x720) matter?  Levels padded by k rows.
export SWEEP_ROUNDS=5
for pad in 0 1 2 3 5 8; do
  echo "== padrows=$pad (level stride $(( (720+pad)*5760 )) B)"
  SWEEP_SHAPE=1440,$((720+pad)),137 python tools/sweep_vortdiv.py "R=8,PADROWS=$pad" "R=6,PADROWS=$pad" 2>&1 | grep -E "^R=|nt-ld\+st, 1 lane"
done
