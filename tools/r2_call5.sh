#!/bin/bash
mkdir -p gpurun_out/r2
B="K=3,RB=12,ZZ=1,D=0"
for shp in 1440,720,16 1440,720,32 1440,720,64; do
SWEEP_SHAPE=$shp SWEEP_ROUNDS=7 SWEEP_NO_YARD=1 timeout -k 10 300 python tools/sweep_vortdiv.py "R=8" "$B,LG=2" "$B,LG=3" "$B,LG=4" "$B,LG=6" "$B,LG=8" "K=3,RB=12,ZZ=1,D=1,LG=4" "K=3,RB=8,ZZ=1,D=0,LG=4" > gpurun_out/r2/sweep_k3_lg_$shp.txt 2>&1 || { tail gpurun_out/r2/sweep_k3_lg_$shp.txt; exit 1; }
head -10 gpurun_out/r2/sweep_k3_lg_$shp.txt
done
