#!/bin/bash
mkdir -p gpurun_out/r2
B="K=3,ZZ=1,RB=12,LG=6,D=0"
SWEEP_ROUNDS=9 SWEEP_NO_YARD=1 timeout -k 10 400 python tools/sweep_vortdiv.py "$B" "$B,NTI=1" "R=8" "$B,NTI=1,LG=8" "$B,LG=8" "K=3,ZZ=1,RB=16,LG=6,D=0,NTI=1" "K=3,ZZ=1,RB=16,LG=6,D=0" > gpurun_out/r2/sweep_k3_$1.txt 2>&1 || { tail gpurun_out/r2/sweep_k3_$1.txt; exit 1; }
head -14 gpurun_out/r2/sweep_k3_$1.txt
