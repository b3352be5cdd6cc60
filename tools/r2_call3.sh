#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -k "vortdiv or stencil or headline or config4 or config5 or golden or fused" > gpurun_out/r2/fma_test.log 2>&1 || { tail -30 gpurun_out/r2/fma_test.log; exit 1; }
tail -3 gpurun_out/r2/fma_test.log
SWEEP_ROUNDS=7 SWEEP_NO_YARD=1 timeout -k 10 400 python tools/sweep_vortdiv.py "R=8" "R=8,STA=2" "R=8,STA=16" "R=8,STA=17" "R=8,STA=18" "R=8,STA=1" "R=8,NT=0" "R=8,STA=16,XL=1" "R=8,XL=1" "R=8,D=0" "R=8,D=0,STA=16" > gpurun_out/r2/sweep_sta.txt 2>&1 || { tail gpurun_out/r2/sweep_sta.txt; exit 1; }
head -20 gpurun_out/r2/sweep_sta.txt
