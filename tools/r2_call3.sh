#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "vortdiv or relvort or diverg" > gpurun_out/r2/pytest_k3_twopass.txt 2>&1 || { tail -30 gpurun_out/r2/pytest_k3_twopass.txt; exit 1; }
tail -2 gpurun_out/r2/pytest_k3_twopass.txt
B="K=3,ZZ=1"
SWEEP_ROUNDS=7 SWEEP_NO_YARD=1 timeout -k 10 400 python tools/sweep_vortdiv.py "R=8" "$B,RB=12,LG=6,D=0" "$B,RB=12,LG=6,D=1" "$B,RB=16,LG=6,D=0" "$B,RB=16,LG=5,D=0" "$B,RB=16,LG=8,D=0" "$B,RB=8,LG=6,D=0" "$B,RB=8,LG=4,D=0" "$B,RB=16,LG=6,D=1" > gpurun_out/r2/sweep_k3_$1.txt 2>&1 || { tail gpurun_out/r2/sweep_k3_$1.txt; exit 1; }
head -16 gpurun_out/r2/sweep_k3_$1.txt
timeout -k 10 200 python bench.py > gpurun_out/r2/bench_$1.json && python -c "
import json;d=json.load(open('gpurun_out/r2/bench_$1.json'));print(d['roofline']);print(d['check_variant'])"
