#!/bin/bash
# round-2 GPU call 1: new at-scale tests, self-checking bench, placement / pitch sweeps
mkdir -p gpurun_out/r2
timeout -k 10 780 python -m pytest tests/test_gpu_scale.py -x -q -m gpu > gpurun_out/r2/scale.log 2>&1
rc=$?
tail -n 30 gpurun_out/r2/scale.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r2/bench_a.json 2> gpurun_out/r2/bench_a.err || { tail -n 20 gpurun_out/r2/bench_a.err; exit 1; }
cat gpurun_out/r2/bench_a.json
timeout -k 10 200 python tools/sweep_placement.py A B > gpurun_out/r2/placement.txt 2>&1 || { tail -n 20 gpurun_out/r2/placement.txt; exit 1; }
tail -n 60 gpurun_out/r2/placement.txt
for shape in 1440,720,137 1536,675,137 1280,810,137 1408,736,137 2048,506,137; do
  SWEEP_SHAPE=$shape SWEEP_ROUNDS=4 timeout -k 10 120 python tools/sweep_vortdiv.py "R=8" "R=8,XL=1" "R=8,XS=1" > gpurun_out/r2/pitch_$shape.txt 2>&1 || { tail gpurun_out/r2/pitch_$shape.txt; exit 1; }
  grep -E "^shape|^R=8|nt-ld\+st, 1 lane|fill both outputs, (linear|4x256)" gpurun_out/r2/pitch_$shape.txt
done
