#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu --maxfail=10 > gpurun_out/r2/all_tests_b.log 2>&1
rc=$?
tail -25 gpurun_out/r2/all_tests_b.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/bench_f1_levels.py > gpurun_out/r2/bench_f1_levels.txt 2>&1 || { tail gpurun_out/r2/bench_f1_levels.txt; exit 1; }
cat gpurun_out/r2/bench_f1_levels.txt
