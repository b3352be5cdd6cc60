#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu --maxfail=10 -k "winddir or catalogue or golden or families or pointwise or known or elementwise or python_surface" > gpurun_out/r2/tests_c.log 2>&1
rc=$?
tail -12 gpurun_out/r2/tests_c.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python tools/bench_ops.py 137 > gpurun_out/r2/bench_ops.txt 2>&1 || { tail -20 gpurun_out/r2/bench_ops.txt; exit 1; }
grep -v "^{" gpurun_out/r2/bench_ops.txt | tail -70
