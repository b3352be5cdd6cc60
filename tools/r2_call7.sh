#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -q -m gpu --maxfail=10 > gpurun_out/r2/all_tests_d.log 2>&1
rc=$?
tail -8 gpurun_out/r2/all_tests_d.log
if [ $rc -ne 0 ]; then exit $rc; fi
DERIVED_SWEEP=",8192,32768" python tools/bench_derived.py 137 > gpurun_out/r2/bench_derived_kappa.txt 2>&1; cat gpurun_out/r2/bench_derived_kappa.txt
BENCH_OPS_ONLY="hleveltemp|aleveltemp|alevelhum|hlevelhum|hlevelthe|fused" python tools/bench_ops.py 137 2>&1 | grep -v "^{" | tail -12
./tools/call_latency_probe | head -6
