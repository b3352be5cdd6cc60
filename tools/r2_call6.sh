#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -q -m gpu --maxfail=10 > gpurun_out/r2/all_tests_c.log 2>&1
rc=$?
tail -15 gpurun_out/r2/all_tests_c.log
exit $rc
