#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 120 ./tools/h2_probe > gpurun_out/r2/h2_probe.txt 2>&1 || { tail gpurun_out/r2/h2_probe.txt; exit 1; }
cat gpurun_out/r2/h2_probe.txt
timeout -k 10 600 python -m pytest tests/test_gpu_scale.py -q -m gpu --maxfail=12 > gpurun_out/r2/scale2.log 2>&1
rc=$?
tail -n 40 gpurun_out/r2/scale2.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_capi_and_host.py -q -m gpu --maxfail=12 > gpurun_out/r2/parity2.log 2>&1
rc=$?
tail -n 25 gpurun_out/r2/parity2.log
exit $rc
