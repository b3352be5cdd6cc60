#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x -k "vortdiv or relvort or diverg or slab or config2 or wind or levels" > gpurun_out/r2/pytest_k3_default.txt 2>&1 || { tail -30 gpurun_out/r2/pytest_k3_default.txt; exit 1; }
tail -3 gpurun_out/r2/pytest_k3_default.txt
timeout -k 10 500 python tools/levelwalk_threshold.py > gpurun_out/r2/levelwalk_threshold.txt 2>&1 || { tail gpurun_out/r2/levelwalk_threshold.txt; exit 1; }
cat gpurun_out/r2/levelwalk_threshold.txt
