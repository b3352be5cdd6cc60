#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "stencil_levels or band_heights or stencils_device" > gpurun_out/r2/pytest_scalar_walk.txt 2>&1 || { tail -30 gpurun_out/r2/pytest_scalar_walk.txt; exit 1; }
tail -2 gpurun_out/r2/pytest_scalar_walk.txt
timeout -k 10 300 python tools/ab_levelwalk_ops.py > gpurun_out/r2/ab_levelwalk_ops.txt 2>&1 || { tail gpurun_out/r2/ab_levelwalk_ops.txt; exit 1; }
cat gpurun_out/r2/ab_levelwalk_ops.txt
timeout -k 10 300 python tools/ab_levelwalk_ops.py 720,360,40 > gpurun_out/r2/ab_levelwalk_ops_small.txt 2>&1 || { tail gpurun_out/r2/ab_levelwalk_ops_small.txt; exit 1; }
cat gpurun_out/r2/ab_levelwalk_ops_small.txt
