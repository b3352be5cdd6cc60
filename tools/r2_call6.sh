#!/bin/bash
mkdir -p gpurun_out/r2
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[1], 'median %.4f min %.4f frac %.4f check %.4f' % (r['kernel_ms_median'], r['kernel_ms_min'], r['frac'], d['check_variant']['kernel_ms_median']), d['config']['placement'])" $1; }
for i in 1 2 3 4; do
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r2/bench_placed_$i.json && show gpurun_out/r2/bench_placed_$i.json || exit 1
done
