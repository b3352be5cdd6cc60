#!/bin/bash
# Everything profiles/README.md quotes for a round, on ONE box in one gpurun call:
#   bash tools/round_end_measure.sh r01   (outputs under gpurun_out/)
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 3 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
echo "bench rc=$?"; cat gpurun_out/bench_n1.json
python3 bench.py --steps 20 --warmup 3 --check --no-cpu-baseline > gpurun_out/bench_n1_check.json 2>/dev/null
echo "bench --check rc=$?"
python3 tools/sweep_vortdiv.py "R=6" "R=8" "R=6,XS=1" > gpurun_out/sweep_same_device_as_bench.txt 2>&1
echo "sweep rc=$?"
bash tools/profile_gpu.sh "$TAG" > gpurun_out/profile_gpu.log 2>&1
echo "profile rc=$?"; tail -25 gpurun_out/profile_gpu.log
python3 tools/bench_ops.py 137 > gpurun_out/per_operator_table.txt 2>&1
echo "ops rc=$?"
# the two stencil-of-a-stencil operators: one fused launch vs the multi-pass path, same box
{ echo "# fused (default)"; BENCH_OPS_ONLY="thermalFront|qvector" python3 tools/bench_ops.py 137 | tail -n 3;
  echo "# MIFC_FUSED2=0 (multi-pass)"; MIFC_FUSED2=0 BENCH_OPS_ONLY="thermalFront|qvector" python3 tools/bench_ops.py 137 | tail -n 3; } > gpurun_out/fused2_ops.txt 2>&1
echo "fused2 rc=$?"
python3 tools/bench_hostpath.py > gpurun_out/hostpath_after.jsonl 2>&1
echo "hostpath rc=$?"
python3 tools/bench_configs.py > gpurun_out/other_configs.jsonl 2>&1
echo "configs rc=$?"
