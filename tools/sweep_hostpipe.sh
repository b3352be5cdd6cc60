#!/bin/bash
# Sweep of the host-pipeline knobs (copy threads x chunk size) on the GPU box:
#   tools/sweep_hostpipe.sh > gpurun_out/hostpipe_sweep.txt
for th in 4 8 16 32; do
  for mib in 8 16 32 64; do
    echo "threads=$th chunk_mib=$mib"
    MIFC_HOST_THREADS=$th MIFC_HOST_CHUNK_MIB=$mib python tools/bench_hostpath.py --only-batched --reps 3 || exit 1
  done
done
