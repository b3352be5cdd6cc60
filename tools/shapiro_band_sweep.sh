#!/bin/bash
# measurement only: parity of the Shapiro paths, then the register-resident kernel over band heights against the LDS form
#   bash tools/shapiro_band_sweep.sh > gpurun_out/shapiro_band_sweep.txt   (GPU box)
set -u
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "shapiro or stencil_levels_ex or catalogue or golden" > gpurun_out/shapiro_regs_tests.txt 2>&1
tail -5 gpurun_out/shapiro_regs_tests.txt
grep -q "passed" gpurun_out/shapiro_regs_tests.txt || exit 1
grep -q "failed" gpurun_out/shapiro_regs_tests.txt && exit 1
for B in ${BANDS:-0 24 48}; do
  echo "band $B (0 = default)"
  MIFC_FUSED2_BAND=$B BENCH_ONLY=shapiro timeout -k 10 120 python3 tools/tested_variants.py 2>&1 | grep shapiro
done
echo "LDS form"
MIFC_SHAPIRO_REGS=0 BENCH_ONLY=shapiro timeout -k 10 120 python3 tools/tested_variants.py 2>&1 | grep shapiro
