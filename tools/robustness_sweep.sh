#!/bin/bash
# The GPU suite under every path-selecting switch: each fallback / alternative path has to give the
# same (reference) results as the default one.   bash tools/robustness_sweep.sh   (on an MI355X)
export TMPDIR=/tmp
for v in "MIFC_FORCE_CELL_KERNEL=1" "MIFC_FUSED2=0 MIFC_SHAPIRO_FUSED=0" "MIFC_HOST_PIPELINE=0" "MIFC_EWISE_MAX_BLOCKS=64" "MIFC_HOST_CHUNK_MIB=4 MIFC_HOST_THREADS=2" "MIFC_VORTDIV_SPLIT=0" "MIFC_VORTDIV_LEVELWALK=0" "MIFC_SHAPIRO_REGS=0" "MIFC_LEVELWALK_MIN_UNITS=1" "MIFC_RAGGED_SPLIT=0 MIFC_LEVELWALK_MIN_UNITS=1" "MIFC_SCALAR_SPLIT_TUNE=TR=8,NL=2,PF=2 MIFC_LEVELWALK_MIN_UNITS=1"; do
  echo "== $v"
  # (the three deselected tests set these switches themselves)
  env $v timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "not one_launch_and_four and not fused_tfp_and_qvector and not band_heights" 2>&1 | tail -n 2 || exit 1
done
