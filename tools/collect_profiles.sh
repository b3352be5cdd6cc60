#!/bin/bash
# After `gpurun -- bash tools/round_end_measure.sh <tag>`: copies what profiles/README.md quotes from the scratch
# directory gpurun_out/ into profiles/<tag>/ (tracked).   bash tools/collect_profiles.sh r02
set -eu
TAG=${1:-r02}
E=gpurun_out/${TAG}_end
P=gpurun_out/prof_${TAG}
D=gpurun_out/prof_derived_${TAG}
O=profiles/${TAG}
mkdir -p "$O"
cp "$E"/bench_n1.json "$E"/bench_repeat.txt "$E"/bench_derived.txt "$E"/bench_f1_levels.txt "$E"/per_operator_table.txt "$E"/hostpath.jsonl \
   "$E"/other_configs.jsonl "$E"/multigpu_gloo_rehearsal.jsonl "$E"/sweep_same_device_as_bench.txt "$E"/box.txt "$O"/
newest() { ls -t "$@" 2>/dev/null | head -1; }
cp "$(newest "$P"/trace/*/*kernel_stats.csv)" "$O"/kernel_stats.csv
cp "$P"/summary.txt "$O"/rocprofv3_summary.txt
cp "$P"/pmc_traffic.json "$O"/pmc_traffic.json
cp "$P"/pmc_traffic.json profiles/pmc_traffic.json
for d in "$P"/pmc_*/; do
  n=$(basename "$d")
  # per-dispatch counter rows of the headline kernel only (the whole file also lists torch's kernels)
  f=$(newest "$d"*/*counter_collection.csv)
  [ -n "$f" ] && { head -1 "$f"; grep vortdiv "$f" | head -400; } > "$O/$n.csv"
done
cp "$D"/summary.txt "$O"/derived_kernel_pmc.txt
[ -f "$E"/tested_variants.txt ] && cp "$E"/tested_variants.txt "$O"/tested_variants.txt
[ -f gpurun_out/strict_tolerance_report.txt ] && cp gpurun_out/strict_tolerance_report.txt "$O"/strict_tolerance_report.txt
ls -la "$O" | head -40
