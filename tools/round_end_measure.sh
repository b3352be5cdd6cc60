#!/bin/bash
# Everything profiles/README.md quotes for a round, on ONE box in one gpurun call:
#   bash tools/round_end_measure.sh r03 [a|b|all]   (outputs under gpurun_out/<tag>_end/)
set -u
TAG=${1:-r03}
PART=${2:-all}   # "a": bench + repeats + rocprofv3 passes; "b": bench once more + every table (each part fits one gpurun call); "all"
O=gpurun_out/${TAG}_end
mkdir -p "$O"
export TMPDIR=/tmp
if [ "$PART" = "b" ]; then
  # the tables of part b are quoted next to a headline number measured in the SAME call, on the same box
  { rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id"; } > "$O/box_tables.txt" 2>&1
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench_n1_with_tables.json" 2> "$O/bench_n1_with_tables.err"
  echo "bench (tables' box) rc=$?"; cat "$O/bench_n1_with_tables.json"
fi
if [ "$PART" != "b" ]; then
# which box: the same binary runs the headline kernel at 66 % on some boxes of the pool and at 73.5 % on others
{ rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id"; echo "HBM vendor: $(cat /sys/class/drm/card*/device/mem_info_vram_vendor 2>/dev/null | head -1)"; echo "vbios: $(cat /sys/class/drm/card*/device/vbios_version 2>/dev/null | head -1)"; } > "$O/box.txt" 2>&1
cat "$O/box.txt"
python3 bench.py --steps 20 --warmup 5 > "$O/bench_n1.json" 2> "$O/bench_n1.err"
echo "bench rc=$?"; cat "$O/bench_n1.json"
echo "# five consecutive processes of 'python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline' on ONE box, same binary" > "$O/bench_repeat.txt"
echo "# Mcells/s  ms_per_step(wall)  kernel_ms_avg(HIP events)  roofline.frac  tested-variant frac (median of 5 bursts)  as-allocated frac  verified  chosen ms per placement pool  tested-variant bursts (ms)" >> "$O/bench_repeat.txt"
for i in 1 2 3 4 5; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d.get('check_variant',{}).get('roofline_frac'), (d.get('roofline_as_allocated') or {}).get('frac'), d['verified'], d['config']['placement'].get('rounds_chosen_ms'), d.get('check_variant',{}).get('bursts_ms'))" >> "$O/bench_repeat.txt"
done
cat "$O/bench_repeat.txt"
W="K=3,RB=12,LG=6,ZZ=1,D=0"   # the level-walking kernel whose waves load and store (MIFC_VORTDIV_SPLIT=0); the measurement knobs exist for it
W4="K=4,RB=12,LG=6,D=1,WPB=2"  # what the library picks for this batch: the split-role kernel
SWEEP_NO_YARD=1 python3 tools/sweep_vortdiv.py "$W4" "$W" "$W,XS=1" "$W,XL=1" "$W,XS=1,XL=1" "R=8" "R=8,XS=1" "R=8,XL=1" > "$O/sweep_same_device_as_bench.txt" 2>&1
echo "sweep rc=$?"
bash tools/profile_gpu.sh "$TAG" > "$O/profile_gpu.log" 2>&1
echo "profile rc=$?"; tail -25 "$O/profile_gpu.log"
bash tools/profile_derived.sh "$TAG" > "$O/profile_derived.log" 2>&1
echo "profile derived rc=$?"; tail -25 "$O/profile_derived.log"
fi
if [ "$PART" = "a" ]; then exit 0; fi
python3 tools/bench_derived.py 137 > "$O/bench_derived.txt" 2>&1
echo "derived rc=$?"
python3 tools/bench_f1_levels.py 137 > "$O/bench_f1_levels.txt" 2>&1
echo "f1 rc=$?"
python3 tools/bench_ops.py 137 > "$O/per_operator_table.txt" 2>&1
echo "ops rc=$?"
python3 tools/tested_variants.py 137 > "$O/tested_variants.txt" 2>&1
echo "tested variants rc=$?"
# round 3: split-role forms against the forms whose waves load and store, per-level calls as one graph, the config-4 step
{ python3 tools/ab_split_ops.py; python3 tools/ab_split_ops.py --tested; } 2>&1 | grep -v amdgpu.ids > "$O/split_role_ops.txt"
echo "split-role A/B rc=$?"
python3 tools/bench_graph_levels.py 2>&1 | grep -v amdgpu.ids > "$O/graph_levels.txt"
echo "graph rc=$?"
for A in "--legacy-step" "" "--levels 8" "--levels 32"; do for T in "" "--all-defined"; do
  python3 tools/bench_multigpu.py --config 4 $A $T --steps 200 2>/dev/null | grep "^{" >> "$O/config4_one_rank_rccl.jsonl"
done; done
echo "config 4 (one rank, RCCL) rc=$?"
for A in "--unfused-ff" ""; do python3 tools/bench_multigpu.py --config 5 --members 6 --steps 3 --check $A 2>/dev/null | grep "^{" >> "$O/config5_one_rank.jsonl"; done
echo "config 5 (one rank) rc=$?"
python3 tools/ragged_width.py > "$O/ragged_width.txt" 2>&1
echo "ragged rc=$?"
python3 tools/bench_hostpath.py > "$O/hostpath.jsonl" 2>&1
echo "hostpath rc=$?"
python3 tools/bench_configs.py > "$O/other_configs.jsonl" 2>&1
echo "configs rc=$?"
# the N-rank drivers rehearsed with two ranks sharing this box's GPU (gloo carries the halo rows through the host)
for C in "--config 4 --check" "--config 4 --all-defined --check" "--config 5 --members 6 --check"; do
  MIFC_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 tools/bench_multigpu.py $C 2>/dev/null | grep "^{" >> "$O/multigpu_gloo_rehearsal.jsonl"
done
echo "multigpu rehearsal rc=$?"; cat "$O/multigpu_gloo_rehearsal.jsonl"
