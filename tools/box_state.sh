#!/bin/bash
# Clocks / power / temperatures of the box WHILE the headline kernel runs (rocm-smi sampled once a second next to a
# long bench.py run), to see what distinguishes the boxes on which the same binary runs at 66 % and at 73.5 % of 8 TB/s.
#   bash tools/box_state.sh > gpurun_out/box_state_under_load.txt
python3 bench.py --steps 25000 --warmup 5 --no-cpu-baseline --no-verify --no-check-variant > /tmp/box_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "fclk|mclk|sclk|socclk|Power \(W\)|Temperature"
  echo "--"
  sleep 1
done
wait $BP
python3 -c "
import json
d=json.loads(open('/tmp/box_bench.json').readline())
print('bench: kernel_ms_avg', d['roofline']['kernel_ms_avg'], 'frac', d['roofline']['frac'], 'placement', d['config']['placement']['probe_ms_min_median_max'])
"
rocm-smi --showuniqueid 2>/dev/null | grep -i unique
rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -iE "partition"
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock Freq|Wavefront|Cacheline|L2:|L3:|Memory Properties|Size:" | sort | uniq -c | head -30
rocm-smi --showrasinfo all 2>/dev/null | grep -iE "UMC|HBM|correct" | head -10
cat /sys/class/drm/card*/device/mem_info_vram_vendor 2>/dev/null | head -2
cat /sys/class/drm/card*/device/vbios_version 2>/dev/null | head -2
cat /sys/module/amdgpu/version 2>/dev/null
