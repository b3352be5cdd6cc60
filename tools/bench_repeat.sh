O=gpurun_out/r02_end2
mkdir -p $O
{ rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id"; } > $O/box.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
echo "# six consecutive processes of 'python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline' on ONE box, same binary; placement: best of 3 pools of 48" > $O/bench_repeat.txt
echo "# Mcells/s  ms_per_step(wall)  kernel_ms_avg(HIP events)  roofline.frac  tested-variant frac  verified  chosen ms per pool" >> $O/bench_repeat.txt
for i in 1 2 3 4 5 6; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d.get('check_variant',{}).get('roofline_frac'), d['verified'], d['config']['placement'].get('rounds_chosen_ms'))" >> $O/bench_repeat.txt
done
cat $O/box.txt $O/bench_repeat.txt
