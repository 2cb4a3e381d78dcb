B="python bench.py --no-cpu-baseline --no-extra-legs --min-seconds 0.6 --clock-warmup 1.0"
run() { echo -n "$* : "; $B "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'Mrays/s', d['roofline']['kernel'], 'avg launch ms', d['roofline']['avg_launch_ms'], 'seg/px', d['config']['segments_per_pixel'])"; }
for n in 60 63 64 65 66 72 80; do run --spheres $n --grid-min 200; done
run --slots 80
run
