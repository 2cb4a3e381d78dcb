B="python bench.py --no-cpu-baseline --no-extra-legs --min-seconds 0.6 --clock-warmup 1.0"
run() { echo -n "$* : "; $B "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'Mrays/s', d['roofline']['kernel'], 'avg launch ms', d['roofline']['avg_launch_ms'])"; }
run
for k in 8 10 12 14 16; do for c in 2048 4096 8192 16384 32768; do run --dirmask-k $k --dirmask-cells $c; done; done
run
for c in 8192 16384 32768 65536; do for k in 8 10 12; do run --spheres 1024 --dirlist-k $k --dirlist-cells $c; done; done
run --spheres 1024
