B="python bench.py --no-cpu-baseline --no-extra-legs --min-seconds 0.6 --clock-warmup 1.0"
run() { echo -n "$* : "; $B "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'Mrays/s', d['roofline']['kernel'], 'avg launch ms', d['roofline']['avg_launch_ms'])"; }
for r in 1 2; do
run --slots 80 --flags 8
run --slots 80
run --slots 80 --batch 1 --flags 8
run --slots 80 --batch 1
done
run
