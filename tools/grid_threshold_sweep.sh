for cfg in "--spheres 64" "--spheres 64 --grid-min 64" "--spheres 96" "--spheres 96 --grid-min 64" "--spheres 128" "--spheres 128 --grid-min 64" "--spheres 160" "--spheres 160 --grid-min 1000" "--spheres 256" "--spheres 256 --grid-min 1000"; do
  python bench.py --no-cpu-baseline --no-extra-legs --min-seconds 0.5 --clock-warmup 0.7 $cfg 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-34s %9.1f  %s' % ('$cfg', d['value'], d['roofline']['kernel']))"
done
