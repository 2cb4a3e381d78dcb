#!/bin/bash
# Where does the sphere grid start to pay?  Grays/s of the headline frame by sphere count, linear scan (re-packing shape or
# plain tiles) against the grid: bash tools/grid_threshold_sweep.sh
for n in 64 72 80 88 96 128; do for gm in 1000 1; do
  python bench.py --no-cpu-baseline --no-extra-legs --min-seconds 0.4 --clock-warmup 0.5 --spheres $n --grid-min $gm 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('spheres %4d  %-7s %9.1f  %s' % ($n, 'linear' if $gm == 1000 else 'grid', d['value'], d['roofline']['kernel']))"
done; done
