# Needs a build whose rt_grid.h reads the cells-per-sphere target from $RT355_GRID_DENSITY (see kGridCellsPerSphere); kept for the record.
for n in 1024 256 128; do for dens in 2 1 0.5 0.35 0.125 0.06; do
  RT355_GRID_DENSITY=$dens python bench.py --no-cpu-baseline --no-extra-legs --min-seconds 0.4 --clock-warmup 0.5 --spheres $n 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('spheres %4d density %-4s %9.1f  %s' % ($n, '$dens', d['value'], d['roofline']['kernel']))"
done; done
