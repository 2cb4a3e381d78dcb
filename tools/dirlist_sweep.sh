#!/bin/bash
# Config 3 (1 024 spheres) by the candidate lists' options: Grays/s per (cells, bins per face edge, reach in cells, spill threshold).
#   bash tools/dirlist_sweep.sh OUT.txt [spheres]
OUT=$(realpath -m "$1"); N=${2:-1024}
: > "$OUT"
C="--no-cpu-baseline --no-extra-legs --min-seconds 0.4 --clock-warmup 0.6 --spheres $N"
run() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-60s %8.1f Mrays/s  render %.3f ms (tail %.3f)' % (' '.join(sys.argv[1:]), d['value'], r['avg_launch_ms'], r.get('tail_avg_launch_ms') or 0))" "$@" >> "$OUT"; }
run
run --dirlist 0
for cells in 4096 8192 32768; do run --dirlist-cells $cells; done
for k in 6 8 10 12; do for reach in 16 32; do run --dirlist-k $k --dirlist-reach $reach; done; done
for t in 0 24 32 48 64 96; do run --tail-spill $t; done
run --dirlist-k 10 --dirlist-reach 32 --tail-spill 48
run --dirlist-k 12 --dirlist-reach 32 --tail-spill 48
run --dirlist-k 12 --dirlist-reach 32 --tail-spill 64 --dirlist-cells 32768
cat "$OUT"
