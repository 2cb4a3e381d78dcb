#!/bin/bash
# Config 3 (1 024 spheres) by the candidate lists' options: Grays/s per (cells, bins per face edge, reach in cells).
#   bash tools/dirlist_sweep.sh OUT.txt [spheres]
OUT=$(realpath -m "$1"); N=${2:-1024}
: > "$OUT"
C="--no-cpu-baseline --no-extra-legs --min-seconds 0.4 --clock-warmup 0.6 --spheres $N"
run() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-60s %8.1f Mrays/s  render %.3f ms (tail %.3f)' % (' '.join(sys.argv[1:]), d['value'], r['avg_launch_ms'], r.get('tail_avg_launch_ms') or 0))" "$@" >> "$OUT"; }
run --dirlist 0
for cells in 2048 4096 8192 16384 32768 65536; do run --dirlist-cells $cells; done
for k in 4 6 8 10 12; do run --dirlist-k $k; done
for reach in 2 4 8 16 32 64; do run --dirlist-reach $reach; done
run --dirlist-cells 4096 --dirlist-k 6
run --dirlist-cells 8192 --dirlist-k 6 --dirlist-reach 8
run --dirlist-cells 32768 --dirlist-k 6 --dirlist-reach 32
for t in 0 16 32 48 64; do run --tail-spill $t; done
cat "$OUT"
