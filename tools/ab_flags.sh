#!/bin/bash
# Interleaved A/B of builds of librt355 over several bench configurations: bash tools/ab_flags.sh OUT lib1.so lib2.so -- "<flags 1>" "<flags 2>" ...
OUT=$1; shift
libs=(); while [ "$1" != "--" ]; do libs+=("$1"); shift; done; shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for flags in "$@"; do
  for round in 1 2; do
    for lib in "${libs[@]}"; do
      RT355_LIB=$(realpath "$lib") python bench.py --no-cpu-baseline --no-extra-legs --min-seconds 0.6 --clock-warmup 1.0 $flags 2>/dev/null |
        python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %-12s round $round  %.1f Mrays/s  %s' % ('$flags', '$(basename $lib .so)', d['value'], d['roofline']['kernel']))" | tee -a "$OUT"
    done
  done
done
true
