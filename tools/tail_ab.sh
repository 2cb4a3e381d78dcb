#!/bin/bash
# Interleaved A/B of builds of librt355 at one RT_OPT_TAIL_SPILL threshold: bash tools/tail_ab.sh OUT "<bench flags>" lib1.so lib2.so ...
OUT=$1; flags=$2; shift 2
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for round in 1 2; do
  for lib in "$@"; do
    RT355_LIB=$(realpath "$lib") python bench.py --no-cpu-baseline --no-extra-legs --steps 100 --warmup 20 $flags 2>/dev/null |
      python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s round $round  %.1f Mrays/s  %.4f ms/launch  %s' % ('$(basename $lib .so)', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel']))" | tee -a "$OUT"
  done
done
true
