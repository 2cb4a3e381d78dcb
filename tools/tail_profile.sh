#!/bin/bash
# kernel-trace stats of the headline bench at several RT_OPT_TAIL_SPILL thresholds: durations of render_blocks and tail_stream
# usage: bash tools/tail_profile.sh <outdir> "<thresholds>" [extra bench flags]
OUT=$(realpath -m "$1"); ts=$2; shift 2
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for t in $ts; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t$t" -o bench -- python3 "$REPO/bench.py" --tail-spill $t --no-cpu-baseline --no-extra-legs --steps 100 --warmup 20 "$@" > "$OUT/t$t.json" 2> "$OUT/t$t.log"
  echo "T=$t"; find "$OUT/t$t" -name "*kernel_stats.csv" -exec head -4 {} \; | cut -d, -f1-8
done
