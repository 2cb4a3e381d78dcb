#!/bin/bash
# A/B of RT_OPT_TAIL_SPILL on one device: the headline workload (8 frames per launch) at several thresholds, interleaved twice.
# usage: bash tools/tail_spill_sweep.sh <outfile> "<thresholds>" "<refill_mins>" [extra bench flags]
out=${1:-gpurun_out/tail_spill_sweep.txt}
ts=${2:-0 16 32 48 64}
rs=${3:-1}
shift 3
mkdir -p "$(dirname "$out")"; : > "$out"
for rep in 1 2; do
  for t in $ts; do
    for r in $rs; do
      python bench.py --tail-spill $t --refill-min $r --no-cpu-baseline --no-extra-legs --steps 100 --warmup 20 "$@" 2>/dev/null |
        python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T=%-3s refill_min=%-2s rep $rep  %.1f Mrays/s  %.4f ms/launch  %s' % ('$t', '$r', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel']))" | tee -a "$out"
      if [ "$t" = 0 ]; then break; fi
    done
  done
done
