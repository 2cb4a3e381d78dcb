#!/bin/bash
# Which RT_OPT_TAIL_SPILL threshold: all legs of the default bench (headline, one frame per launch, 80 slots) and the 4K frame
# at several thresholds, on one device.  usage: bash tools/tail_defaults_sweep.sh <outfile> "<thresholds>"
out=${1:-gpurun_out/tail_defaults.txt}; ts=${2:-0 24 32 40}
mkdir -p "$(dirname "$out")"; : > "$out"
line() {
  python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']; o = c.get('one_frame_per_launch'); s = c.get('reference_80_slots'); print('%-26s headline %.1f (%.4f ms, tail %s)  one-frame %s  80-slot %s' % ('$1', d['value'], d['roofline']['avg_launch_ms'], d['roofline'].get('tail_avg_launch_ms'), o and o['mrays_per_s'], s and s['mrays_per_s']))"
}
for rep in 1 2; do
  for t in $ts; do
    python bench.py --tail-spill $t --no-cpu-baseline --min-seconds 0.6 --clock-warmup 1.0 2>/dev/null | line "T=$t rep $rep" | tee -a "$out"
  done
  python bench.py --tail-spill 32 --tail-spill-single 32 --no-cpu-baseline --min-seconds 0.6 --clock-warmup 1.0 2>/dev/null | line "T=32, single frames too, rep $rep" | tee -a "$out"
done
for t in $ts; do
  python bench.py --tail-spill $t --width 3840 --height 2160 --no-cpu-baseline --no-extra-legs --min-seconds 0.6 --clock-warmup 1.0 2>/dev/null | line "4K T=$t" | tee -a "$out"
done
true
