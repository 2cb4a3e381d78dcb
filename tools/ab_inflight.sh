#!/bin/bash
# Interleaved A/B of RT_OPT_INFLIGHT (single-frame launches in flight) inside bench.py's own process: bash tools/ab_inflight.sh OUT "4 5" [rounds]
OUT=$1; KS=${2:-"4 5"}; R=${3:-3}; mkdir -p "$OUT"
for r in $(seq 1 $R); do for k in $KS; do
  python bench.py --no-cpu-baseline --inflight $k > "$OUT/hl_$k.$r.json" 2>/dev/null || exit 1
  python bench.py --no-cpu-baseline --bounces 1 --inflight $k > "$OUT/b1_$k.$r.json" 2>/dev/null || exit 1
  python bench.py --no-cpu-baseline --spheres 1024 --inflight $k > "$OUT/c3_$k.$r.json" 2>/dev/null || exit 1
done; done
python - "$OUT" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]
    print("%-14s headline %8.1f   one frame per launch %8.1f (%d in flight)" % (f.split("/")[-1][:-5], d["value"], c["one_frame_per_launch"]["mrays_per_s"], c["one_frame_per_launch"]["launches_in_flight"]))
PY
