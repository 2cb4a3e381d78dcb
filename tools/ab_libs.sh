#!/bin/bash
# Interleaved A/B of several builds of librt355 on ONE box: bash tools/ab_libs.sh OUT lib1.so lib2.so ...
# Each round runs bench.py (headline + one-frame-per-launch + 80-slot legs, no CPU baseline) once per build.
OUT=$1; shift
mkdir -p "$OUT"
for round in 1 2 3; do
  for lib in "$@"; do
    name=$(basename "$lib" .so)
    RT355_LIB=$(realpath "$lib") python bench.py --no-cpu-baseline --min-seconds 0.6 --clock-warmup 1.0 > "$OUT/$name.$round.json" 2> "$OUT/$name.$round.err" || { echo "FAILED $lib"; tail -5 "$OUT/$name.$round.err"; }
    python - "$OUT/$name.$round.json" "$name" "$round" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
c = d["config"]
print("%-16s round %s: headline %8.1f  one-frame %8.1f  80-slot %8.1f  valu.frac %.3f  kernel %s" % (
    sys.argv[2], sys.argv[3], d["value"], c["one_frame_per_launch"]["mrays_per_s"], c["reference_80_slots"]["mrays_per_s"],
    d["valu"]["frac"], d["roofline"]["kernel"]))
PY
  done
done
