#!/bin/bash
# Interleaved A/B of whole source trees on ONE box (each tree runs its own bench.py against its own build):
#   bash tools/ab_trees.sh OUT . tools/ab/base [more trees]      (trees are built beforehand; tools/ab/ is git-ignored)
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
for round in 1 2 3; do
  for tree in "$@"; do
    name=$(basename "$(realpath "$tree")")
    (cd "$tree" && python bench.py --no-cpu-baseline --min-seconds 0.6 --clock-warmup 1.0 > "$OUT/$name.$round.json" 2> "$OUT/$name.$round.err") || { echo "FAILED $tree"; tail -5 "$OUT/$name.$round.err"; }
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
