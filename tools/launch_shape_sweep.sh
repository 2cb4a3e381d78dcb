#!/bin/bash
# A/B on one device of the launch shapes 3 / 4 and of RT_OPT_BATCH_INFLIGHT: bash tools/launch_shape_sweep.sh gpurun_out/x
OUT=$1; mkdir -p $OUT
run() { # name, args...
  local name=$1; shift
  python bench.py --no-extra-legs --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - "$OUT/$name.json" "$name" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
r = d["roofline"]
print("%-28s %9.1f Mrays/s  valu.frac %.3f  launch %.4f ms (kernel %.4f) resolve %.4f ms  %s" % (sys.argv[2], d["value"], d.get("valu", {}).get("frac", 0), r["avg_launch_ms"], r["kernel_avg_duration_ms"], r["resolve_avg_launch_ms"], r.get("kernel", "")))
PY
}
for round in 1 2 3; do
  run v3_i1.$round --batch-inflight 1
  run v3_i2.$round --batch-inflight 2
done
run v3_i3 --batch-inflight 3
run v3_i4 --batch-inflight 4
run v4_i2 --variant 4 --batch-inflight 2
run v3_i2_b4 --batch-inflight 2 --batch 4
run v3_i1_b4 --batch-inflight 1 --batch 4
run v3_i2_b16 --batch-inflight 2 --batch 16
run v3_i1_b16 --batch-inflight 1 --batch 16
run driver_flags_i2 --steps 20 --warmup 5
run driver_flags_i1 --steps 20 --warmup 5 --batch-inflight 1
