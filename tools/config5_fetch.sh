set -e
OUT=$(realpath -m gpurun_out/c5pmc); REPO=$(pwd); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for sky in 2048 8192; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_$sky" -o pmc -- python3 "$REPO/bench.py" --no-cpu-baseline --no-extra-legs --width 3840 --height 2160 --sky $sky --steps 32 --warmup 8 --min-seconds 0.05 --clock-warmup 0.2 > "$OUT/bench_$sky.json" 2> "$OUT/log_$sky.txt"
  echo "sky $sky done"
done
cd $REPO
for sky in 2048 8192; do echo "skybox ${sky}x$((sky/2)):"; python3 tools/pmc_summary.py $OUT/fetch_$sky | grep render; python3 - $OUT/bench_$sky.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print("  segments per launch %.0f, frames per launch %d, alg bytes per launch %.0f, Mrays/s %.0f" % (d["valu"]["kernel_mrays_per_s"]*r["avg_launch_ms"]*1e3, d["config"]["frames_per_launch"], r["alg_bytes_per_launch"], d["value"]))
PY
done
