#!/bin/bash
# Collects the judged profiles of the default bench on the GPU box: kernel-trace stats of `python3 bench.py`
# and separate --pmc passes (never combined with other trace domains).  Run from the repo root under gpurun:
#   bash tools/profile_round.sh gpurun_out/prof_rNN
set -e
OUT=$(realpath -m "$1")
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the headline leg only (--no-extra-legs), so that every render launch of the trace carries the 8 frames the JSON line's
# roofline.avg_launch_ms is about; the full default line (with the extra legs) is traced separately
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$REPO/bench.py" --no-extra-legs > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
echo "stats pass (headline leg) done"
# the same with two launches in flight (RT_OPT_BATCH_INFLIGHT = 2): kernels overlap, the trace's average duration is not the
# device time per launch any more (tools/trace_union.py)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_two_in_flight" -o bench -- python3 "$REPO/bench.py" --no-extra-legs --no-cpu-baseline --batch-inflight 2 > "$OUT/bench_two_in_flight_under_rocprof.json" 2> "$OUT/stats_two_in_flight.log"
echo "stats pass (headline leg, two launches in flight) done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_all_legs" -o bench -- python3 "$REPO/bench.py" --no-cpu-baseline > "$OUT/bench_all_legs_under_rocprof.json" 2> "$OUT/stats_all_legs.log"
echo "stats pass (all legs) done"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -o pmc -- python3 "$REPO/bench.py" --no-cpu-baseline --no-extra-legs --steps 48 --warmup 16 --min-seconds 0.05 --clock-warmup 0.2 > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.log"
  echo "pmc pass $i ($set) done"
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT"/pmc[0-9] --json "$OUT/traffic.json" > "$OUT/pmc_summary.txt"
cat "$OUT/pmc_summary.txt"
find "$OUT/stats" -name "*kernel_stats.csv" -exec head -4 {} \;
